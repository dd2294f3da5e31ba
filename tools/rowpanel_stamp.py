"""Developer tool: where does a gemm_rowpanel workgroup spend its life?  Builds csrc/gemm_rowpanel.hip (+ the dispatching files)
with -DDC_STAMP into a scratch .so, launches one shape and prints medians over workgroups of the phase durations of wave 0
(shader cycles, s_memtime).  usage: rowpanel_stamp.py <cout> <kind p|r|l|lg> [rows]"""
import ctypes, os, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffcodec_amd import lib, ops
PKG = os.path.dirname(lib.LIB_PATH)
so = "/tmp/libdc_rpstamp.so"
srcs = ["igemm.hip", "conv3x3_tile.hip", "gemm_dma.hip", "gemm_wide.hip", "gemm_rowpanel.hip"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDC_STAMP", "-o", so] +
                      [os.path.join(PKG, "csrc", s) for s in srcs])
L = ctypes.CDLL(so)
L.dc_conv_igemm_bf16.argtypes = [ctypes.POINTER(lib.ConvDesc), ctypes.c_void_p]
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 320
kind = sys.argv[2] if len(sys.argv) > 2 else "p"
m = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
c = 320
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 1, m, c, generator=g).to("cuda", torch.bfloat16)
ln = (1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g), 1e-5) if "l" in kind else None
pc = ops.PackedConv(torch.randn(cout, c, generator=g) / math.sqrt(c), torch.zeros(cout), "cuda", geglu="g" in kind, ln=ln)
oc = cout // 2 if "g" in kind else cout
out = torch.empty(1, 1, m, oc, device="cuda", dtype=torch.bfloat16)
res = torch.randn(1, 1, m, cout, generator=g).to("cuda", torch.bfloat16) if kind == "r" else None
mr = ops.ln_finalize(ops.row_stats(x), c, 1e-5) if ln is not None else None
nblk = m // 256
ws = torch.zeros(nblk * 64, device="cuda", dtype=torch.int64)
d = lib.ConvDesc(x1=x.data_ptr(), x2=0, w=pc.w.data_ptr(), bias=pc.bias.data_ptr(), gn_ab=0, row_add=0, residual=res.data_ptr() if res is not None else 0,
                 out=out.data_ptr(), splitk_ws=ws.data_ptr(), N=1, H=1, W=m, C1=c, C2=0, Cout=cout, ksize=1, stride=1, pad=1, upsample=0, Ho=1, Wo=m,
                 gn_silu=0, epilogue=1 if "g" in kind else 0, out_f32=0, out_scale=1.0, splitk=1, gn_batch=0, act=0, row_add_stride=0,
                 ln_stats=mr.data_ptr() if mr is not None else 0, ln_colsum=pc.colsum.data_ptr() if mr is not None else 0, stats_out=0, gn_part_out=0)
for _ in range(3):
    assert L.dc_conv_igemm_bf16(ctypes.byref(d), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = ws.view(nblk, 64).cpu().double()
S = min(cout // 64, 20)
med = lambda v: v.median().item()
print(f"M={m} N={cout} kind={kind}: workgroups {nblk}, stages {cout // 64}")
print("  prologue (A frags + 2 W stages landed): %.0f cycles" % med(t[:, 1] - t[:, 0]))
for s in range(S):
    top, mf, ep = t[:, 2 + 3 * s], t[:, 3 + 3 * s], t[:, 4 + 3 * s]
    nxt = t[:, 2 + 3 * (s + 1)] if s + 1 < S else None
    print("  stage %2d: issue+MFMA %6.0f | epilogue %6.0f | wait+barrier to next stage %6.0f" %
          (s, med(mf - top), med(ep - mf), med(nxt - ep) if nxt is not None else float("nan")))
last = 4 + 3 * (S - 1)
print("  workgroup lifetime (to the end of stage %d): %.0f cycles; first-round workgroups only: %.0f" %
      (S - 1, med(t[:, last] - t[:, 0]), med((t[:, last] - t[:, 0])[:256])))
print("  kernel span: %.0f cycles" % (t[:, last].max() - t[:, 0].min()).item())
