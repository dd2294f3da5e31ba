"""Developer tool: where does a conv3x3_tile wave spend its life?  Builds the conv sources with -DDC_STAMP into a scratch .so,
launches one shape and prints the median per-wave phase durations (shader cycles, s_memtime)."""
import ctypes, os, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffcodec_amd import lib, ops
PKG = os.path.dirname(lib.LIB_PATH)
so = "/tmp/libdc_cstamp.so"
srcs = ["igemm.hip", "conv3x3_tile.hip", "gemm_dma.hip", "gemm_wide.hip", "gemm_rowpanel.hip", "norm.hip"]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDC_STAMP", "-o", so] + extra +
                      [os.path.join(PKG, "csrc", s) for s in srcs])
L = ctypes.CDLL(so)
L.dc_conv_igemm_bf16.argtypes = [ctypes.POINTER(lib.ConvDesc), ctypes.c_void_p]
args = [a for a in sys.argv[1:] if not a.startswith("-D")]
n, h, c, cout = [int(a) for a in args[:4]] if len(args) >= 4 else (32, 64, 320, 320)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, h, h, c, generator=g).to("cuda", torch.bfloat16)
pc = ops.PackedConv(torch.randn(cout, c, 3, 3, generator=g) / math.sqrt(9 * c), torch.zeros(cout), "cuda")
out = torch.empty(n, h, h, cout, device="cuda", dtype=torch.bfloat16)
nblk = n * (h // 8) * (h // 16) * math.ceil(cout / 160)
ws = torch.zeros(nblk * 4 * 8, device="cuda", dtype=torch.int64)
d = lib.ConvDesc(x1=x.data_ptr(), x2=0, w=pc.w.data_ptr(), bias=pc.bias.data_ptr(), gn_ab=0, row_add=0, residual=0, out=out.data_ptr(),
                 splitk_ws=ws.data_ptr(), N=n, H=h, W=h, C1=c, C2=0, Cout=cout, ksize=3, stride=1, pad=1, upsample=0, Ho=h, Wo=h,
                 gn_silu=0, epilogue=0, out_f32=0, out_scale=1.0, splitk=1, gn_batch=0, act=0, row_add_stride=0, ln_stats=0, ln_colsum=0,
                 stats_out=0, gn_part_out=0)
for _ in range(3):
    assert L.dc_conv_igemm_bf16(ctypes.byref(d), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.dc_conv_igemm_bf16(ctypes.byref(d), torch.cuda.current_stream().cuda_stream) == 0
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = ws.view(nblk * 4, 8).cpu().double()
steps = (c // 64) * 9
med = lambda v: v.median().item()
print("waves", nblk * 4, "K-steps", steps)
print("median cycles per wave: prologue %.0f | K loop %.0f | epilogue %.0f | total %.0f" %
      (med(t[:, 1] - t[:, 0]), med(t[:, 2] - t[:, 1]), med(t[:, 3] - t[:, 2]), med(t[:, 3] - t[:, 0])))
print("  K loop per step: sync (vmcnt+lgkmcnt wait + barrier) %.0f | fragment reads + DMA issue + MFMA issue %.0f | halo swap (per 9 steps) %.0f" %
      (med(t[:, 4]) / steps, med(t[:, 5]) / steps, med(t[:, 6]) / max(1, c // 64 - 1)))
print("  MFMA cycles per step per wave (40 x 16): 640")
e = ws.view(nblk * 4, 8).cpu()[:, 7]
parts = [((e >> sh) & 0xffff).double().median().item() for sh in (0, 16, 32, 48)]
print("  epilogue parts: operand loads + barrier-in %.0f | math + ds_write %.0f | barrier-mid %.0f | row reads + stores %.0f" % tuple(parts))
ok = t[:, 0] > 0
span = (t[ok, 3].max() - t[ok, 0].min()).item()
print("kernel span %.0f cycles in %.1f us (event-timed) = %.2f GHz shader clock; waves stamped %d of %d" % (span, us, span / us / 1e3, int(ok.sum()), nblk * 4))
# rounds: sort workgroup start times per CU slot is unknown; print start-time quantiles relative to the first start
st = (t[ok, 0] - t[ok, 0].min()).sort().values
print("wave start time quantiles (cycles): " + " ".join("%.0f" % st[int(q * (len(st) - 1))].item() for q in (0.0, 0.24, 0.26, 0.49, 0.51, 0.74, 0.76, 1.0)))
