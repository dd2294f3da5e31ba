"""Developer tool: where does a gemm_dma workgroup spend its life?  Builds csrc/gemm_dma.hip with -DDC_STAMP into a scratch
.so, launches one shape and prints the median of the per-workgroup phase durations (shader cycles, s_memtime)."""
import ctypes, os, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffcodec_amd import lib, ops
PKG = os.path.dirname(lib.LIB_PATH)
so = "/tmp/libdc_stamp.so"
srcs = ["igemm.hip", "conv3x3_tile.hip", "gemm_dma.hip", "gemm_wide.hip"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDC_STAMP", "-o", so] +
                      [os.path.join(PKG, "csrc", s) for s in srcs])
L = ctypes.CDLL(so)
L.dc_conv_igemm_bf16.argtypes = [ctypes.POINTER(lib.ConvDesc), ctypes.c_void_p]
n, h, c = 32, 64, 320
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 320
g = torch.Generator().manual_seed(0)
x = torch.randn(n, h, h, c, generator=g).to("cuda", torch.bfloat16)
pc = ops.PackedConv(torch.randn(cout, c, 1, 1, generator=g) / math.sqrt(c), torch.zeros(cout), "cuda")
out = torch.empty(n, h, h, cout, device="cuda", dtype=torch.bfloat16)
m = n * h * h
nblk = math.ceil(m / 128) * math.ceil(cout / 160)
ws = torch.zeros(nblk * 8, device="cuda", dtype=torch.int64)
d = lib.ConvDesc(x1=x.data_ptr(), x2=0, w=pc.w.data_ptr(), bias=pc.bias.data_ptr(), gn_ab=0, row_add=0, residual=0, out=out.data_ptr(),
                 splitk_ws=ws.data_ptr(), N=n, H=h, W=h, C1=c, C2=0, Cout=cout, ksize=1, stride=1, pad=1, upsample=0, Ho=h, Wo=h,
                 gn_silu=0, epilogue=0, out_f32=0, out_scale=1.0, splitk=1, gn_batch=0, act=0, row_add_stride=0, ln_stats=0, ln_colsum=0, stats_out=0, gn_part_out=0)
for _ in range(3):
    assert L.dc_conv_igemm_bf16(ctypes.byref(d), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = ws.view(nblk, 8).cpu().double()
ph = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0], t[:, 4] - t[:, 2], t[:, 5] - t[:, 4],
                  t[:, 6] - t[:, 5], t[:, 3] - t[:, 6]], 1)
print("workgroups", nblk, " median cycles: prologue(first stage) %.0f | K loop %.0f | epilogue %.0f | total %.0f\n   epilogue parts: barrier-in %.0f | "
      "operand loads + math + ds_write %.0f | barrier-mid %.0f | row stores %.0f" % tuple(ph.median(0).values.tolist()))
print("kernel span (cycles): %.0f" % (t[:, 3].max() - t[:, 0].min()).item())
