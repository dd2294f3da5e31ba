"""Developer tool: where does a long-context attention wave spend its life?  Builds csrc/attention.hip with -DDC_STAMP into a scratch
.so, runs the d = 40 64x64 self-attention once and prints median per-wave phase sums (shader cycles, s_memtime)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffcodec_amd import lib
PKG = os.path.dirname(lib.LIB_PATH)
so = "/tmp/libdc_astamp.so"
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDC_STAMP", "-o", so] + extra +
                      [os.path.join(PKG, "csrc", "attention.hip")])
L = ctypes.CDLL(so)
vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
L.dc_attention_bf16.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ctypes.c_float, vp]
L.dc_attn_stamp_read.argtypes = [vp, ci]
B, H, N, D = 32, 8, 4096, 40
q = torch.randn(B, N, 3 * H * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B, N, H * D, device="cuda", dtype=torch.bfloat16)
st = 3 * H * D
for _ in range(2):
    rc = L.dc_attention_bf16(q.data_ptr(), q.data_ptr() + H * D * 2, q.data_ptr() + 2 * H * D * 2, o.data_ptr(), B, H, N, N, D, st, st, st, H * D,
                             D ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
nw = 1 << 15
buf = torch.zeros(nw * 8, dtype=torch.int64)
assert L.dc_attn_stamp_read(buf.data_ptr(), nw * 8) == 0
t = buf.view(nw, 8).double()
t = t[t[:, 4] > 0]
tiles = N // 64
m = lambda c: t[:, c].median().item()
print("waves sampled", len(t), "key tiles per wave", tiles)
print("median cycles per key tile and wave: QK^T + row max %.0f | exp + convert (+ rescale) %.0f | PV %.0f | stage next tile + barrier %.0f | sum %.0f" %
      (m(0) / tiles, m(1) / tiles, m(2) / tiles, m(3) / tiles, (m(0) + m(1) + m(2) + m(3)) / tiles))
if t[:, 5].max() > 0:
    print("  (ping-pong form: the four columns are MFMA block {PV(t-1), QK^T(t)} | barrier wait | softmax block | barrier wait)")
print("MFMA floor per tile and wave: QK^T 12 x 32 = 384, PV 16 x 32 = 512;  whole wave %.0f cycles" % m(4))
