import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
g = torch.Generator().manual_seed(0)
for (n, hw, cin, cout) in [(32, 64, 320, 4), (2, 64, 320, 4), (16, 512, 128, 4), (1, 512, 128, 4)]:
    x = torch.randn(n, hw, hw, cin, generator=g).to("cuda", torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    pc = ops.PackedConv(w, torch.zeros(cout), "cuda", mfma_small_cout=True)
    ab = ops.group_norm_ab(x, torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), 32, 1e-5)
    f = lambda: ops.conv(x, pc, gn_ab=ab, gn_silu=True, out_f32=True)
    ref = f().clone()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            f()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"n={n} {hw}x{hw} {cin}->{cout}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us   checksum {float(ref.double().sum()):.6f} absmax {float(ref.abs().max()):.5f}", flush=True)
