"""conv_in 4 -> 320 timing (developer tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops
for n in (2, 32):
    x = torch.randn(n, 64, 64, 4, device="cuda").to(torch.bfloat16)
    pc = ops.PackedConv(torch.randn(320, 4, 3, 3) / 6, torch.zeros(320), "cuda")
    f = lambda: ops.conv(x, pc)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"n={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
