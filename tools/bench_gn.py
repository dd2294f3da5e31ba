"""GroupNorm statistics: chunk-slab + finalize pair vs the one-launch small-tensor kernel (developer tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops

for (n, hw, c) in [(2, 64, 320), (2, 32, 640), (2, 16, 1280), (2, 8, 1280), (2, 8, 2560), (2, 64, 960), (32, 8, 1280), (32, 8, 2560), (32, 16, 1280), (32, 32, 640)]:
    x = torch.randn(n, hw, hw, c, device="cuda").to(torch.bfloat16)
    ga, be = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    res = []
    for limit in (0, 1 << 40):
        ops.GN_DIRECT_MAX_PIXELS = limit
        f = lambda: ops.group_norm_ab(x, ga, be, 32, 1e-5)
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
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"n={n} {hw}x{hw}x{c} ({x.numel()*2/2**20:.1f} MiB): slabs+finalize {res[0]:7.1f} us   direct {res[1]:7.1f} us", flush=True)
