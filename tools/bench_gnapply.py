"""Micro-benchmark of dc_gn_apply_nhwc_bf16 (GroupNorm affine + SiLU pass) on the decode shapes at model batch 32 / 2 (developer tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
for (n, hw, c) in [(32, 8, 1280), (32, 8, 2560), (32, 16, 1280), (32, 16, 2560), (32, 32, 640), (32, 32, 320), (32, 64, 320), (2, 64, 320), (2, 8, 1280)]:
    x = torch.randn(n, hw, hw, c, device="cuda").to(torch.bfloat16)
    ab = torch.randn(n, c, 2, device="cuda")
    f = lambda: ops.gn_apply(x, ab, silu=True)
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
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"n={n} {hw}x{hw}x{c}: {us:7.1f} us  {x.numel() * 4 / us / 1e3:7.1f} GB/s", flush=True)
