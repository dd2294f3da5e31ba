"""Split-K sweep on the weight-streaming 3x3 layers (developer tool, GPU only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops

g = torch.Generator().manual_seed(0)
for (n, h, cin, cout, k) in [(32, 8, 1280, 1280, 3), (32, 8, 2560, 1280, 3), (2, 8, 1280, 1280, 3), (2, 16, 1280, 1280, 3),
                             (32, 8, 5120, 1280, 1), (32, 8, 1280, 1280, 1)]:
    x = torch.randn(n, h, h, cin, generator=g).to("cuda", torch.bfloat16)
    pc = ops.PackedConv(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k), torch.zeros(cout), "cuda")
    for sk in (1, 2, 3, 4, 6, 8, 16):
        f = lambda: ops.conv(x, pc, splitk=sk, pad=k // 2)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                f()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"n={n} {h}x{h} {cin}->{cout} k{k} splitk={sk}: {us:8.1f} us  {2.0*n*h*h*cout*cin*k*k/us/1e6:7.1f} TF/s", flush=True)
