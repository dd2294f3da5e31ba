import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffcodec_amd import ops
g = torch.Generator().manual_seed(0)
for (n, hw, c) in [(32, 64, 320), (32, 32, 640), (32, 16, 1280), (2, 64, 320), (2, 32, 640), (2, 16, 1280)]:
    x = torch.randn(n, hw, hw, c, generator=g).to("cuda", torch.bfloat16)
    pc = ops.PackedConv(torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c), torch.zeros(c), "cuda")
    line = f"n={n} {hw}x{hw}x{c} s2:"
    for sk in (None, 1, 2, 3, 4, 6):
        f = lambda: ops.conv(x, pc, stride=2, pad=1, splitk=sk)
        try:
            for _ in range(3):
                f()
        except Exception as e:
            line += f"  sk={sk}: ERR"
            continue
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                f()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        line += f"  sk={sk}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us"
    print(line, flush=True)
