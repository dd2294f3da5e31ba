"""Micro-benchmark of the fp32 NCHW extractor convs (developer tool, GPU only): the 19 shapes of one 16-frame control pyramid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [(3, 16, 512, 1), (16, 32, 512, 2), (32, 32, 256, 1), (32, 64, 256, 2), (64, 64, 128, 1), (64, 160, 128, 2), (160, 160, 64, 2),
          (160, 320, 64, 1), (160, 64, 64, 1), (160, 320, 32, 2), (160, 320, 32, 1), (320, 640, 16, 2), (320, 640, 16, 1), (640, 1280, 8, 1),
          (640, 64, 8, 1), (320, 64, 16, 1), (160, 64, 32, 1), (64, 1, 64, 1), (64, 1, 8, 1)]
g = torch.Generator().manual_seed(0)
tot = 0.0
for (cin, cout, hw, s) in SHAPES:
    x = torch.randn(N, cin, hw, hw, generator=g).cuda()
    pc = ops.PackedConvF32(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, torch.zeros(cout), "cuda")
    f = lambda: ops.conv3x3_nchw_f32(x, pc, stride=s, silu=True)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    tot += us
    print(f"{us:9.1f} us {2.0 * y.numel() * cin * 9 / us / 1e6:7.1f} TFLOP/s  N={N} {cin}->{cout} {hw}x{hw} s{s}")
print(f"sum {tot / 1e3:.2f} ms")
