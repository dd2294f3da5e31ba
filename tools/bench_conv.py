"""Micro-benchmark of the igemm / conv launchers on the SD-1.5 decode shapes (developer tool, GPU only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):        # A/B another build of the same ABI (tools/build_dev.sh)
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])

DEV = "cuda"
# (n,h,w,c1,c2,cout,k,up,gn)
SHAPES = [
    (2, 64, 64, 320, 0, 320, 3, 0, 1), (2, 64, 64, 640, 320, 320, 3, 0, 1), (2, 32, 32, 640, 0, 640, 3, 0, 1),
    (2, 32, 32, 1280, 640, 640, 3, 0, 1), (2, 16, 16, 1280, 0, 1280, 3, 0, 1), (2, 16, 16, 1280, 1280, 1280, 3, 0, 1),
    (2, 8, 8, 1280, 0, 1280, 3, 0, 1), (2, 8, 8, 1280, 1280, 1280, 3, 0, 1), (2, 16, 16, 1280, 0, 1280, 3, 1, 0),
    (2, 32, 32, 640, 0, 640, 3, 1, 0), (1, 512, 512, 128, 0, 128, 3, 0, 1), (1, 256, 256, 256, 0, 256, 3, 0, 1),
    (1, 64, 64, 512, 0, 512, 3, 0, 1), (1, 128, 128, 512, 0, 512, 3, 0, 1),
    (2, 64, 64, 320, 0, 320, 1, 0, 0), (2, 64, 64, 320, 0, 960, 1, 0, 0), (2, 64, 64, 1280, 0, 320, 1, 0, 0),
    (2, 32, 32, 640, 0, 640, 1, 0, 0), (2, 16, 16, 1280, 0, 1280, 1, 0, 0), (2, 16, 16, 5120, 0, 1280, 1, 0, 0),
    (2, 8, 8, 1280, 0, 1280, 1, 0, 0), (2, 16, 16, 1280, 0, 3840, 1, 0, 0),
]
if len(sys.argv) > 1:
    mult = int(sys.argv[1])
    SHAPES = [(s[0] * mult,) + s[1:] for s in SHAPES]
if len(sys.argv) > 2 and sys.argv[2] == "nogn":      # production un-fuses GroupNorm for Cout > 160
    SHAPES = [s[:8] + (0,) for s in SHAPES if s[6] == 3]
g = torch.Generator().manual_seed(0)
print("us  TFLOP/s  shape")
for (n, h, w, c1, c2, cout, k, up, gn) in SHAPES:
    x1 = torch.randn(n, h, w, c1, generator=g).to(DEV, torch.bfloat16)
    x2 = torch.randn(n, h, w, c2, generator=g).to(DEV, torch.bfloat16) if c2 else None
    cin = c1 + c2
    pc = ops.PackedConv(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k), torch.zeros(cout), DEV)
    ab = torch.randn(n, cin, 2, generator=g).to(DEV) if gn else None
    f = lambda: ops.conv(x1, pc, x2=x2, gn_ab=ab, gn_silu=bool(gn), upsample=bool(up))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()            # replay 20 launches from a hipGraph: removes the Python launch cost
    with torch.cuda.graph(gr):
        for _ in range(20):
            f()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ho = h * (2 if up else 1)
    fl = 2.0 * n * ho * ho * cout * cin * k * k
    print(f"{us:9.1f} {fl / us / 1e6:8.1f}  n={n} {h}x{w} {c1}+{c2}->{cout} k{k} up{up} gn{gn}")
