"""Launch the dominant implicit-GEMM shapes a few times (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (h, c, cout, k) in [(64, 320, 320, 3), (32, 640, 640, 3), (64, 320, 320, 1)]:
    x = torch.randn(n, h, h, c, generator=g).to("cuda", torch.bfloat16)
    pc = ops.PackedConv(torch.randn(cout, c, k, k, generator=g) / math.sqrt(c * k * k), torch.zeros(cout), "cuda")
    for _ in range(5):
        y = ops.conv(x, pc)
    torch.cuda.synchronize()
    print(f"shape n={n} {h}x{h} {c}->{cout} k{k}: algorithmic bytes = {x.numel()*2 + pc.w.numel()*2 + y.numel()*2}")
