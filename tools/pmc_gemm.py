import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops
g = torch.Generator().manual_seed(0)
n, h, c, cout = 32, 64, 320, int(sys.argv[1]) if len(sys.argv) > 1 else 320
x = torch.randn(n, h, h, c, generator=g).to("cuda", torch.bfloat16)
pc = ops.PackedConv(torch.randn(cout, c, 1, 1, generator=g) / math.sqrt(c), torch.zeros(cout), "cuda")
for _ in range(4):
    y = ops.conv(x, pc)
torch.cuda.synchronize()
