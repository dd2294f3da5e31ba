import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops
B, nq, d = 16, 4096, 40
c = 8 * d
q = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
k = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
v = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attention(q, k, v, 8)
torch.cuda.synchronize()
