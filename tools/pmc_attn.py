"""Developer tool (GPU, under rocprofv3 --pmc ...): three launches of the 64x64 self-attention of the decode loop (d = 40, 8 heads).
usage: [DC_LIB_PATH=...] python3 tools/pmc_attn.py [model batch, default 32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nq, d = 4096, 40
c = 8 * d
q = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
k = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
v = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attention(q, k, v, 8)
torch.cuda.synchronize()
