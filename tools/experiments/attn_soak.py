import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
torch.manual_seed(0)
shapes = [(36, 8, 600, 320, 16), (36, 8, 1024, 512, 16), (64, 8, 512, 256, 8), (64, 8, 512, 256, 32), (40, 8, 1100, 1000, 40), (32, 8, 4096, 4096, 40),
          (16, 8, 4096, 4096, 40), (32, 8, 1024, 1024, 80), (34, 8, 700, 330, 80), (40, 4, 512, 256, 64), (32, 8, 256, 256, 160), (8, 2, 512, 512, 128),
          (32, 8, 4096, 77, 40), (32, 8, 1024, 77, 80), (32, 8, 256, 77, 160), (4, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80)]
worst = 0
for (b, heads, nq, nk, d) in shapes:
    c = heads * d
    q = torch.randn(b, nq, c).to("cuda", torch.bfloat16)
    k = torch.randn(b, nk, c).to("cuda", torch.bfloat16)
    v = torch.randn(b, nk, c).to("cuda", torch.bfloat16)
    outs = [ops.attention(q, k, v, heads) for _ in range(16)]
    torch.cuda.synchronize()
    bs = min(b, 4)
    qh, kh, vh = (t[:bs].float().view(bs, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(bs, nq, c)
    err = max(float((o[:bs].float() - ref).abs().max()) for o in outs)
    neq = sum(int(not torch.equal(o, outs[0])) for o in outs)
    worst = max(worst, neq)
    print((b, heads, nq, nk, d), "launches differing from the first:", neq, "of 15;  max err vs SDPA (first 4 samples): %.4f" % err)
print("SOAK", "FAILED" if worst else "OK")
