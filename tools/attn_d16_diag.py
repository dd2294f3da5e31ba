"""Developer tool (GPU): one diagnostic pass over the d = 16 ping-pong attention form.  For every launch it reports WHICH outputs differ
from an fp32 SDPA reference: workgroup (sample, head, 512-query block), wave, 32-query block of the wave, lane-in-block range, and which
head-dim columns (the two lane halves own d columns {0-3, 8-11} and {4-7, 12-15}); and the ratio out / ref of the bad rows (a constant
ratio per row = a wrong normaliser, anything else = wrong weights).  usage: DC_LIB_PATH=... python tools/attn_d16_diag.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
torch.manual_seed(0)
d, heads = 16, 8
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def run(b, nq, nk):
    c = heads * d
    q = torch.randn(b, nq, c).to("cuda", torch.bfloat16)
    k = torch.randn(b, nk, c).to("cuda", torch.bfloat16)
    v = torch.randn(b, nk, c).to("cuda", torch.bfloat16)
    qh, kh, vh = (t.float().view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(b, nq, heads, d)
    nbad = 0
    for it in range(launches):
        o = ops.attention(q, k, v, heads).float().view(b, nq, heads, d)
        bad = (o - ref).abs() > 0.05
        if not bad.any():
            continue
        nbad += 1
        rows = bad.any(dim=3).nonzero()                      # (sample, query, head)
        print(f"  launch {it}: {int(bad.sum())} bad elements in {len(rows)} (sample, query, head) rows", flush=True)
        seen = {}
        for s, qi, h in rows.tolist():
            key = (s, h, qi // 512, (qi % 512) // 64, (qi % 64) // 32)
            seen.setdefault(key, []).append(qi % 32)
        for (s, h, qb, wave, u), lanes in sorted(seen.items())[:12]:
            qs = [qb * 512 + wave * 64 + u * 32 + l for l in lanes]
            cols = bad[s, qs, h].any(dim=0).nonzero().flatten().tolist()
            ratio = (o[s, qs[0], h] / ref[s, qs[0], h]).tolist()
            print(f"    sample {s} head {h} qblock {qb} wave {wave} u {u}: lanes {min(lanes)}..{max(lanes)} ({len(lanes)}), bad d columns {cols}, "
                  f"out/ref of the first bad row: {' '.join('%.2f' % r for r in ratio)}", flush=True)
    print(f"b={b} nq={nq} nk={nk}: {b * heads * ((nq + 511) // 512)} workgroups, bad launches {nbad}/{launches}", flush=True)


run(36, 512, 320)
run(32, 512, 1024)
run(36, 600, 320)
