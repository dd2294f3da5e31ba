"""Micro-benchmark of dc_attention_bf16 on the SD-1.5 decode shapes (developer tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):        # A/B another build of the same ABI (tools/build_dev.sh)
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for (nq, nk, d) in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80)]:
    c = 8 * d
    q = torch.randn(B, nq, c, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, nk, c, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, nk, c, device="cuda").to(torch.bfloat16)
    f = lambda: ops.attention(q, k, v, 8)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            f()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{us:9.1f} us {4.0 * B * 8 * nq * nk * d / us / 1e6:8.1f} TFLOP/s  B={B} Nq={nq} Nk={nk} d={d}")
