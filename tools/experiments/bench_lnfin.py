import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
for (m, c) in [(131072, 320), (32768, 640), (8192, 1280), (2048, 1280), (8192, 320), (512, 1280)]:
    parts = ops.row_stats_parts(c)
    st = torch.randn(m, parts, 2, device="cuda").abs()
    f = lambda: ops.ln_finalize(st, c, 1e-5)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"M={m} C={c} parts={parts}: {e0.elapsed_time(e1) / 20 * 1e3:6.2f} us", flush=True)
