"""Launch the long-K 1x1 / linear shapes of the decode at model batch 32 a few times each (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
passes: tools/pmc_summary.py tells the shapes of one kernel apart by grid size).  Developer tool, GPU only.
usage: python tools/pmc_gemm.py [batch]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
g = torch.Generator().manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (rows per sample, K, N, kind): r = bias + residual + row statistics, l = folded LayerNorm, lg = folded LayerNorm + GEGLU
for (hw, cin, cout, kind) in [(4096, 1280, 320, "r"), (1024, 640, 640, "r"), (1024, 2560, 640, "r"), (256, 1280, 1280, "r"), (256, 5120, 1280, "r"),
                              (1024, 640, 1920, "l"), (256, 1280, 3840, "l"), (1024, 640, 5120, "lg"), (256, 1280, 10240, "lg")]:
    m = B * hw
    x = torch.randn(1, m, cin, generator=g).to("cuda", torch.bfloat16)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    ln = (1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g), 1e-5) if "l" in kind else None
    pc = ops.PackedConv(w, torch.zeros(cout), "cuda", geglu="g" in kind, ln=ln)
    oc = cout // 2 if "g" in kind else cout
    res = torch.randn(1, m, cout, generator=g).to("cuda", torch.bfloat16) if kind == "r" else None
    st = torch.empty((m, ops.row_stats_parts(cout), 2), device="cuda") if kind == "r" else None
    mr = ops.ln_finalize(ops.row_stats(x), cin, 1e-5) if ln is not None else None
    for _ in range(5):
        y = ops.linear(x, pc, residual=res, stats_out=st, ln_stats=mr)
    torch.cuda.synchronize()
    by = 2 * (m * cin + cout * cin + m * oc + (m * cout if res is not None else 0))
    print(f"shape M={m} K={cin} N={cout} {kind}: algorithmic bytes = {by}")
