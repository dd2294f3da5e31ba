"""Micro-benchmark of the 1x1 / linear launcher on the SD-1.5 decode shapes at model batch 32 (developer tool, GPU only):
plain, residual, GEGLU and folded-LayerNorm epilogues.  20 launches replayed from a hipGraph per shape."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):        # A/B another build of the same ABI (tools/build_dev.sh)
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (rows per sample, cin, cout, kind)  kind: p plain, r residual(+stats), g geglu, l ln-folded, lg ln-folded geglu
ONLY_K = int(os.environ.get("DC_BENCH_K", "0"))
SHAPES = [(4096, 320, 320, "p"), (4096, 320, 320, "r"), (4096, 320, 960, "l"), (4096, 320, 2560, "lg"), (4096, 1280, 320, "r"),
          (1024, 640, 640, "r"), (1024, 640, 1920, "l"), (1024, 640, 5120, "lg"), (1024, 2560, 640, "r"),
          (256, 1280, 1280, "r"), (256, 1280, 3840, "l"), (256, 1280, 10240, "lg"), (256, 5120, 1280, "r"),
          (64, 1280, 1280, "r"), (64, 1280, 10240, "lg"), (64, 5120, 1280, "r")]
g = torch.Generator().manual_seed(0)
print("us  TFLOP/s  GB/s  shape")
for (hw, cin, cout, kind) in SHAPES:
    if ONLY_K and cin != ONLY_K:
        continue
    m = B * hw
    x = torch.randn(1, m, cin, generator=g).to(DEV, torch.bfloat16)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    ln = (1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g), 1e-5) if "l" in kind else None
    pc = ops.PackedConv(w, torch.zeros(cout), DEV, geglu="g" in kind, ln=ln)
    oc = cout // 2 if "g" in kind else cout
    res = torch.randn(1, m, cout, generator=g).to(DEV, torch.bfloat16) if kind == "r" else None
    st = torch.empty((m, ops.row_stats_parts(cout), 2), device=DEV) if kind == "r" and not os.environ.get("DC_BENCH_NOSTATS") else None
    mr = ops.ln_finalize(ops.row_stats(x), cin, 1e-5) if ln is not None else None
    f = lambda: ops.linear(x, pc, residual=res, stats_out=st, ln_stats=mr)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
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
    fl = 2.0 * m * cout * cin
    by = 2.0 * (m * cin + cout * cin + m * oc + (m * cout if res is not None else 0))
    print(f"{us:9.1f} {fl / us / 1e6:8.1f} {by / us / 1e3:8.1f}  M={m} K={cin} N={cout} {kind}")
