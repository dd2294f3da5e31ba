"""Developer tool (GPU): outputs of the 1x1 / linear launcher on the long-K decode shapes, saved for a bit-comparison between two builds /
knob settings (DC_GEMM_P8=0 vs 2 with the developer library).  usage: python tools/check_gemm_p8.py save <file> | cmp <a> <b>"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        eq = torch.equal(a[k], b[k])
        d = (a[k].float() - b[k].float()).abs().max().item()
        print(f"{k}: {'bit-identical' if eq else 'DIFFERENT'}  max |diff| {d:.4g}  (max |value| {a[k].float().abs().max().item():.3g})")
    sys.exit(0)
from diffcodec_amd import lib, ops
if os.environ.get("DC_LIB_PATH"):
    lib.LIB_PATH = os.path.abspath(os.environ["DC_LIB_PATH"])
g = torch.Generator().manual_seed(0)
out = {}
for (m, cin, cout, kind) in [(8192, 1280, 3840, "l"), (8192, 1280, 10240, "lg"), (32768, 640, 5120, "lg"), (8192, 1280, 1280, "p"), (2048, 1280, 10240, "lg"),
                             (8192, 1280, 1280, "r"), (512, 128, 256, "p"), (256, 192, 512, "g"), (1024, 320, 2560, "lg"), (768, 448, 768, "l"), (32768, 640, 1920, "l"), (1024, 256, 640, "lg"), (512, 128, 384, "p"), (14336, 640, 1920, "g"), (8192, 256, 4224, "lg"), (16384, 128, 2560, "g"), (9984, 384, 3584, "g")]:
    x = torch.randn(1, m, cin, generator=g).to("cuda", torch.bfloat16)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    ln = (1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g), 1e-5) if "l" in kind else None
    pc = ops.PackedConv(w, 0.1 * torch.randn(cout, generator=g), "cuda", geglu="g" in kind, ln=ln)
    res = torch.randn(1, m, cout, generator=g).to("cuda", torch.bfloat16) if kind == "r" else None
    mr = ops.ln_finalize(ops.row_stats(x), cin, 1e-5) if ln is not None else None
    y = ops.linear(x, pc, residual=res, ln_stats=mr, out_scale=0.75 if kind == "r" else 1.0)
    torch.cuda.synchronize()
    for _ in range(3):
        y2 = ops.linear(x, pc, residual=res, ln_stats=mr, out_scale=0.75 if kind == "r" else 1.0)
        assert torch.equal(y, y2), f"launch-to-launch difference at M={m} K={cin} N={cout} {kind}"
    out[f"M={m} K={cin} N={cout} {kind}"] = y.cpu()
torch.save(out, sys.argv[2])
print("saved", len(out), "outputs to", sys.argv[2])
