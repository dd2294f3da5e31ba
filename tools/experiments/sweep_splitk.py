"""which split-K would the best be, and what does the heuristic pick?  every 3x3 / 1x1 shape of the decode with < 512 tiles."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffcodec_amd import ops
g = torch.Generator().manual_seed(0)
shapes = []
for n in (32, 2):
    shapes += [(n, 8, 1280, 1280, 3), (n, 8, 2560, 1280, 3), (n, 16, 1280, 1280, 3), (n, 16, 2560, 1280, 3), (n, 16, 1920, 1280, 3), (n, 16, 640, 1280, 3),
               (n, 32, 640, 640, 3), (n, 32, 1280, 640, 3), (n, 32, 320, 640, 3), (n, 8, 1280, 1280, 1), (n, 8, 5120, 1280, 1), (n, 8, 2560, 1280, 1),
               (n, 16, 1280, 1280, 1), (n, 16, 5120, 1280, 1), (n, 16, 2560, 1280, 1), (n, 32, 640, 640, 1), (n, 32, 2560, 640, 1)]
shapes += [(2, 64, 320, 320, 3), (2, 64, 640, 320, 3), (2, 64, 960, 320, 3), (2, 64, 320, 320, 1), (2, 64, 1280, 320, 1)]
for (n, h, cin, cout, k) in shapes:
    x = torch.randn(n, h, h, cin, generator=g).to("cuda", torch.bfloat16)
    pc = ops.PackedConv(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k), torch.zeros(cout), "cuda")
    res = {}
    for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, None, 1):
        try:
            f = lambda: ops.conv(x, pc, splitk=sk, pad=k // 2)
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10):
                    f()
            gr.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10 * 1e3
            res[sk] = min(t, res.get(sk, 1e9))
        except Exception:
            pass
    best = min((v, s) for s, v in res.items() if s is not None)
    flag = "  <-- heuristic %.0f%% slower" % (100 * (res[None] / best[0] - 1)) if res[None] > 1.06 * best[0] else ""
    m = n * h * h; kt = (9 if k == 3 else 1) * (cin // 64)
    tile3 = k == 3
    pick = ops._pick_splitk(m, cout, kt, cin // 64 if tile3 else None, rows_per_image=h * h)
    tab = " ".join(f"{s_}:{v:.0f}" for s_, v in sorted((a, b) for a, b in res.items() if a is not None))
    print(f"n={n} {h}x{h} {cin}->{cout} k{k}: pick {pick} = {res[None]:6.1f} us | best sk={best[1]} {best[0]:6.1f} us{flag} | {tab}", flush=True)
