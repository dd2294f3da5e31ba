"""experiment: two independent half-batch pipelines on two streams vs one full batch (same 16 units)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from diffcodec_amd import clip_decode as CD
from diffcodec_amd.synthetic import synth_text
bench.torch = torch
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
F = 16
units = CD.plan_units(1 + 2 * 12, 12, 512, 512)[:F]
pe, npe = (t.to(dev) for t in synth_text(1))
kw = dict(num_inference_steps=20, guidance_scale=4.5, controlnet_conditioning_scale=1.7)

def make(units_sub, dual):
    pipe, _ = bench.build_pipeline(0, dev)
    pipe.enable_hip_graphs(True, steps_per_graph=1)
    pipe.enable_dual_stream(dual)
    srcs = []
    for s in range(2):
        src = CD.SyntheticSource(512, 512, device=dev, seed=1234 + 17 * s)
        fpn = sorted({(u.frame, u.prev, u.next) for u in units_sub})
        noise = {f: CD.frame_noise(f, 512, 512, 4321 + s).to(dev) for f, _, _ in fpn}
        srcs.append(CD.ResidentSource(src, fpn, noise=noise))
    return pipe, srcs

def run(pipe, srcs, us, i):
    return CD.decode_units(pipe, us, srcs[i % 2], pe, npe, batch=len(us), frame_size=(512, 512), **kw)

full = make(units, True)
for i in range(2):
    run(*full, units, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    run(*full, units, 2 + i)
torch.cuda.synchronize()
print("one pipeline, 16 units: %.1f ms per step" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
del full
torch.cuda.empty_cache()
for dual in (False, True):
    a, b = make(units[:8], dual), make(units[8:], dual)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def work(p, us, st, n0, n):
        with torch.cuda.stream(st):
            for i in range(n):
                run(*p, us, n0 + i)
    for mode in ("threads", "one thread"):
        def step(n0, n):
            if mode == "threads":
                ta = threading.Thread(target=work, args=(a, units[:8], sa, n0, n)); tb = threading.Thread(target=work, args=(b, units[8:], sb, n0, n))
                ta.start(); tb.start(); ta.join(); tb.join()
            else:
                for i in range(n):
                    work(a, units[:8], sa, n0 + i, 1); work(b, units[8:], sb, n0 + i, 1)
        for i in range(2):                       # warm-up and graph capture: one thread, one pipeline at a time
            work(a, units[:8], sa, i, 1); torch.cuda.synchronize(); work(b, units[8:], sb, i, 1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(2, 3)
        torch.cuda.synchronize()
        print("two pipelines of 8 units, dual_stream=%s, %s: %.1f ms per 16 units" % (dual, mode, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del a, b
    torch.cuda.empty_cache()
