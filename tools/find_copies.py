#!/usr/bin/env python3
"""Developer tool: where do the device-to-device copies (`__amd_rocclr_copyBuffer`) and other torch-side kernels of one eager
single-frame decode come from?  Runs one decode under torch.profiler with Python stacks and prints, per torch operator that
launched a copy / fill / ATen kernel, the call count and the innermost frames of this repo.
usage: python tools/find_copies.py [frames]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from diffcodec_amd import clip_decode as CD
from diffcodec_amd.synthetic import synth_text


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    device = torch.device("cuda:0")
    pipe, _ = bench.build_pipeline(0, device)
    pipe.enable_hip_graphs(False)
    pipe.enable_dual_stream(False)
    units = CD.plan_units(13, 12, 512, 512)[:F]
    src = CD.SyntheticSource(512, 512, device=device, seed=1234)
    fpn = sorted({(u.frame, u.prev, u.next) for u in units})
    noise = {f: CD.frame_noise(f, 512, 512, 4321).to(device) for f, _, _ in fpn}
    rs = CD.ResidentSource(src, fpn, noise=noise)
    pe, npe = (t.to(device) for t in synth_text(1))
    kw = dict(num_inference_steps=20, guidance_scale=4.5, controlnet_conditioning_scale=1.7)
    for _ in range(2):
        CD.decode_units(pipe, units, rs, pe, npe, batch=F, frame_size=(512, 512), **kw)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        CD.decode_units(pipe, units, rs, pe, npe, batch=F, frame_size=(512, 512), **kw)
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    agg = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue                                         # outermost ATen call only
        if not ev.kernels:
            continue
        stack = [s for s in (ev.stack or []) if root in s or "diffcodec" in s]
        where = " <- ".join(s.replace(root + "/", "") for s in stack[:3]) or "(no repo frame)"
        agg[(ev.name, where, tuple(sorted({k.name[:60] for k in ev.kernels})))] += 1
    for (name, where, kernels), n in agg.most_common(60):
        print(f"{n:6d}  {name:28s} {where}\n        kernels: {', '.join(kernels)}")


if __name__ == "__main__":
    main()
