#!/usr/bin/env python3
"""bench.py — decoded frames/s of the DiffCodec decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--no-graphs]

A "step" is one pass of the hot path over one batch of F synthetic 512x512 inter frames resident in HBM:
control pyramid + FDN gamma/beta (once per frame), 20 DDIM steps of {DualFlowControlNet, UNet} with CFG
(model batch 2F), CFG+DDIM update, VAE decode, postprocess.  SD-1.5 topology, random-init weights in the
diffusers key layout (no checkpoints are reachable offline), bf16 compute with fp32 accumulation.
N > 1: one process per GPU (torch.distributed / RCCL); frames are sharded across ranks with no data-path
collective; rank 0 synthesises the weights and broadcasts the packed tensors once (outside the timed region).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

STEPS_DDIM = 20
SIZE = 512
# algorithmic work per decoded frame (CFG on, step-invariant parts hoisted): TFLOP
TFLOP_PER_FRAME = (20 * 2 * (803.3 + 268.6) + 30.2 + 19.9 + 2514.5) / 1000.0
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_pipeline(rank, device):
    from diffcodec_amd import weights as W
    from diffcodec_amd.controlnet import HipDualFlowControlNet
    from diffcodec_amd.pipeline import StableDiffusionDualFlowControlNetPipeline
    from diffcodec_amd.scheduler import DDIMScheduler
    from diffcodec_amd.unet import HipUNet2DConditionModel
    from diffcodec_amd.vae import HipAutoencoderKL
    t0 = time.time()
    specs = (W.unet_spec(), W.controlnet_spec(), W.vae_spec())
    if rank == 0:
        sds = [W.synthesize(s, seed=i) for i, s in enumerate(specs)]
    else:   # shapes only; the packed device tensors are overwritten by the broadcast from rank 0
        sds = [{k: torch.empty(shape) for k, (_, shape) in s.items()} for s in specs]
    log(f"[rank {rank}] state dicts ready in {time.time() - t0:.1f}s")
    unet = HipUNet2DConditionModel(sds[0], W.SD15_UNET_CONFIG, device)
    cn = HipDualFlowControlNet(sds[1], W.SD15_UNET_CONFIG, device)
    vae = HipAutoencoderKL(sds[2], W.SD15_VAE_CONFIG, device)
    pipe = StableDiffusionDualFlowControlNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, controlnet=cn,
                                                     scheduler=DDIMScheduler(), safety_checker=None, feature_extractor=None)
    log(f"[rank {rank}] operators packed + uploaded in {time.time() - t0:.1f}s")
    return pipe, (sds if rank == 0 else None)


def cpu_baseline(sds, threads, device_decode=None):
    """Oracle (CPU restatement, kind 'port') on the host cores: ONE whole 512x512 frame through the fp32 loop (pyramid +
    20 DDIM steps with CFG at model batch 2 + VAE decode), timed as it is — no extrapolation.  The same frame decoded by
    the device pipeline gives the PSNR of the metric's "vs ref" half.  Reported next to the GPU number; never part of `value`."""
    from diffcodec_amd import selftest as T, weights as W
    from diffcodec_amd.synthetic import synth_controls, synth_latents, synth_text
    from oracle import pipeline_ref as R
    torch.set_num_threads(threads)
    usd, csd, vsd = sds
    cond, flow = synth_controls(1, SIZE)
    pe, npe = synth_text(1)
    lat = synth_latents(1, SIZE)
    kw = dict(num_inference_steps=STEPS_DDIM, guidance_scale=4.5, controlnet_conditioning_scale=1.7)
    t0 = time.time()
    ref = R.decode_frame(usd, csd, vsd, W.SD15_UNET_CONFIG, W.SD15_VAE_CONFIG, cond, flow, pe, npe, lat, **kw)
    frame_s = time.time() - t0
    assert torch.isfinite(ref).all()
    out = dict(value=1.0 / frame_s, unit="frames/s", cores=threads, kind="port",
               sample=f"one whole frame: pyramid + 20 DDIM steps (CFG, ControlNet+UNet, batch 2) + VAE decode at 512x512 fp32, "
                      f"{frame_s:.1f}s measured (not extrapolated)")
    if device_decode is not None:
        img = device_decode(dict(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=cond, flow_cond=flow, latents=lat), kw)
        out["psnr_db_device_vs_cpu"] = round(T.psnr(img.float().cpu(), ref), 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16, help="inter frames decoded per step per GPU (1 = single-frame latency config)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=1, help="denoising steps captured per hipGraph")
    ap.add_argument("--dual-stream", type=int, default=1, help="1: ControlNet and UNet down path on two HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-shape igemm timing leg (profiling runs)")
    args = ap.parse_args()

    from diffcodec_amd import ops, sharding
    from diffcodec_amd.synthetic import synth_controls, synth_latents, synth_text
    rank, local, world = sharding.init_from_env()
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("DC_FORCE_DEVICE") is not None:      # rehearsal of the N>1 path on a 1-GPU box (with DC_DIST_BACKEND=gloo)
        local = int(os.environ["DC_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist

    pipe, sds = build_pipeline(rank, device)
    if world > 1:
        t0 = time.time()
        nbytes = sharding.broadcast_params(sharding.module_param_tensors(pipe.unet, pipe.controlnet, pipe.vae))
        torch.cuda.synchronize()
        log(f"[rank {rank}] weight broadcast {nbytes / 1e9:.2f} GB in {time.time() - t0:.2f}s")
    pipe.enable_hip_graphs(not args.no_graphs, steps_per_graph=args.steps_per_graph)
    pipe.enable_dual_stream(bool(args.dual_stream))

    F = args.frames
    # two alternating input sets (distinct tensors) so that no per-call cache can carry work across steps
    sets = []
    for s in range(2):
        cond, flow = synth_controls(F, SIZE, seed=1234 + 17 * s + 1000 * rank)
        pe, npe = synth_text(F, seed=77 + s)
        lat = synth_latents(F, SIZE, seed=4321 + s + 1000 * rank)
        sets.append(dict(controlnet_cond=cond.to(device), flow_cond=flow.to(device), prompt_embeds=pe.to(device),
                         negative_prompt_embeds=npe.to(device), latents=lat.to(device)))
    kw = dict(num_inference_steps=STEPS_DDIM, guidance_scale=4.5, controlnet_conditioning_scale=1.7, output_type="pt")

    def one_step(i):
        return pipe(**sets[i % 2], **kw).images

    for i in range(args.warmup):
        tw = time.perf_counter()
        out = one_step(i)
        torch.cuda.synchronize()
        log(f"[rank {rank}] warmup step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all() and out.shape == (F, 3, SIZE, SIZE)
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    # ---- single-frame latency configuration (BASELINE configs[1] read literally: one 512x512 frame per pass), reported
    #      beside `value`; measured after the timed region, never part of it
    single = None
    if rank == 0 and F != 1 and not args.no_roofline:
        one = [{k: v[:1].contiguous() for k, v in s_.items()} for s_ in sets]
        for i in range(2):
            pipe(**one[i % 2], **kw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(3):
            pipe(**one[i % 2], **kw)
        torch.cuda.synchronize()
        single = dict(frames_per_step=1, ms_per_frame=round((time.perf_counter() - t1) / 3 * 1e3, 2))
        single["frames_per_s"] = round(1e3 / single["ms_per_frame"], 3)

    # ---- U-Net forward alone (north_star's target is quoted on it): model batch 2F exactly as inside a denoising step
    #      (text K/V cached, CFG halves sharing their common prefix, ControlNet residuals added), graph-replayed, after the
    #      timed region.  FLOPs = the reference's algorithmic work, 803.3 GFLOP per sample-forward.
    unet_fwd = None
    if rank == 0 and not args.no_roofline:
        one_step(0)                                            # leaves contexts / controls / buffers of the F-frame batch in place
        st = pipe._state
        ttab = torch.full((STEPS_DDIM,), 500.0, device=device)
        stepc = torch.zeros(1, device=device, dtype=torch.int32)
        down, mid = pipe.controlnet.forward_nhwc(st["x_in"], ttab, 1.7, step_dev=stepc, cfg_shared=True)
        fwd = lambda: pipe.unet.forward_nhwc(st["x_in"], ttab, down, mid, step_dev=stepc, cfg_shared=True)
        fwd()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fwd()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = 0.8033 * 2 * F / (ms * 1e-3)
        unet_fwd = dict(model_batch=2 * F, ms=round(ms, 3), tflops=round(tf, 1), frac_of_bf16_peak=round(tf / PEAK_BF16_TFLOPS, 4),
                        note="U-Net forward only, 803.3 GFLOP of reference work per sample (all kernels: GEMM, attention, norms)")
        del gr, down, mid

    # ---- roofline leg (after the timed region): HIP events around every MFMA implicit-GEMM launch of one eager frame
    roof = None
    if rank == 0 and not args.no_roofline:
        # An eager frame is host-launch-bound (events would time the gaps), so: record every igemm launch of one eager
        # frame, then time each DISTINCT launch shape back-to-back (10 launches between two HIP events on the launch
        # stream) and weight by its count in the frame.
        pipe.enable_hip_graphs(False)
        ops.PROFILE = {}
        one_step(0)
        torch.cuda.synchronize()
        uniq, ops.PROFILE = ops.PROFILE, None
        n_launches = sum(u[0] for u in uniq.values())
        tot_ms = tot_fl = 0.0
        rows = []
        for label, (cnt, flops, relaunch) in uniq.items():
            for _ in range(3):
                relaunch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                relaunch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tot_ms += cnt * ms
            tot_fl += cnt * flops
            rows.append((cnt * ms, cnt, ms * 1e3, flops / (ms * 1e-3) / 1e12, label))
        rows.sort(reverse=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "igemm_shapes.txt"), "w") as fh:
            fh.write("total_ms_per_frame count us_per_launch TFLOP/s shape\n")
            for r in rows:
                fh.write(f"{r[0]:9.3f} {r[1]:5d} {r[2]:9.2f} {r[3]:8.1f} {r[4]}\n")
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        traffic, traffic_note = None, "no PMC summary committed"
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_igemm.json")
        if os.path.exists(pmc):       # PMC counters cannot be read from inside the process: committed rocprofv3 --pmc passes
            with open(pmc) as fh:
                sh = json.load(fh)["shapes"][0]
            traffic = sh["hbm_bytes"]
            traffic_note = (f"HBM bytes per launch of {sh['shape']}: 2*FETCH_SIZE+WRITE_SIZE from separate rocprofv3 --pmc passes "
                            f"(profiles/r01_pmc_igemm.json), {sh['ratio']}x its algorithmic {sh['algorithmic_bytes']} B")
        roof = dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4),
                    traffic=traffic, traffic_note=traffic_note,
                    kernel="dc_conv_igemm_bf16 family: conv3x3_tile_kernel / gemm_dma_kernel / igemm_kernel", launches_per_step=n_launches,
                    avg_launch_us=round(tot_ms * 1e3 / max(1, n_launches), 2), igemm_ms_per_step=round(tot_ms, 2),
                    note="sum of algorithmic 2*M*N*K over the igemm launches of one frame / sum of their HIP-event launch durations "
                         "(each distinct launch shape timed back-to-back x10 on the launch stream, weighted by its count)")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # host-core baseline: single-GPU runs only
        dev_decode = lambda inp, kw_: pipe(**{k: v.to(device) for k, v in inp.items()}, output_type="pt", **kw_).images
        cpu = cpu_baseline(sds, threads=max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("DC_CPU_THREADS", "16")))),
                           device_decode=dev_decode)

    if rank == 0:
        frames = F * world * args.steps
        fps = frames / dt
        line = {
            "metric": "decoded frames/sec @ 512x512, 20-step DDIM, GOP-12",
            "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"512x512 inter frames, {F} per step per GPU, 20-step DDIM, CFG 4.5 (model batch {2 * F}), "
                                   f"control scale 1.7, SD-1.5 + DualFlowControlNet + VAE decode, random-init weights",
                       "frames_per_step_per_gpu": F, "hip_graphs": not args.no_graphs, "parallelism": f"frame-shard x{world}"},
            "frame_tflop_algorithmic": round(TFLOP_PER_FRAME, 2),
            "frame_mfma_frac": round(fps / world * TFLOP_PER_FRAME / PEAK_BF16_TFLOPS, 4),
            "single_frame": single, "unet_forward": unet_fwd, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
