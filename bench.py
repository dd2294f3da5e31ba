#!/usr/bin/env python3
"""bench.py — decoded frames/s of the DiffCodec decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--config c2|c4] [--no-graphs]

A "step" is one pass of the hot path over one batch of F synthetic inter frames resident in HBM, driven exactly as a clip is
decoded (`clip_decode`): GOP-12 clip -> decode units -> round-robin shard over the ranks -> one batched `pipe(...)` call per
rank: control pyramid + FDN gamma/beta (once per frame), 20 DDIM steps of {DualFlowControlNet, UNet} with CFG (model batch
2F), CFG+DDIM update, VAE decode, postprocess.  SD-1.5 topology, random-init weights in the diffusers key layout (no
checkpoint is reachable offline), bf16 compute with fp32 accumulation.
  --config c2 (default)  BASELINE configs[1]: 512x512 frames, one DualFlowControlNet (the configuration `metric` is quoted on)
  --config c4            BASELINE configs[3]: 960x512 frames = two 512x512 windows each, GOP-4, DualFlowControlNet +
                         ResControlNet (warp_cond); frames/s counts whole 960x512 frames
  (the line also carries `validation_config`: what validation.py:37,106,132-146 runs — UniPC multistep, 40 steps, FreeU, CFG,
   a string prompt through the text tower — timed after the headline region)
N > 1: one process per GPU (torch.distributed / RCCL); units are sharded across ranks with no data-path collective; rank 0
synthesises the weights and broadcasts the packed tensors once (outside the timed region).  Prints ONE JSON line on rank 0.
`python bench.py --gpus N` with no launcher around it starts its own N ranks (child processes, before this process makes any
GPU call); under `torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.  The timed step is the decode of this rank's
units only — decoded frames stay on the rank that decoded them; the optional uint8 gather is timed separately (`gather_ms`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = None          # imported in main(), after the self-launch decision: the launching parent never touches the GPU

STEPS_DDIM = 20
SIZE = 512
# algorithmic work per decoded 512x512 frame (CFG on, step-invariant parts hoisted): TFLOP
TFLOP_PER_FRAME = (20 * 2 * (803.3 + 268.6) + 30.2 + 19.9 + 2514.5) / 1000.0
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0          # HBM3E spec (≈6300 achievable), same guide


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_pipeline(rank, device, dual=False):
    from diffcodec_amd import weights as W
    from diffcodec_amd.controlnet import HipDualFlowControlNet
    from diffcodec_amd.pipeline import StableDiffusionDualFlowControlNetPipeline
    from diffcodec_amd.rescontrolnet import HipResControlNet
    from diffcodec_amd.scheduler import DDIMScheduler
    from diffcodec_amd.unet import HipUNet2DConditionModel
    from diffcodec_amd.vae import HipAutoencoderKL
    t0 = time.time()
    specs = [W.unet_spec(), W.controlnet_spec(), W.vae_spec()] + ([W.rescontrolnet_spec()] if dual else [])
    if rank == 0:
        sds = [W.synthesize(s, seed=i) for i, s in enumerate(specs)]
    else:   # shapes only; the packed device tensors are overwritten by the broadcast from rank 0
        sds = [{k: torch.empty(shape) for k, (_, shape) in s.items()} for s in specs]
    log(f"[rank {rank}] state dicts ready in {time.time() - t0:.1f}s")
    unet = HipUNet2DConditionModel(sds[0], W.SD15_UNET_CONFIG, device)
    cn = HipDualFlowControlNet(sds[1], W.SD15_UNET_CONFIG, device)
    vae = HipAutoencoderKL(sds[2], W.SD15_VAE_CONFIG, device)
    nets = [cn, HipResControlNet(sds[3], W.SD15_UNET_CONFIG, device)] if dual else cn
    pipe = StableDiffusionDualFlowControlNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, controlnet=nets,
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
    usd, csd, vsd = sds[:3]
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


# PMC reference shape per kernel family (tools/pmc_conv.py launches exactly these; algorithmic bytes = inputs + weights + output)
# value: (shape label, algorithmic bytes, launch grid in threads of that shape — what tells a family's PMC rows apart)
# keys: (prefix the in-situ family label must START with, substring of the rocprofv3 kernel name).  The 1x1 family aggregates three
# kernels; its PMC reference shape is a gemm_rowpanel_kernel launch, so its rows are looked up under that kernel's name.
PMC_SHAPES = {("gemm_dma_kernel", "gemm_rowpanel_kernel"): ("1x1 n=32 64x64 320->320 (M=131072 N=320 K=320; the gemm_rowpanel_kernel launch of the gemm_dma + gemm_wide + gemm_p8 + gemm_rowpanel family, 512 panels x 512 threads)", 167976960, 262144),
              ("conv3x3_tile_kernel", "conv3x3_tile_kernel"): ("3x3 n=32 64x64 320->320 (M=131072 N=320 K=2880)", 169615360, 524288),
              ("attn_kernel", "attn_kernel"): ("attention B=32 H=8 N=4096 d=40", 335544320, 1048576)}
PMC_FILES = ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")


def latest_pmc(family):
    """HBM traffic per launch of the family's PMC reference shape, from the committed rocprofv3 --pmc passes (hardware counters
    cannot be read in-process): (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per the gfx950 note of the guide."""
    key = next((k for k in PMC_SHAPES if family.startswith(k[0])), None)
    for name in PMC_FILES:
        p = os.path.join(ROOT, "profiles", name)
        if key is None or not os.path.exists(p):
            continue
        with open(p) as fh:
            summ = json.load(fh)
        shape, alg, grid = PMC_SHAPES[key]
        rows = [v for k, v in summ.items() if key[1] in k and k.endswith(f"grid={grid}") and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
        if not rows:
            continue
        r = rows[0]                                  # the launch group whose grid is the reference shape's
        hbm = int((2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024)
        return hbm, (f"HBM bytes per launch of {shape}: (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes "
                     f"(profiles/{name}) = {hbm / alg:.3f}x its algorithmic {alg} B; `achieved` is the family aggregate over all of "
                     f"its shapes in one step")
    return None, "no PMC summary committed for this kernel family"


def self_launch(n):
    """`--gpus n` without a launcher: n ranks as child processes of this one (the shape of train_control.sh:19's
    `accelerate launch --num_processes 8`).  Called before torch is imported here, so the parent never initialises the GPU;
    children are fresh interpreters (no exec of a GPU-touching process).  Rank 0's stdout (the JSON line) is this process's
    stdout; the exit code is non-zero if any rank fails, and the surviving ranks of a failed job are terminated by PID."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                log(f"rank {procs.index(p)} exited with code {code}: stopping the other ranks")
                for q in alive:
                    q.terminate()
        time.sleep(0.2)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None, help="decode units (512x512 windows) per step per GPU; default 16 "
                    "(c4: 8 frames = 16 windows); 1 = single-frame latency configuration")
    ap.add_argument("--config", choices=("c2", "c4"), default="c2")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=1, help="denoising steps captured per hipGraph")
    ap.add_argument("--dual-stream", type=int, default=1, help="1: ControlNet and UNet down path on two HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the post-timing legs (profiling runs)")
    ap.add_argument("--c3-batch", type=int, default=44, help="units per pipe call of the strong-scaling C3 pass")
    ap.add_argument("--shapes-out", default=None, help="write the in-situ per-shape launch table of the roofline leg to this file")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                                  # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ.get('WORLD_SIZE')}: refusing to report "
                 f"a {os.environ.get('WORLD_SIZE', 1)}-rank run as an {args.gpus}-GPU number")
    global torch
    import torch

    from diffcodec_amd import clip_decode as CD, lib, sharding
    from diffcodec_amd.synthetic import synth_text
    rank, local, world = sharding.init_from_env()
    if os.environ.get("DC_FORCE_DEVICE") is not None:      # rehearsal of the N>1 path on a 1-GPU box (with DC_DIST_BACKEND=gloo)
        local = int(os.environ["DC_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    backend_world = dist.get_world_size() if dist.is_initialized() else 1
    backend_name = dist.get_backend() if dist.is_initialized() else "none"

    c4 = args.config == "c4"
    pipe, sds = build_pipeline(rank, device, dual=c4)
    bcast = None
    if world > 1:
        t0 = time.time()
        mods = [pipe.unet, pipe.vae] + (list(pipe.controlnet) if c4 else [pipe.controlnet])
        nbytes = sharding.broadcast_params(sharding.module_param_tensors(*mods))
        torch.cuda.synchronize()
        bcast = dict(gigabytes=round(nbytes / 1e9, 3), seconds=round(time.time() - t0, 3), note="one-off weight broadcast from rank 0 in 256 MB buckets, outside the timed region")
        log(f"[rank {rank}] weight broadcast {nbytes / 1e9:.2f} GB in {time.time() - t0:.2f}s over {backend_name} (world {backend_world})")
    pipe.enable_hip_graphs(not args.no_graphs, steps_per_graph=args.steps_per_graph)
    pipe.enable_dual_stream(bool(args.dual_stream))

    # ---- the clip: units -> shard -> resident inputs.  c2: GOP-12, one 512x512 window per frame; c4: GOP-4, 960x512 = 2 windows
    F = args.frames if args.frames is not None else 16
    height, width, gop = (SIZE, 960, 4) if c4 else (SIZE, SIZE, 12)
    tiles_per_frame = len(CD.plan_units(gop + 1, gop, height, width)) // (gop - 1)
    frames_needed = -(-F * world // tiles_per_frame)
    num_frames = 1 + -(-frames_needed // (gop - 1)) * gop
    all_units = CD.plan_units(num_frames, gop, height, width)[:F * world]
    mine = CD.shard(all_units, rank, world)
    assert len(mine) == F
    pe, npe = (t.to(device) for t in synth_text(1))
    kw = dict(num_inference_steps=STEPS_DDIM, guidance_scale=4.5,
              controlnet_conditioning_scale=[1.7, 1.0] if c4 else 1.7)
    sources = []                     # two alternating input sets (distinct tensors): no per-call cache carries work across steps
    for s in range(2):
        src = CD.SyntheticSource(height, width, device=device, seed=1234 + 17 * s, with_warp=c4)
        fpn = sorted({(u.frame, u.prev, u.next) for u in mine})
        noise = {f: CD.frame_noise(f, height, width, 4321 + s).to(device) for f, _, _ in fpn}
        sources.append(CD.ResidentSource(src, fpn, noise=noise))

    def one_step(i, units=None):
        u = mine if units is None else units
        return CD.decode_units(pipe, u, sources[i % 2], pe, npe, batch=len(u), frame_size=(height, width), **kw)

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
    rank_ms = None
    if world > 1:
        per = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per, torch.tensor([dt], device=device, dtype=torch.float64))
        per = [p.item() for p in per]
        dt = max(per)                                            # the contract's MAX over ranks
        rank_ms = dict(min=round(min(per) / args.steps * 1e3, 3), max=round(max(per) / args.steps * 1e3, 3),
                       per_rank=[round(p / args.steps * 1e3, 3) for p in per])

    # ---- the optional exchange, timed on its own (never part of `value`): uint8 gather of one step's units onto rank 0
    gather_ms = None
    if world > 1:
        u8 = CD.units_to_u8(out)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        allu = CD.gather_units(u8, all_units, rank, world)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = round((time.perf_counter() - t1) * 1e3, 3)
        assert (allu is not None and allu.shape[0] == F * world) if rank == 0 else allu is None
        del allu, u8

    # ---- BASELINE configs[2] as a STRONG-scaling job (every rank takes part): one 97-frame GOP-12 clip = 88 inter-frame units
    #      dealt round-robin over the ranks (11 per rank at 8 ranks), decoded in batches of at most --c3-batch units
    c3 = None
    if not c4 and not args.no_roofline and args.config == "c2":
        units3 = CD.plan_units(97, 12, SIZE, SIZE)
        mine3 = CD.shard(units3, rank, world)
        b3 = max(1, min(args.c3_batch, len(mine3)))
        while len(mine3) % b3:                                   # equal batches: one captured graph shape per rank
            b3 -= 1
        fpn3 = sorted({(u.frame, u.prev, u.next) for u in mine3})
        src3 = CD.ResidentSource(CD.SyntheticSource(SIZE, SIZE, device=device, seed=977), fpn3,
                                 noise={f: CD.frame_noise(f, SIZE, SIZE, 99).to(device) for f, _, _ in fpn3})
        run3 = lambda: CD.decode_units(pipe, mine3, src3, pe, npe, batch=b3, frame_size=(SIZE, SIZE), **kw)
        run3()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        o3 = run3()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        d3 = time.perf_counter() - t1
        assert torch.isfinite(o3).all() and o3.shape[0] == len(mine3)
        if world > 1:
            tm = torch.tensor([d3], device=device, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            d3 = tm.item()
        c3 = dict(scaling="strong", units=len(units3), units_per_rank=len(mine3), batch=b3, seconds=round(d3, 4),
                  frames_per_s=round(len(units3) / d3, 3),
                  note="BASELINE configs[2]: the 88 inter frames of one 97-frame GOP-12 clip at 512x512, sharded round-robin over the ranks")
        del o3, src3

    extras = rank == 0 and not args.no_roofline

    def timed_units(units, reps=3):
        for i in range(2):
            one_step(i, units)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(reps):
            one_step(i, units)
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / reps * 1e3

    # ---- other batch shapes of the same workload, after the timed region: one frame (BASELINE configs[1] read literally) and
    #      one GOP-12 (its 11 inter frames) per pass
    single = gop12 = None
    if extras and not c4:
        if F != 1:
            ms = timed_units(mine[:1])
            single = dict(frames_per_step=1, ms_per_frame=round(ms, 2), frames_per_s=round(1e3 / ms, 3))
        if F >= 11 and F != 11:
            ms = timed_units(mine[:11])
            gop12 = dict(frames_per_step=11, ms_per_step=round(ms, 2), frames_per_s=round(11e3 / ms, 3),
                         note="one GOP-12 = 11 inter frames decoded as one batch")

    # ---- U-Net forward alone (north_star's target is quoted on it): model batch 2F exactly as inside a denoising step
    #      (text K/V cached, CFG halves sharing their common prefix, ControlNet residuals added), graph-replayed, after the
    #      timed region.  FLOPs = the reference's algorithmic work, 803.3 GFLOP per sample-forward.
    unet_fwd = vae_enc = None
    if extras and not c4:
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
        # VAE encode (the latent-init decode variant, train_controlnet.py:1081 / pipeline.ipynb cell 7): 1116.7 GFLOP per frame
        x = torch.rand((min(F, 8), SIZE, SIZE, 3), device=device).mul_(2).sub_(1).to(torch.bfloat16)
        for _ in range(2):
            pipe.vae.encode_moments_nhwc(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            pipe.vae.encode_moments_nhwc(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        vae_enc = dict(frames=x.shape[0], ms=round(ms, 3), frames_per_s=round(x.shape[0] * 1e3 / ms, 2),
                       tflops=round(1.1167 * x.shape[0] / (ms * 1e-3), 1),
                       note="AutoencoderKL.encode 512x512 -> 64x64 moments, 1116.7 GFLOP of reference work per frame")

    # ---- the configuration validation.py:37,106,132-146 actually runs (UniPC multistep, 40 steps, FreeU, CFG, a string prompt
    #      through the text tower), after the timed region and never the headline: same clip driver, same units, the fused /
    #      graphed loop with dc_cfg_unipc_step as its scheduler kernel.  The text tower is HipCLIPTextModel with random-init
    #      weights; the tokenizer's vocabulary files are not reachable offline, so a stand-in maps the words to ids.
    validation = None
    if extras and not c4:
        from diffcodec_amd import weights as W
        from diffcodec_amd.scheduler import UniPCMultistepScheduler
        from diffcodec_amd.text_encoder import HipCLIPTextModel

        class WordHashTokenizer:
            """stand-in for CLIPTokenizer (BOS, one id per word, EOS, EOS padding to 77): the bench needs ids, not English"""
            model_max_length = 77

            def __call__(self, texts, **kw):
                rows = []
                for t in texts:
                    ids = [49406] + [1 + (hash_word(w) % 49000) for w in t.split()][:75] + [49407]
                    rows.append(ids + [49407] * (77 - len(ids)))
                return type("Tok", (), {"input_ids": torch.tensor(rows, dtype=torch.int64)})()

        def hash_word(w):
            h = 2166136261
            for ch in w.encode():
                h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
            return h

        t0 = time.perf_counter()
        clip = HipCLIPTextModel(W.synthesize(W.clip_text_spec(), seed=7), None, device)
        ddim = pipe.scheduler
        pipe.text_encoder, pipe.tokenizer = clip, WordHashTokenizer()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vpe, vnpe = pipe.encode_prompt("a city street at dusk, high quality", device, 1, True, negative_prompt="blurry, low quality")
        torch.cuda.synchronize()
        text_ms = (time.perf_counter() - t1) * 1e3
        pipe.scheduler = UniPCMultistepScheduler()
        pipe.enable_freeu(s1=0.9, s2=0.2, b1=1.2, b2=1.4)
        vkw = dict(kw, num_inference_steps=40, guidance_scale=7.5)
        try:
            vstep = lambda i: CD.decode_units(pipe, mine, sources[i % 2], vpe, vnpe, batch=len(mine), frame_size=(height, width), **vkw)
            vstep(0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            vout = vstep(1)
            torch.cuda.synchronize()
            vms = (time.perf_counter() - t1) * 1e3
            assert torch.isfinite(vout).all()
        finally:
            pipe.disable_freeu()
            pipe.scheduler = ddim
            pipe.text_encoder = pipe.tokenizer = None
        validation = dict(scheduler="UniPCMultistepScheduler (bh2, order 2)", steps=40, freeu=dict(s1=0.9, s2=0.2, b1=1.2, b2=1.4),
                          guidance_scale=7.5, frames_per_step=F, ms_per_step=round(vms, 2), frames_per_s=round(F * 1e3 / vms, 3),
                          text_encode_ms=round(text_ms, 2), fused_loop=True, hip_graphs=not args.no_graphs,
                          note="validation.py:37,106,132-146: UniPC + FreeU + CFG 7.5 + string prompt (stand-in tokenizer, random-init "
                               "CLIP tower); 40 denoising steps per frame, so 2x the headline's model work per frame")
        del clip, vout

    # ---- roofline leg (after the timed region): ONE eager step with every C-ABI launch bracketed by two HIP events on the
    #      launch stream (lib.LaunchTimer): each kernel's own duration inside the real step — real operands, real cache
    #      state, host launch gaps excluded.  Agrees with the rocprofv3 --kernel-trace --stats CSV under profiles/.
    roof = families = None
    if extras:
        pipe.enable_hip_graphs(False)
        pipe.enable_dual_stream(False)                          # one stream: kernels do not overlap, durations add up
        one_step(1)
        torch.cuda.synchronize()
        lib.TIMER = lib.LaunchTimer()
        one_step(0)
        torch.cuda.synchronize()
        timer, lib.TIMER = lib.TIMER, None
        summ = timer.summary()
        tot_ms = sum(f["ms"] for f in summ.values())
        families = []
        if args.shapes_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.shapes_out)) or ".", exist_ok=True)
        with open(args.shapes_out or os.devnull, "w") as fh:
            fh.write("# in-situ launch durations of one eager step (HIP events on the launch stream), per kernel family and shape\n")
            for fam, f in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                tfl = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] else 0.0
                gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["ms"] else 0.0
                families.append(dict(kernel=fam, calls=f["calls"], ms_per_step=round(f["ms"], 2), share=round(f["ms"] / tot_ms, 4),
                                     tflops=round(tfl, 1), frac_mfma_peak=round(tfl / PEAK_BF16_TFLOPS, 4), hbm_gbs=round(gbs, 1),
                                     frac_hbm_peak=round(gbs / PEAK_HBM_GBS, 4)))
                fh.write(f"{f['ms']:10.3f} ms {f['calls']:6d} calls {tfl:8.1f} TFLOP/s {gbs:8.1f} GB/s  {fam}\n")
                for shape, (cnt, ms, fl, by) in sorted(f["shapes"].items(), key=lambda kv: -kv[1][1]):
                    if shape:
                        fh.write(f"    {ms:9.3f} ms {cnt:5d} x {ms / cnt * 1e3:9.2f} us {fl / (ms * 1e-3) / 1e12 if ms else 0:8.1f} TFLOP/s "
                                 f"{by / (ms * 1e-3) / 1e9 if ms else 0:8.1f} GB/s  {shape}\n")
        dom = max((f for f in families if f["tflops"] > 50), key=lambda f: f["ms_per_step"])
        d = summ[dom["kernel"]]
        traffic, traffic_note = latest_pmc(dom["kernel"])
        roof = dict(bound="mfma", achieved=dom["tflops"], peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=dom["frac_mfma_peak"],
                    traffic=traffic, traffic_note=traffic_note, kernel=dom["kernel"], launches_per_step=d["calls"],
                    avg_launch_us=round(d["ms"] * 1e3 / d["calls"], 2), kernel_ms_per_step=round(d["ms"], 2),
                    all_kernels_ms_per_step=round(tot_ms, 2),
                    note="dominant kernel family by time: algorithmic 2*M*N*K of its launches in one step / their summed in-situ launch "
                         "durations (HIP events on the launch stream around every launch of one eager, single-stream step)")
        # the HBM-bound kernels north_star asks GB/s for
        for f in families:
            f["bound"] = "mfma" if f["tflops"] > 50 else "hbm"
        pipe.enable_hip_graphs(not args.no_graphs, steps_per_graph=args.steps_per_graph)
        pipe.enable_dual_stream(bool(args.dual_stream))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not c4:      # host-core baseline: single-GPU runs only
        single_pipe = pipe
        dev_decode = lambda inp, kw_: single_pipe(**{k: v.to(device) for k, v in inp.items()}, output_type="pt", **kw_).images
        cpu = cpu_baseline(sds, threads=max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("DC_CPU_THREADS", "16")))),
                           device_decode=dev_decode)

    if rank == 0:
        units_done = F * world * args.steps
        frames = units_done / tiles_per_frame
        fps = frames / dt
        tflop_frame = TFLOP_PER_FRAME if not c4 else None
        line = {
            "metric": "decoded frames/sec @ 512x512, 20-step DDIM, GOP-12" if not c4 else
                      "decoded frames/sec @ 960x512 (2 x 512x512 windows), 20-step DDIM, GOP-4, dual ControlNet",
            "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"512x512 inter frames of a GOP-12 clip, {F} per step per GPU, 20-step DDIM, CFG 4.5 (model batch {2 * F}), "
                                    f"control scale 1.7, SD-1.5 + DualFlowControlNet + VAE decode, random-init weights") if not c4 else
                                   (f"960x512 inter frames of a GOP-4 clip as {F} 512x512 windows per step per GPU, 20-step DDIM, CFG 4.5, "
                                    f"DualFlowControlNet (1.7) + ResControlNet (1.0, warp_cond) + VAE decode, random-init weights"),
                       "frames_per_step_per_gpu": F / tiles_per_frame, "units_per_step_per_gpu": F, "hip_graphs": not args.no_graphs,
                       "parallelism": f"unit-shard x{world}", "backend": backend_name, "backend_world_size": backend_world},
            "frame_tflop_algorithmic": None if c4 else round(TFLOP_PER_FRAME, 2),
            "frame_mfma_frac": None if c4 else round(fps / world * TFLOP_PER_FRAME / PEAK_BF16_TFLOPS, 4),
            "single_frame": single, "gop12_batch": gop12, "c3_clip_strong": c3, "gather_ms": gather_ms,
            "rank_ms_per_step": rank_ms, "weight_broadcast": bcast,
            "unet_forward": unet_fwd, "vae_encode": vae_enc, "validation_config": validation,
            "roofline": roof, "kernel_families": families, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
