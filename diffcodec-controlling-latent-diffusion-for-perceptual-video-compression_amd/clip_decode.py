"""GOP / tile decode driver: a clip -> decode units -> shard over ranks -> batched decode -> (optional) gather + blend.

What it mirrors in the reference: the per-frame harness of validation.py:85-146 (for every inter frame: its two intra
anchors + its forward / backward flow file -> `load_controls_and_flows` -> `pipe(...)`), the intra/inter split of
uvc_codec_eval.py:19-26 (`get_inter_frames`: every gop_size-th frame is intra) and the data layout both read:

    <root>/<video>/images/frame_%04d.png
    <root>/<video>/optical_flow/optical_flow_gop_<G>_raft/flow_<prev>_<cur>.flo        (forward:  prev anchor -> frame)
    <root>/<video>/optical_flow_bwd/optical_flow_gop_<G>_raft/flow_<next>_<cur>.flo    (backward: next anchor -> frame)

BASELINE configs 3-5 are this driver with different parameters:
    C3  GOP-12, 512x512, unit = inter frame, units round-robin over 8 ranks;
    C4  GOP-4, 960x512 as two 512x512 windows per frame, pipe built with [DualFlowControlNet, ResControlNet] + warp_cond;
    C5  GOP-12, 1080p tiled, 50 steps, unit = (frame, tile) sharded per tile (`shard="tile"`).

Units are independent: there is NO data-path collective.  The only collective here is the optional
gather of decoded units onto one rank (`gather=True`), needed when the tiles of one frame were decoded on different ranks
and must be blended.  What is gathered are the units as uint8 IMAGES (0.79 MB per 512x512 unit, one padded tensor
`dist.gather`, no pickling): the reference's notebook also blends the tiles as the 8-bit images `pipe(...)` returned
(patch_exp.ipynb cell 7), so a unit is quantised before the blend on every path and a sharded decode equals the
single-rank one bit for bit.  Trailing-frame policy: an inter frame needs a closing anchor, so frames after the last intra frame
are not decodable bidirectionally; `sharding.gop_inter_frames` drops them (the reference's `get_inter_frames` merely
counts them for metrics).  `trailing_frames()` reports which frames were dropped."""
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import sharding
from .tiled_decode import plan_tiles


@dataclass(frozen=True)
class DecodeUnit:
    uid: int                 # position in the clip's unit list (what `shard_units` deals)
    frame: int               # inter frame index
    prev: int                # previous intra anchor
    next: int                # next intra anchor
    window: Tuple[int, int, int, int]   # (y1, y2, x1, x2) in frame pixels; the whole frame when it is one tile
    tile: int                # index of the window within its frame


def plan_units(num_frames, gop_size, height, width, tile=512, overlap=64) -> List[DecodeUnit]:
    """All decode units of a clip, frame-major: inter frames (uvc_codec_eval.py:19-26) x full-size windows (tiled_decode)."""
    wins = [(0, height, 0, width)] if (height, width) == (tile, tile) else plan_tiles(height, width, tile, overlap)
    units = []
    for f, p, n in sharding.gop_inter_frames(num_frames, gop_size):
        for ti, w in enumerate(wins):
            units.append(DecodeUnit(len(units), f, p, n, tuple(w), ti))
    return units


def trailing_frames(num_frames, gop_size):
    """Inter frames without a closing anchor (dropped; see module docstring)."""
    last_intra = ((num_frames - 1) // gop_size) * gop_size
    return [f for f in range(last_intra + 1, num_frames)]


def shard(units, rank, world, mode="unit"):
    """mode 'unit': round-robin over units (C3, C5 per-patch shard); 'frame': all windows of a frame on one rank, frames
    round-robin (blend without a gather)."""
    if mode == "unit":
        return [units[i] for i in sharding.shard_units(len(units), rank, world)]
    if mode != "frame":
        raise ValueError("mode must be 'unit' or 'frame'")
    frames = sorted({u.frame for u in units})
    mine = {frames[i] for i in sharding.shard_units(len(frames), rank, world)}
    return [u for u in units if u.frame in mine]


class DirectorySource:
    """Controls of one inter frame from the reference's on-disk layout (validation.py:85-93)."""

    def __init__(self, root, video, gop_size, size, device="cuda", flow_dir="optical_flow", flow_bwd_dir="optical_flow_bwd"):
        self.root, self.video, self.gop, self.size, self.device = root, video, gop_size, tuple(size), device
        self.flow_dir, self.flow_bwd_dir = flow_dir, flow_bwd_dir

    def paths(self, frame, prev, nxt):
        base = os.path.join(self.root, self.video)
        sub = f"optical_flow_gop_{self.gop}_raft"
        return (os.path.join(base, "images", f"frame_{prev:04d}.png"), os.path.join(base, "images", f"frame_{nxt:04d}.png"),
                os.path.join(base, self.flow_dir, sub, f"flow_{prev:04d}_{frame:04d}.flo"),
                os.path.join(base, self.flow_bwd_dir, sub, f"flow_{nxt:04d}_{frame:04d}.flo"))

    def controls(self, frame, prev, nxt):
        from .io_utils import load_controls_and_flows
        return load_controls_and_flows(*self.paths(frame, prev, nxt), size=self.size, device=self.device)


class SyntheticSource:
    """Seeded synthetic controls of the shapes the reference feeds `pipe(...)`; frame f of every rank
    sees the same tensors, so sharded and unsharded decodes of a clip are comparable."""

    def __init__(self, height, width, device="cuda", seed=1234, with_warp=False):
        self.h, self.w, self.device, self.seed, self.with_warp = height, width, device, seed, with_warp

    def controls(self, frame, prev, nxt):
        from .synthetic import synth_controls
        s = max(self.h, self.w)
        cond, flow = synth_controls(1, s, seed=self.seed + 7919 * frame)
        return cond[:, :, :self.h, :self.w].contiguous().to(self.device), flow[:, :, :self.h, :self.w].contiguous().to(self.device)

    def warp(self, frame):
        g = torch.Generator().manual_seed(self.seed + 104729 * frame + 1)
        return torch.rand(1, 3, self.h, self.w, generator=g).to(self.device)


class ResidentSource:
    """Controls already resident in HBM (bench.py: inputs are on the device when the timed region starts): wraps another
    source and keeps what it returned, per frame."""

    def __init__(self, source, frames_prev_next, noise=None):
        """noise: optional {frame: latents [1,4,H/8,W/8] on the device} (otherwise `frame_noise` draws per call on the host)."""
        self.with_warp = getattr(source, "with_warp", False)
        self._c = {f: source.controls(f, p, n) for f, p, n in frames_prev_next}
        self._w = {f: source.warp(f) for f, _, _ in frames_prev_next} if self.with_warp else {}
        self._n = noise

    def noise(self, frame, height, width, seed):
        return None if self._n is None else self._n[frame]

    def controls(self, frame, prev, nxt):
        return self._c[frame]

    def warp(self, frame):
        return self._w[frame]


def frame_noise(frame, height, width, seed, channels=4):
    """Initial latents of a frame [1,4,H/8,W/8]: a CPU generator seeded per frame (pipeline.py:269-278 draws on the
    generator's device), so the result does not depend on which rank / batch decodes the frame."""
    g = torch.Generator().manual_seed(int(seed) * 1000003 + int(frame))
    return torch.randn((1, channels, height // 8, width // 8), generator=g)


@torch.no_grad()
def decode_units(pipe, units, source, prompt_embeds, negative_prompt_embeds=None, *, batch=16, seed=0, frame_size=None,
                 output="pt", unit_hw=(512, 512), **pipe_kwargs):
    """Decode `units` (this rank's share) `batch` at a time.  Returns fp32 [len(units), 3, th, tw] in [0,1] on the device
    (output='pt') or the latents (output='latent').  Controls of a frame are loaded once and cropped per window; flows keep
    frame units (patch_exp.ipynb does not re-base them).  An empty share (more ranks than units) returns an empty tensor
    of the same trailing shape, so callers can concatenate / gather rank outputs without a special case."""
    if not units:
        th, tw = unit_hw
        shape = (0, 4, th // 8, tw // 8) if output == "latent" else (0, 3, th, tw)
        return torch.empty(shape, device=pipe.device, dtype=torch.float32)
    for name, t in (("prompt_embeds", prompt_embeds), ("negative_prompt_embeds", negative_prompt_embeds)):
        if t is not None and t.shape[0] != 1 and t.shape[0] < min(batch, len(units)):
            raise ValueError(f"{name} has batch {t.shape[0]}: pass one row (shared by every unit) or at least one per unit "
                             f"of a chunk ({min(batch, len(units))})")
    cache = {}

    def frame_inputs(u):
        if u.frame not in cache:
            cache.clear()                                # units are frame-major: one frame's controls live at a time
            cond, flow = source.controls(u.frame, u.prev, u.next)
            h, w = (cond.shape[-2], cond.shape[-1]) if frame_size is None else frame_size
            warp = source.warp(u.frame) if getattr(source, "with_warp", False) else None
            nz = source.noise(u.frame, h, w, seed) if hasattr(source, "noise") else None
            cache[u.frame] = (cond, flow, frame_noise(u.frame, h, w, seed) if nz is None else nz, warp)
        return cache[u.frame]

    outs = []
    for i in range(0, len(units), batch):
        chunk = units[i:i + batch]
        cc, fc, lt, wc = [], [], [], []
        for u in chunk:
            cond, flow, noise, warp = frame_inputs(u)
            y1, y2, x1, x2 = u.window
            cc.append(cond[:, :, y1:y2, x1:x2])
            fc.append(flow[:, :, y1:y2, x1:x2])
            lt.append(noise[:, :, y1 // 8:y2 // 8, x1 // 8:x2 // 8])
            if warp is not None:
                wc.append(warp[:, :, y1:y2, x1:x2])
        n = len(chunk)
        pe = prompt_embeds.expand(n, -1, -1).contiguous() if prompt_embeds.shape[0] == 1 else prompt_embeds[:n]
        npe = None if negative_prompt_embeds is None else (
            negative_prompt_embeds.expand(n, -1, -1).contiguous() if negative_prompt_embeds.shape[0] == 1 else negative_prompt_embeds[:n])
        extra = dict(warp_cond=torch.cat(wc, 0).contiguous()) if wc else {}
        res = pipe(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=torch.cat(cc, 0).contiguous(),
                   flow_cond=torch.cat(fc, 0).contiguous(), latents=torch.cat(lt, 0).contiguous(),
                   output_type="latent" if output == "latent" else "pt", **extra, **pipe_kwargs).images
        outs.append(res.float())
    return torch.cat(outs, 0).contiguous()


def units_to_u8(images):
    """fp32 unit images [n,3,th,tw] in [0,1] -> uint8 [n,th,tw,3]: `image_processor.postprocess(output_type='pil')`'s
    quantisation (pipeline.py:397-398: x * 255, round, uint8), applied to the tensor on whatever device it lives."""
    return (images.permute(0, 2, 3, 1) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()


def gather_units(local_u8, units, rank=None, world=None, mode="unit", dst=0):
    """Collect this rank's uint8 unit images [n_local, th, tw, 3] on `dst` as [len(units), th, tw, 3] in unit order.  Every
    rank can list every other rank's share (`shard` is a pure function), so only pixels travel: ONE `dist.gather` of
    tensors padded to the largest share — device tensors over RCCL, host tensors over gloo (which gathers on the CPU
    only).  Returns None on the other ranks."""
    import torch.distributed as dist
    ini = dist.is_available() and dist.is_initialized()
    if rank is None or world is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if ini else (0, 1)
    shares = [[u.uid for u in shard(units, r, world, mode)] for r in range(world)]
    if local_u8.dtype != torch.uint8 or local_u8.shape[0] != len(shares[rank]):
        raise ValueError("gather_units takes this rank's uint8 unit images, one per unit of its share")
    if world == 1 or not ini:
        out = torch.empty((len(units),) + tuple(local_u8.shape[1:]), dtype=torch.uint8, device=local_u8.device)
        out[shares[rank]] = local_u8
        return out
    send = local_u8 if dist.get_backend() != "gloo" else local_u8.cpu()
    per = max(len(s_) for s_ in shares)
    pad = torch.zeros((per,) + tuple(send.shape[1:]), dtype=torch.uint8, device=send.device)
    pad[:send.shape[0]] = send
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((len(units),) + tuple(pad.shape[1:]), dtype=torch.uint8, device=pad.device)
    for r in range(world):
        if shares[r]:
            out[shares[r]] = bufs[r][:len(shares[r])]
    return out


def blend_frames(unit_u8, units, height, width, overlap=64):
    """uint8 unit images [n,th,tw,3] of COMPLETE frames -> {frame: uint8 [H,W,3]} (device blend kernel when the tensor
    lives on the GPU, `tiling.merge_ramp` on the host otherwise; both blend the 8-bit tiles in fp32 with the same op
    order, so the two are bit-identical)."""
    from . import ops, tiling
    if unit_u8.dtype != torch.uint8:
        raise ValueError("blend_frames takes uint8 unit images (units_to_u8)")
    frames = {}
    by_frame = {}
    for k, u in enumerate(units):
        by_frame.setdefault(u.frame, []).append((u.tile, k, u.window))
    for f, lst in by_frame.items():
        lst.sort()
        idx = [k for _, k, _ in lst]
        coords = [w for _, _, w in lst]
        tiles = unit_u8[idx]
        if len(lst) == 1 and coords[0] == (0, height, 0, width):
            frames[f] = tiles[0].cpu().numpy()
        elif tiles.is_cuda:
            frames[f] = ops.blend_tiles_ramp(tiles.permute(0, 3, 1, 2).float().contiguous(), coords, (height, width), overlap,
                                             scale=1.0).cpu().numpy()
        else:
            frames[f] = tiling.merge_ramp([t.numpy() for t in tiles], coords, (height, width), order="hwc", feather=overlap)
    return frames


@torch.no_grad()
def decode_clip(pipe, source, num_frames, gop_size, height, width, prompt_embeds, negative_prompt_embeds=None, *, tile=512,
                overlap=64, batch=16, seed=0, rank=None, world=None, shard_mode="unit", gather=True, **pipe_kwargs):
    """Whole pipeline for one clip on this rank.  Returns dict(units=all units, mine=this rank's, images=this rank's fp32 unit
    images, frames={frame: uint8 HxWx3} on the gathering rank (or for locally complete frames when gather=False))."""
    if rank is None or world is None:
        ini = torch.distributed.is_available() and torch.distributed.is_initialized()
        rank = torch.distributed.get_rank() if ini else 0
        world = torch.distributed.get_world_size() if ini else 1
    units = plan_units(num_frames, gop_size, height, width, tile, overlap)
    mine = shard(units, rank, world, shard_mode)
    unit_hw = (min(height, tile), min(width, tile))            # a frame smaller than the tile is one frame-sized unit (ranks with an empty
    #                                                            share must pad their gather buffers to the SAME unit size as the others)
    images = decode_units(pipe, mine, source, prompt_embeds, negative_prompt_embeds, batch=batch, seed=seed,
                          frame_size=(height, width), unit_hw=unit_hw, **pipe_kwargs)
    u8 = units_to_u8(images)
    frames = None
    if gather and world > 1:
        allimg = gather_units(u8, units, rank, world, shard_mode, dst=0)
        if rank == 0:
            frames = blend_frames(allimg, units, height, width, overlap)
    else:
        per_frame = {}
        for u in units:
            per_frame[u.frame] = per_frame.get(u.frame, 0) + 1
        have = {}
        for u in mine:
            have[u.frame] = have.get(u.frame, 0) + 1
        complete = [k for k, u in enumerate(mine) if have[u.frame] == per_frame[u.frame]]
        if complete:
            frames = blend_frames(u8[complete], [mine[k] for k in complete], height, width, overlap)
    return dict(units=units, mine=mine, images=images, frames=frames)
