"""Tiled decode driver for frames larger than the model's 512x512 (BASELINE configs 4 and 5).

The reference only sketches this in a notebook (patch_exp.ipynb cells 1-7): controls are cut with `crop_into_tiles`
(tile 512, overlap 64), each tile goes through `pipe(...)`, flows are NOT re-based per tile, and only tile 0 is ever
run; its edge tiles are undersized (128 px), which the model cannot take (flownet.py:79 asserts the 64/32/16/8 pyramid).
Policy here (documented deviation): every tile is a full `tile` x `tile` window and the last row/column of windows is
shifted inward to end at the frame border, so overlaps grow at the border instead of tiles shrinking.  Tiles of all
frames are independent decode units: they are batched through the pipeline `batch` at a time (and are what
`sharding.shard_units` deals across GPUs for config 5).  Blending: `tiling.merge_ramp` (half-cosine ramps on inner
edges only; the reference's `merge_costiles` window zeroes lines inside every tile and is kept only for parity)."""
import numpy as np
import torch

from . import tiling


def plan_tiles(height, width, tile=512, overlap=64):
    """Full-size window origins covering [0,height) x [0,width): stride tile-overlap, last window shifted inward."""
    if height < tile or width < tile:
        raise ValueError(f"frame {height}x{width} is smaller than one {tile}x{tile} tile")

    def axis(n):
        stride = tile - overlap
        pos = list(range(0, max(n - tile, 0) + 1, stride))
        if pos[-1] + tile < n:
            pos.append(n - tile)
        return pos

    return [(y, y + tile, x, x + tile) for y in axis(height) for x in axis(width)]


@torch.no_grad()
def decode_tiled(pipe, controlnet_cond, flow_cond, prompt_embeds, negative_prompt_embeds=None, tile=512, overlap=64,
                 batch=8, latents=None, generator=None, feather=None, device_blend=True, **pipe_kwargs):
    """controlnet_cond [1,6,H,W], flow_cond [1,4,H,W] (full frame) -> uint8 image [H,W,3].
    `latents` (optional) is the full-frame noise [1,4,H/8,W/8]; each tile takes its window of it, so overlapping regions
    start from identical noise."""
    _, _, h, w = controlnet_cond.shape
    coords = plan_tiles(h, w, tile, overlap)
    if latents is None:
        gdev = generator.device if isinstance(generator, torch.Generator) else "cpu"
        latents = torch.randn((1, 4, h // 8, w // 8), generator=generator, device=gdev)
    tiles_out = []
    for i in range(0, len(coords), batch):
        chunk = coords[i:i + batch]
        cc = torch.cat([controlnet_cond[:, :, y1:y2, x1:x2] for (y1, y2, x1, x2) in chunk], 0).contiguous()
        fc = torch.cat([flow_cond[:, :, y1:y2, x1:x2] for (y1, y2, x1, x2) in chunk], 0).contiguous()   # flows keep frame units
        lt = torch.cat([latents[:, :, y1 // 8:y2 // 8, x1 // 8:x2 // 8] for (y1, y2, x1, x2) in chunk], 0).contiguous()
        pe = prompt_embeds.expand(len(chunk), -1, -1).contiguous()
        npe = None if negative_prompt_embeds is None else negative_prompt_embeds.expand(len(chunk), -1, -1).contiguous()
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=cc, flow_cond=fc, latents=lt,
                   output_type="pt", **pipe_kwargs).images                                   # [n, 3, tile, tile] fp32 in [0,1]
        tiles_out.append(out.float())
    tiles_dev = torch.cat(tiles_out, 0).contiguous()
    f = overlap if feather is None else feather
    if device_blend and tiles_dev.is_cuda and 2 * f <= tile:
        from . import ops
        return ops.blend_tiles_ramp(tiles_dev, coords, (h, w), f).cpu().numpy(), coords        # one gather kernel, uint8 comes back
    host = [np.asarray(t.permute(1, 2, 0).cpu().numpy() * 255.0, np.float32) for t in tiles_dev]
    return tiling.merge_ramp(host, coords, (h, w), order="hwc", feather=f), coords
