"""Tiled full-resolution decode helpers — SURVEY.md §8 row a22 (config 5), host side: the reference's
patch_utils.py with the same names and semantics.

    crop_into_tiles                        patch_utils.py:189-209   stride = tile - overlap, edge tiles are smaller
    merge_tiles                            patch_utils.py:212-248   flat average of overlaps
    merge_costiles                         patch_utils.py:13-80     raised-cosine feather per tile, weighted average
    merge_latent_tiles_from_pixel_coords   patch_utils.py:83-174    Hann-window blend of latent tiles

Reference quirks kept: results are truncated to uint8 with `astype` (no rounding); the latent merge unpacks each
coordinate tuple as (x1, x2, y1, y2) (patch_utils.py:140) although `crop_into_tiles` emits (y1, y2, x1, x2).
cv2 is not available here; it is only reached in the reference when a tile's shape differs from its target region
(INTER_LANCZOS4 resize) — that branch uses PIL's LANCZOS instead and is documented as not bit-identical."""
import numpy as np
import torch
import torch.nn.functional as F


def crop_into_tiles(img, tile_size, overlap=0, order="hwc"):
    if order == "hwc":
        h, w, _ = img.shape
    else:
        _, h, w = img.shape
    stride_y, stride_x = tile_size[0] - overlap, tile_size[1] - overlap
    tiles, coords = [], []
    for y in range(0, h, stride_y):
        for x in range(0, w, stride_x):
            y2, x2 = min(y + tile_size[0], h), min(x + tile_size[1], w)
            tiles.append(img[y:y2, x:x2, :] if order == "hwc" else img[:, y:y2, x:x2])
            coords.append((y, y2, x, x2))
    return tiles, coords, (h, w)


def _resize_hw(a2d, th, tw):
    from PIL import Image
    return np.asarray(Image.fromarray(np.asarray(a2d, np.float32), mode="F").resize((tw, th), Image.LANCZOS))


def _fit(tile, th, tw, order):
    if order == "hwc":
        if tile.shape[0] != th or tile.shape[1] != tw:
            tile = np.stack([_resize_hw(tile[:, :, c], th, tw) for c in range(tile.shape[2])], axis=2)
    else:
        if tile.shape[1] != th or tile.shape[2] != tw:
            tile = np.stack([_resize_hw(tile[c], th, tw) for c in range(tile.shape[0])])
    return tile


def _accumulate(tiles, coords, full_shape, order, mask_fn):
    h, w = full_shape
    c = tiles[0].shape[2] if order == "hwc" else tiles[0].shape[0]
    shape = (h, w, c) if order == "hwc" else (c, h, w)
    out, weight = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
    for tile, (y1, y2, x1, x2) in zip(tiles, coords):
        th, tw = y2 - y1, x2 - x1
        tile = _fit(tile, th, tw, order)
        m2 = mask_fn(th, tw)
        if order == "hwc":
            m = np.repeat(m2[:, :, None], c, axis=2)
            out[y1:y2, x1:x2, :] += tile.astype(np.float32) * m
            weight[y1:y2, x1:x2, :] += m
        else:
            m = np.repeat(m2[None, :, :], c, axis=0)
            out[:, y1:y2, x1:x2] += tile.astype(np.float32) * m
            weight[:, y1:y2, x1:x2] += m
    out /= np.maximum(weight, 1e-8)
    return out.astype(np.uint8)


def merge_tiles(tiles, coords, full_shape, order="hwc"):
    return _accumulate(tiles, coords, full_shape, order, lambda th, tw: np.ones((th, tw), np.float32))


def merge_costiles(tiles, coords, full_shape, order="hwc", feather=64):
    def cosine_window(n):
        x = np.linspace(-np.pi, np.pi, n)
        return (np.cos(x) + 1) / 2

    def mask(th, tw):
        wy, wx = np.ones(th), np.ones(tw)
        if feather > 0:
            f = min(feather, th // 2)
            wy[:f] = cosine_window(f)[:f]
            wy[-f:] = cosine_window(f)[-f:]
            f = min(feather, tw // 2)
            wx[:f] = cosine_window(f)[:f]
            wx[-f:] = cosine_window(f)[-f:]
        return np.outer(wy, wx).astype(np.float32)

    return _accumulate(tiles, coords, full_shape, order, mask)


def merge_ramp(tiles, coords, full_shape, order="hwc", feather=64):
    """Blend used by this package's tiled-decode driver (NOT a reference function): half-cosine ramps 0 -> 1 over
    `feather` pixels, applied only on tile edges that lie inside the frame (edges on the frame border keep weight 1), so
    weights sum to a strictly positive value everywhere.  `merge_costiles` above keeps the reference's arithmetic, whose
    window (a full 0->1->0 hump per edge) blacks out lines f-1 pixels from every tile edge."""
    h, w = full_shape

    def ramp(n, lo_inner, hi_inner):
        wv = np.ones(n, np.float32)
        f = min(feather, n // 2)
        if f > 0:
            r = (0.5 - 0.5 * np.cos(np.pi * (np.arange(f, dtype=np.float32) + 0.5) / f)).astype(np.float32)
            if lo_inner:
                wv[:f] = r
            if hi_inner:
                wv[-f:] = r[::-1]
        return wv

    c = tiles[0].shape[2] if order == "hwc" else tiles[0].shape[0]
    shape = (h, w, c) if order == "hwc" else (c, h, w)
    out, weight = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
    for tile, (y1, y2, x1, x2) in zip(tiles, coords):
        m2 = np.outer(ramp(y2 - y1, y1 > 0, y2 < h), ramp(x2 - x1, x1 > 0, x2 < w))
        tile = _fit(tile, y2 - y1, x2 - x1, order)
        if order == "hwc":
            out[y1:y2, x1:x2, :] += tile.astype(np.float32) * m2[:, :, None]
            weight[y1:y2, x1:x2, :] += m2[:, :, None]
        else:
            out[:, y1:y2, x1:x2] += tile.astype(np.float32) * m2[None]
            weight[:, y1:y2, x1:x2] += m2[None]
    return np.clip(np.rint(out / weight), 0, 255).astype(np.uint8)


def merge_latent_tiles_from_pixel_coords(latents, pixel_coords, full_latent_shape, original_image_size, eps: float = 1e-8):
    assert len(latents) == len(pixel_coords), "latents and coords length mismatch"
    device, dtype = latents[0].device, latents[0].dtype
    _, _, h_lat, w_lat = full_latent_shape
    h_px, w_px = original_image_size
    out = torch.zeros(full_latent_shape, device=device, dtype=dtype)
    weight = torch.zeros_like(out)

    def hann2d(h, w):
        wy = torch.ones(1, device=device, dtype=dtype) if h <= 1 else torch.hann_window(h, periodic=False, device=device, dtype=dtype)
        wx = torch.ones(1, device=device, dtype=dtype) if w <= 1 else torch.hann_window(w, periodic=False, device=device, dtype=dtype)
        m = wy.unsqueeze(1) * wx.unsqueeze(0)
        return m / (m.max() + 1e-12)

    for tile, (x1_px, x2_px, y1_px, y2_px) in zip(latents, pixel_coords):       # unpack order as in patch_utils.py:140
        ly1 = max(0, min(int(round(y1_px * (h_lat / float(h_px)))), h_lat))
        ly2 = max(0, min(int(round(y2_px * (h_lat / float(h_px)))), h_lat))
        lx1 = max(0, min(int(round(x1_px * (w_lat / float(w_px)))), w_lat))
        lx2 = max(0, min(int(round(x2_px * (w_lat / float(w_px)))), w_lat))
        th, tw = ly2 - ly1, lx2 - lx1
        if th <= 0 or tw <= 0:
            continue
        assert tile.dim() == 4 and tile.size(0) == 1, "expected tile shape (1,C,H,W)"
        if tile.shape[-2:] != (th, tw):
            tile = F.interpolate(tile, size=(th, tw), mode="bilinear", align_corners=False)
        mask = hann2d(th, tw)[None, None].expand(1, tile.size(1), th, tw)
        out[:, :, ly1:ly2, lx1:lx2] += tile * mask
        weight[:, :, ly1:ly2, lx1:lx2] += mask
    return out / torch.maximum(weight, torch.tensor(eps, device=device, dtype=dtype))
