"""Synthetic decode inputs of the shapes the reference feeds `pipe(...)` (validation.py:85-93,132-146):
smooth [B,6,H,W] anchor frames in [0,1] and smooth bidirectional [B,4,H,W] flow in pixel units whose
forward/backward consistency error straddles the 0.3 occlusion threshold (control_utils.py:16), so both the
mask and the double-hole branch (extractors.py:307-310) are exercised.  Definition: SURVEY.md §8(d)."""
import torch
import torch.nn.functional as F


def synth_controls(b, size, seed=1234):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(b, 6, size, size, generator=g)
    img = F.avg_pool2d(img, 9, 1, 4, count_include_pad=False)
    lo = torch.randn(b, 2, size // 16, size // 16, generator=g) * 8.0
    fwd = F.interpolate(lo, size=(size, size), mode="bilinear", align_corners=False)
    bwd = -fwd + 0.5 * F.interpolate(torch.randn(b, 2, size // 16, size // 16, generator=g), size=(size, size),
                                     mode="bilinear", align_corners=False)
    return img, torch.cat([fwd, bwd], 1)


def synth_text(b, seed=77, tokens=77, dim=768):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(b, tokens, dim, generator=g), torch.randn(b, tokens, dim, generator=g)


def synth_latents(b, size, seed=4321, channels=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(b, channels, size // 8, size // 8, generator=g)
