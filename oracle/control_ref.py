"""ORACLE (test infrastructure only) — plain-torch fp32 restatement of the reference's control pyramid.

Follows, function by function:
  resize_and_normalize_flow   controlnet/control_utils.py:74-97
  compute_mask                controlnet/control_utils.py:11-17
  feature_warper              controlnet/control_utils.py:36-72   (FeatureWarperSoftsplat.forward)
  fdn                         controlnet/control_utils.py:19-34   (FDN.forward)
  bi_dir_feature_extractor    controlnet/extractors.py:209-316    (Bi_Dir_FeatureExtractor.forward)
  bi_dir_residue_extractor    controlnet/extractors.py:67-207     (Bi_Dir_ResidueExtractor.forward)
  warp_extractor              controlnet/extractors.py:26-65      (WarpExtractor.forward)

Weights come from a flat state dict in the reference's key layout (prefix e.g. "feature_extractor.").
Pinned by tests/golden/control_*.npz, captured from the imported reference modules
(oracle/make_goldens.py) with this package's C splat standing in for the CUDA-only kernel.

The reference hard-codes 512x512 (pyramid widths 64/32/16/8, extractors.py:278, flownet.py:79); here the
widths derive from the input (H/8, H/16, H/32, H/64) so reduced-size test cases run the same arithmetic.
"""
import torch
import torch.nn.functional as F

from .splat import softsplat


def _conv(sd, key, x, stride=1, padding=1, groups=1):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding, groups=groups)


def resize_and_normalize_flow(flow, th, tw):
    r = F.interpolate(flow, size=(th, tw), mode="bilinear", align_corners=False)
    u = r[:, 0] / ((tw - 1) / 2.0)
    v = r[:, 1] / ((th - 1) / 2.0)
    return torch.stack([u, v], dim=1)


def compute_mask(flow_bwd, flow_fwd):
    metric = torch.ones_like(flow_fwd[:, :1])
    warped = softsplat(flow_bwd, flow_fwd, metric, "soft")
    diff = flow_fwd + warped
    return (torch.norm(diff, p=2, dim=1, keepdim=True) > 0.3).float()


def feature_warper(sd, p, feat, flow, mask=None):
    m = F.silu(_conv(sd, p + "metric_net.0", feat))
    metric = _conv(sd, p + "metric_net.2", m)
    warped = softsplat(feat, flow, metric, "soft")
    if mask is not None:
        warped = warped * (1 - mask)
    return warped, metric


def fdn(sd, p, x, local_features, groups=32):
    normalized = F.group_norm(x, groups, None, None, 1e-5)
    assert local_features.shape[2:] == x.shape[2:]
    gamma = _conv(sd, p + "conv_gamma", local_features)
    beta = _conv(sd, p + "conv_beta", local_features)
    return normalized * (1 + gamma) + beta


def fdn_gamma_beta(sd, p, local_features):
    return _conv(sd, p + "conv_gamma", local_features), _conv(sd, p + "conv_beta", local_features)


def _pre_extractor(sd, p, x):
    x = F.silu(_conv(sd, p + "0", x))
    x = F.silu(_conv(sd, p + "2", x, stride=2))
    x = F.silu(_conv(sd, p + "4", x))
    x = F.silu(_conv(sd, p + "6", x, stride=2))
    x = F.silu(_conv(sd, p + "8", x))
    return x


def bi_dir_feature_extractor(sd, p, local_conditions, flow, return_aux=False):
    first_frame = local_conditions[:, 3:]     # extractors.py:266 — "first" is channels 3..5
    last_frame = local_conditions[:, :3]
    flow_fwd = flow[:, :2]
    flow_bwd = flow[:, 2:]
    ff = _pre_extractor(sd, p + "first_pre_extractor.", first_frame)
    lf = _pre_extractor(sd, p + "last_pre_extractor.", last_frame)
    h = local_conditions.shape[-2]
    flow_res = [h // 8, h // 16, h // 32, h // 64]
    outs, aux = [], []
    for idx in range(4):
        ff = F.silu(_conv(sd, p + f"extractors_first.{idx}.0", ff, stride=2))
        lf = F.silu(_conv(sd, p + f"extractors_last.{idx}.0", lf, stride=2))
        r = flow_res[idx]
        flow_f = resize_and_normalize_flow(flow_fwd, r, r)
        flow_b = resize_and_normalize_flow(flow_bwd, r, r)
        occ_f = compute_mask(flow_f, flow_b)
        occ_b = compute_mask(flow_b, flow_f)
        wf, cf = feature_warper(sd, p + f"wrapper.{idx}.", ff, flow_f, occ_f)
        wl, cb = feature_warper(sd, p + f"wrapper.{idx}.", lf, flow_b, occ_b)
        conf = torch.clamp(torch.cat([cf, cb], dim=1), min=0)
        w_norm = conf / (conf.sum(dim=1, keepdim=True) + 1e-6)
        fused = w_norm[:, :1] * wf + w_norm[:, 1:] * wl
        holes = (occ_f + occ_b) > 1.5
        if holes.any():
            fused = torch.where(holes.expand_as(fused), 0.5 * (wf + wl), fused)
        outs.append(_conv(sd, p + f"zero_convs.{idx}", fused))
        aux.append(dict(flow_f=flow_f, flow_b=flow_b, occ_f=occ_f, occ_b=occ_b, fused=fused))
    return (outs, aux) if return_aux else outs


def _conv_block(sd, p, x, stride):
    x = F.silu(_conv(sd, p + "block.0", x, stride=stride))
    return F.silu(_conv(sd, p + "block.2", x))


def warp_extractor(sd, p, x):
    f1 = _conv_block(sd, p + "enc1.", x, 4)
    f2 = _conv_block(sd, p + "enc2.", f1, 2)
    f3 = _conv_block(sd, p + "enc3.", f2, 2)
    f4 = _conv_block(sd, p + "enc4.", f3, 2)
    f5 = _conv_block(sd, p + "enc5.", f4, 2)
    return [_conv(sd, p + f"zero_convs.{i}", f) for i, f in enumerate((f2, f3, f4, f5))]


def bi_dir_residue_extractor(sd, p, prev_frame, next_frame, flow_fwd, flow_bwd):
    def pre(q, x):
        x = F.silu(_conv(sd, q + "0", x))
        x = F.silu(_conv(sd, q + "2", x, stride=2))
        return F.silu(_conv(sd, q + "4", x, stride=2))

    h = prev_frame.shape[-2]
    xp = pre(p + "prev_pre.", prev_frame)
    xn = pre(p + "next_pre.", next_frame)
    pf, nf = [], []
    for i in range(4):
        xp = F.silu(_conv(sd, p + f"prev_pyramids.{i}.0", xp, stride=2))
        xn = F.silu(_conv(sd, p + f"next_pyramids.{i}.0", xn, stride=2))
        pf.append(xp)
        nf.append(xn)
    outs = []
    for i, res in enumerate([h // 8, h // 16, h // 32, h // 64]):
        factor = h // res
        ffd = F.interpolate(flow_fwd, size=(res, res), mode="bilinear", align_corners=False) / factor
        fbd = F.interpolate(flow_bwd, size=(res, res), mode="bilinear", align_corners=False) / factor
        ffd = _conv(sd, p + f"flow_refiners.{i}", ffd, groups=2)
        fbd = _conv(sd, p + f"flow_refiners.{i}", fbd, groups=2)
        occ_f = compute_mask(ffd, fbd)
        occ_b = compute_mask(fbd, ffd)
        wp, cp = feature_warper(sd, p + f"warpers.{i}.", pf[i], ffd, occ_f)
        wn, cn = feature_warper(sd, p + f"warpers.{i}.", nf[i], fbd, occ_b)
        conf = torch.clamp(torch.cat([cp, cn], dim=1), min=0.0)
        w_norm = conf / (conf.sum(dim=1, keepdim=True) + 1e-6)
        fused = w_norm[:, :1] * wp + w_norm[:, 1:] * wn
        outs.append(_conv(sd, p + f"zero_convs.{i}", fused))
    return outs
