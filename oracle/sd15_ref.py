"""ORACLE (test infrastructure only) — plain-torch fp32 restatement of the third-party model arithmetic the
reference's decode path calls: diffusers `UNet2DConditionModel`, `ControlNetModel` blocks (as driven by
controlnet/flownet.py:51-138 and controlnet/flow_resnet.py:52-144), `AutoencoderKL` encode/decode.

diffusers is NOT vendored in /root/reference and is not installed here (floor 0.35.0.dev0,
train_controlnet.py:68; unpinned in controlnet/requirements.txt).  This file restates the published SD-1.5
topology; call sites it anchors on: pipeline.py:341-367,391; flownet.py:74-75,83-124.  The reference holds
no numeric fixture at that boundary -> **parity unpinned** for this file (DESIGN.md "Oracle").

All functions are functional over a flat state dict in diffusers key layout.
"""
import math

import torch
import torch.nn.functional as F

from . import control_ref as C

SD15_UNET = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, num_heads=8,
                 cross_attention_dim=768, in_channels=4, out_channels=4, groups=32,
                 down_cross=(True, True, True, False))
SD15_VAE = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                in_channels=3, out_channels=3, groups=32, scaling_factor=0.18215)


def timestep_embedding(timesteps, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) [recalled]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _lin(sd, k, x):
    return F.linear(x, sd[k + ".weight"], sd.get(k + ".bias"))


def _conv(sd, k, x, stride=1, padding=1):
    return F.conv2d(x, sd[k + ".weight"], sd.get(k + ".bias"), stride=stride, padding=padding)


def _gn(sd, k, x, groups, eps):
    return F.group_norm(x, groups, sd[k + ".weight"], sd[k + ".bias"], eps)


def time_embed(sd, t, n, ch0):
    t = torch.as_tensor(t).reshape(-1).expand(n) if torch.as_tensor(t).numel() == 1 else torch.as_tensor(t)
    e = timestep_embedding(t, ch0)
    return _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", e)))


def resnet(sd, p, x, temb, groups, eps):
    h = _conv(sd, p + "conv1", F.silu(_gn(sd, p + "norm1", x, groups, eps)))
    if temb is not None and (p + "time_emb_proj.weight") in sd:
        h = h + _lin(sd, p + "time_emb_proj", F.silu(temb))[:, :, None, None]
    h = _conv(sd, p + "conv2", F.silu(_gn(sd, p + "norm2", h, groups, eps)))
    if (p + "conv_shortcut.weight") in sd:
        x = _conv(sd, p + "conv_shortcut", x, padding=0)
    return x + h


def _attn(sd, p, x, ctx, heads):
    q = _lin(sd, p + "to_q", x)
    k = _lin(sd, p + "to_k", ctx)
    v = _lin(sd, p + "to_v", ctx)
    b, n, c = q.shape
    d = c // heads
    q = q.view(b, n, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(b, n, c)
    return _lin(sd, p + "to_out.0", o)


def transformer(sd, p, x, ctx, heads, groups):
    b, c, h, w = x.shape
    res = x
    y = _gn(sd, p + "norm", x, groups, 1e-6)
    y = _conv(sd, p + "proj_in", y, padding=0)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    q = p + "transformer_blocks.0."
    n1 = F.layer_norm(y, (c,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], 1e-5)
    y = y + _attn(sd, q + "attn1.", n1, n1, heads)
    n2 = F.layer_norm(y, (c,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], 1e-5)
    y = y + _attn(sd, q + "attn2.", n2, ctx, heads)
    n3 = F.layer_norm(y, (c,), sd[q + "norm3.weight"], sd[q + "norm3.bias"], 1e-5)
    hg = _lin(sd, q + "ff.net.0.proj", n3)
    hid, gate = hg.chunk(2, dim=-1)
    y = y + _lin(sd, q + "ff.net.2", hid * F.gelu(gate))
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    y = _conv(sd, p + "proj_out", y, padding=0)
    return y + res


def _down_blocks(sd, cfg, sample, emb, ctx, after_block=None):
    """Shared by UNet and ControlNet.  `after_block(i, sample)` is the FDN hook of flownet.py:98-106."""
    g = cfg["groups"]
    boc = cfg["block_out_channels"]
    res = [sample]
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            sample = resnet(sd, f"down_blocks.{i}.resnets.{j}.", sample, emb, g, 1e-5)
            if cfg["down_cross"][i]:
                sample = transformer(sd, f"down_blocks.{i}.attentions.{j}.", sample, ctx, cfg["num_heads"], g)
            res.append(sample)
        if i != len(boc) - 1:
            sample = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", sample, stride=2)
            res.append(sample)
        if after_block is not None:
            sample = after_block(i, sample)
    return sample, res


def _mid_block(sd, cfg, sample, emb, ctx):
    g = cfg["groups"]
    sample = resnet(sd, "mid_block.resnets.0.", sample, emb, g, 1e-5)
    sample = transformer(sd, "mid_block.attentions.0.", sample, ctx, cfg["num_heads"], g)
    return resnet(sd, "mid_block.resnets.1.", sample, emb, g, 1e-5)


def fourier_filter(x, threshold, scale):
    """diffusers.utils.torch_utils.fourier_filter [recalled]: scale the centred (2*threshold)^2 low-frequency block."""
    xf = torch.fft.fftshift(torch.fft.fftn(x.float(), dim=(-2, -1)), dim=(-2, -1))
    h, w = x.shape[-2:]
    mask = torch.ones_like(xf.real)
    mask[..., h // 2 - threshold:h // 2 + threshold, w // 2 - threshold:w // 2 + threshold] = scale
    return torch.fft.ifftn(torch.fft.ifftshift(xf * mask, dim=(-2, -1)), dim=(-2, -1)).real


def apply_freeu(resolution_idx, hidden, res_hidden, s1, s2, b1, b2):
    """diffusers.utils.torch_utils.apply_freeu [recalled] (enabled at validation.py:106)."""
    if resolution_idx in (0, 1):
        b, s = (b1, s1) if resolution_idx == 0 else (b2, s2)
        half = hidden.shape[1] // 2
        hidden = torch.cat([hidden[:, :half] * b, hidden[:, half:]], 1)
        res_hidden = fourier_filter(res_hidden, 1, s)
    return hidden, res_hidden


def unet_forward(sd, cfg, sample, t, ctx, down_res=None, mid_res=None, freeu=None):
    """UNet2DConditionModel.forward as called at pipeline.py:358-367 [recalled topology]."""
    boc = cfg["block_out_channels"]
    g = cfg["groups"]
    emb = time_embed(sd, t, sample.shape[0], boc[0])
    sample = _conv(sd, "conv_in", sample)
    sample, res = _down_blocks(sd, cfg, sample, emb, ctx)
    if down_res is not None:
        res = [a + b for a, b in zip(res, down_res)]
    sample = _mid_block(sd, cfg, sample, emb, ctx)
    if mid_res is not None:
        sample = sample + mid_res
    nb = len(boc)
    for i in range(nb):
        cross = cfg["down_cross"][nb - 1 - i]
        for j in range(cfg["layers_per_block"] + 1):
            skip = res.pop()
            if freeu is not None:
                sample, skip = apply_freeu(i, sample, skip, **freeu)
            sample = torch.cat([sample, skip], dim=1)
            sample = resnet(sd, f"up_blocks.{i}.resnets.{j}.", sample, emb, g, 1e-5)
            if cross:
                sample = transformer(sd, f"up_blocks.{i}.attentions.{j}.", sample, ctx, cfg["num_heads"], g)
        if i != nb - 1:
            sample = F.interpolate(sample, scale_factor=2.0, mode="nearest")
            sample = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", sample)
    sample = F.silu(_gn(sd, "conv_norm_out", sample, g, 1e-5))
    return _conv(sd, "conv_out", sample)


def dualflow_controlnet_forward(sd, cfg, sample, t, ctx, controlnet_cond, flow_cond, conditioning_scale=1.0,
                                pyramid=None, warp_cond=None, residual_variant=False):
    """DualFlowControlNet.forward (flownet.py:51-138); with residual_variant=True, ResControlNet.forward
    (flow_resnet.py:52-144: pyramid = residue pyramid + warp pyramid)."""
    boc = cfg["block_out_channels"]
    emb = time_embed(sd, t, sample.shape[0], boc[0])                      # flownet.py:67-75
    if pyramid is None:
        if residual_variant:
            p = C.bi_dir_residue_extractor(sd, "feature_extractor.", controlnet_cond[:, :3], controlnet_cond[:, 3:],
                                           flow_cond[:, :2], flow_cond[:, 2:])
            w = C.warp_extractor(sd, "warp_extractor.", warp_cond)
            pyramid = [a + b for a, b in zip(p, w)]                       # flow_resnet.py:90,106-112
        else:
            pyramid = C.bi_dir_feature_extractor(sd, "feature_extractor.", controlnet_cond, flow_cond)  # :78
    p64, p32, p16, p08 = pyramid
    sample = _conv(sd, "conv_in", sample)                                 # :83
    sample = C.fdn(sd, "fdn64.", sample, p64, cfg["groups"])              # :84

    def hook(i, s):                                                       # :98-106 (fdn08 twice, shared)
        if i == 0:
            return C.fdn(sd, "fdn32.", s, p32, cfg["groups"])
        if i == 1:
            return C.fdn(sd, "fdn16.", s, p16, cfg["groups"])
        return C.fdn(sd, "fdn08.", s, p08, cfg["groups"])

    sample, res = _down_blocks(sd, cfg, sample, emb, ctx, after_block=hook)
    sample = _mid_block(sd, cfg, sample, emb, ctx)                        # :112-118
    down = [_conv(sd, f"controlnet_down_blocks.{i}", r, padding=0) * conditioning_scale for i, r in enumerate(res)]
    mid = _conv(sd, "controlnet_mid_block", sample, padding=0) * conditioning_scale   # :120-128
    return down, mid


# ------------------------------------------------------------------------------------------- VAE
def _vae_attn(sd, p, x, groups):
    b, c, h, w = x.shape
    y = _gn(sd, p + "group_norm", x, groups, 1e-6)
    y = y.reshape(b, c, h * w).transpose(1, 2)
    q = _lin(sd, p + "to_q", y)
    k = _lin(sd, p + "to_k", y)
    v = _lin(sd, p + "to_v", y)
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = _lin(sd, p + "to_out.0", o)
    return x + o.transpose(1, 2).reshape(b, c, h, w)


def _vae_mid(sd, p, x, g):
    x = resnet(sd, p + "resnets.0.", x, None, g, 1e-6)
    x = _vae_attn(sd, p + "attentions.0.", x, g)
    return resnet(sd, p + "resnets.1.", x, None, g, 1e-6)


def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode as called at pipeline.py:391 (caller divides by scaling_factor) [recalled]."""
    g = cfg["groups"]
    boc = cfg["block_out_channels"]
    x = _conv(sd, "post_quant_conv", z, padding=0)
    x = _conv(sd, "decoder.conv_in", x)
    x = _vae_mid(sd, "decoder.mid_block.", x, g)
    nb = len(boc)
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", x, None, g, 1e-6)
        if i != nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, g, 1e-6))
    return _conv(sd, "decoder.conv_out", x)


def vae_encode_moments(sd, cfg, x):
    """AutoencoderKL.encode -> (mean, logvar) (train_controlnet.py:1081; pipeline.ipynb cell 7) [recalled]."""
    g = cfg["groups"]
    boc = cfg["block_out_channels"]
    x = _conv(sd, "encoder.conv_in", x)
    nb = len(boc)
    for i in range(nb):
        for j in range(cfg["layers_per_block"]):
            x = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", x, None, g, 1e-6)
        if i != nb - 1:
            x = F.pad(x, (0, 1, 0, 1))
            x = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=0)
    x = _vae_mid(sd, "encoder.mid_block.", x, g)
    x = F.silu(_gn(sd, "encoder.conv_norm_out", x, g, 1e-6))
    x = _conv(sd, "encoder.conv_out", x)
    x = _conv(sd, "quant_conv", x, padding=0)
    mean, logvar = x.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_encode_sample(sd, cfg, x, noise):
    mean, logvar = vae_encode_moments(sd, cfg, x)
    return mean + torch.exp(0.5 * logvar) * noise
