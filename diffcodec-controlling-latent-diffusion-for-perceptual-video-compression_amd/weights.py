"""Checkpoint layout for the decode path: key/shape specs in the diffusers state-dict layout the reference
loads (validation.py:30-53), synthetic weights of that layout (no checkpoints are reachable offline), and
safetensors directory loaders.

Key layout follows SURVEY.md §8(b): SD-1.5 `unet/`, `vae/` directories and the ControlNet single safetensors
written by `save_pretrained` (train_controlnet.py:851-852) with the custom keys
`feature_extractor.*`, `fdn{64,32,16,08}.*` (controlnet/flownet.py:40-47).
"""
import json
import os
from collections import OrderedDict

import torch

SD15_UNET_CONFIG = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, num_heads=8,
                        cross_attention_dim=768, in_channels=4, out_channels=4, groups=32,
                        down_cross=(True, True, True, False), time_cond_proj_dim=None)
SD15_CLIP_TEXT_CONFIG = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                             vocab_size=49408, max_position_embeddings=77, layer_norm_eps=1e-5, eos_token_id=2)
SD15_VAE_CONFIG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                       in_channels=3, out_channels=3, groups=32, scaling_factor=0.18215)


def inject_channels(cfg):
    """flownet.py:38 hard-codes (320,320,640,1280) = (c0,c0,c1,c2) of the SD-1.5 block widths."""
    b = cfg["block_out_channels"]
    return (b[0], b[0], b[1], b[2])


# ------------------------------------------------------------------------------------------------ specs
def _resnet(spec, p, cin, cout, temb):
    spec[p + "norm1.weight"] = ("norm_w", (cin,))
    spec[p + "norm1.bias"] = ("norm_b", (cin,))
    spec[p + "conv1.weight"] = ("w", (cout, cin, 3, 3))
    spec[p + "conv1.bias"] = ("b", (cout,))
    if temb:
        spec[p + "time_emb_proj.weight"] = ("w", (cout, temb))
        spec[p + "time_emb_proj.bias"] = ("b", (cout,))
    spec[p + "norm2.weight"] = ("norm_w", (cout,))
    spec[p + "norm2.bias"] = ("norm_b", (cout,))
    spec[p + "conv2.weight"] = ("w", (cout, cout, 3, 3))
    spec[p + "conv2.bias"] = ("b", (cout,))
    if cin != cout:
        spec[p + "conv_shortcut.weight"] = ("w", (cout, cin, 1, 1))
        spec[p + "conv_shortcut.bias"] = ("b", (cout,))


def _transformer(spec, p, c, ctx):
    spec[p + "norm.weight"] = ("norm_w", (c,))
    spec[p + "norm.bias"] = ("norm_b", (c,))
    spec[p + "proj_in.weight"] = ("w", (c, c, 1, 1))
    spec[p + "proj_in.bias"] = ("b", (c,))
    q = p + "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        spec[q + n + ".weight"] = ("norm_w", (c,))
        spec[q + n + ".bias"] = ("norm_b", (c,))
    for a, kdim in (("attn1", c), ("attn2", ctx)):
        spec[q + a + ".to_q.weight"] = ("w", (c, c))
        spec[q + a + ".to_k.weight"] = ("w", (c, kdim))
        spec[q + a + ".to_v.weight"] = ("w", (c, kdim))
        spec[q + a + ".to_out.0.weight"] = ("w", (c, c))
        spec[q + a + ".to_out.0.bias"] = ("b", (c,))
    spec[q + "ff.net.0.proj.weight"] = ("w", (8 * c, c))
    spec[q + "ff.net.0.proj.bias"] = ("b", (8 * c,))
    spec[q + "ff.net.2.weight"] = ("w", (c, 4 * c))
    spec[q + "ff.net.2.bias"] = ("b", (c,))
    spec[p + "proj_out.weight"] = ("w", (c, c, 1, 1))
    spec[p + "proj_out.bias"] = ("b", (c,))


def _encoder_half(spec, cfg):
    boc = cfg["block_out_channels"]
    c0, temb, ctx = boc[0], 4 * boc[0], cfg["cross_attention_dim"]
    spec["conv_in.weight"] = ("w", (c0, cfg["in_channels"], 3, 3))
    spec["conv_in.bias"] = ("b", (c0,))
    spec["time_embedding.linear_1.weight"] = ("w", (temb, c0))
    spec["time_embedding.linear_1.bias"] = ("b", (temb,))
    spec["time_embedding.linear_2.weight"] = ("w", (temb, temb))
    spec["time_embedding.linear_2.bias"] = ("b", (temb,))
    cin = c0
    skip = [c0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            _resnet(spec, f"down_blocks.{i}.resnets.{j}.", cin, cout, temb)
            if cfg["down_cross"][i]:
                _transformer(spec, f"down_blocks.{i}.attentions.{j}.", cout, ctx)
            cin = cout
            skip.append(cout)
        if i != len(boc) - 1:
            spec[f"down_blocks.{i}.downsamplers.0.conv.weight"] = ("w", (cout, cout, 3, 3))
            spec[f"down_blocks.{i}.downsamplers.0.conv.bias"] = ("b", (cout,))
            skip.append(cout)
    cm = boc[-1]
    _resnet(spec, "mid_block.resnets.0.", cm, cm, temb)
    _transformer(spec, "mid_block.attentions.0.", cm, ctx)
    _resnet(spec, "mid_block.resnets.1.", cm, cm, temb)
    return skip


def unet_spec(cfg=SD15_UNET_CONFIG):
    spec = OrderedDict()
    boc = cfg["block_out_channels"]
    temb, ctx = 4 * boc[0], cfg["cross_attention_dim"]
    skip = _encoder_half(spec, cfg)
    nb = len(boc)
    rev = list(reversed(boc))
    prev = boc[-1]
    for i in range(nb):
        cout = rev[i]
        cross = cfg["down_cross"][nb - 1 - i]
        for j in range(cfg["layers_per_block"] + 1):
            sc = skip.pop()
            _resnet(spec, f"up_blocks.{i}.resnets.{j}.", prev + sc, cout, temb)
            if cross:
                _transformer(spec, f"up_blocks.{i}.attentions.{j}.", cout, ctx)
            prev = cout
        if i != nb - 1:
            spec[f"up_blocks.{i}.upsamplers.0.conv.weight"] = ("w", (cout, cout, 3, 3))
            spec[f"up_blocks.{i}.upsamplers.0.conv.bias"] = ("b", (cout,))
    spec["conv_norm_out.weight"] = ("norm_w", (boc[0],))
    spec["conv_norm_out.bias"] = ("norm_b", (boc[0],))
    spec["conv_out.weight"] = ("w", (cfg["out_channels"], boc[0], 3, 3))
    spec["conv_out.bias"] = ("b", (cfg["out_channels"],))
    return spec


def _feature_extractor_spec(spec, p, inj):
    """Bi_Dir_FeatureExtractor (extractors.py:211-262); shapes as captured in SURVEY.md appendix."""
    half = [c // 2 for c in inj]
    for side in ("first", "last"):
        chans = [(3, 16), (16, 32), (32, 32), (32, 64), (64, 64)]
        for idx, (ci, co) in zip((0, 2, 4, 6, 8), chans):
            spec[f"{p}{side}_pre_extractor.{idx}.weight"] = ("w", (co, ci, 3, 3))
            spec[f"{p}{side}_pre_extractor.{idx}.bias"] = ("b", (co,))
        cin = 64
        for i in range(4):
            spec[f"{p}extractors_{side}.{i}.0.weight"] = ("w", (half[i], cin, 3, 3))
            spec[f"{p}extractors_{side}.{i}.0.bias"] = ("b", (half[i],))
            cin = half[i]
    for i in range(4):
        spec[f"{p}wrapper.{i}.metric_net.0.weight"] = ("w", (64, half[i], 3, 3))
        spec[f"{p}wrapper.{i}.metric_net.0.bias"] = ("b", (64,))
        spec[f"{p}wrapper.{i}.metric_net.2.weight"] = ("w", (1, 64, 3, 3))
        spec[f"{p}wrapper.{i}.metric_net.2.bias"] = ("b", (1,))
        spec[f"{p}zero_convs.{i}.weight"] = ("zero_w", (inj[i], half[i], 3, 3))
        spec[f"{p}zero_convs.{i}.bias"] = ("zero_b", (inj[i],))


def controlnet_spec(cfg=SD15_UNET_CONFIG):
    """DualFlowControlNet state dict (flownet.py:23-47 over diffusers ControlNetModel)."""
    spec = OrderedDict()
    skip = _encoder_half(spec, cfg)
    for i, c in enumerate(skip):
        spec[f"controlnet_down_blocks.{i}.weight"] = ("zero_w", (c, c, 1, 1))
        spec[f"controlnet_down_blocks.{i}.bias"] = ("zero_b", (c,))
    cm = cfg["block_out_channels"][-1]
    spec["controlnet_mid_block.weight"] = ("zero_w", (cm, cm, 1, 1))
    spec["controlnet_mid_block.bias"] = ("zero_b", (cm,))
    inj = inject_channels(cfg)
    _feature_extractor_spec(spec, "feature_extractor.", inj)
    for name, c in zip(("fdn64", "fdn32", "fdn16", "fdn08"), inj):
        for g in ("conv_gamma", "conv_beta"):
            spec[f"{name}.{g}.weight"] = ("w", (c, c, 3, 3))
            spec[f"{name}.{g}.bias"] = ("b", (c,))
    return spec


def rescontrolnet_spec(cfg=SD15_UNET_CONFIG):
    """ResControlNet state dict (flow_resnet.py:23-48): same encoder / FDN / zero-convs, `feature_extractor.*` is the
    Bi_Dir_ResidueExtractor (extractors.py:78-147), plus `warp_extractor.*` (extractors.py:31-48)."""
    spec = OrderedDict()
    skip = _encoder_half(spec, cfg)
    for i, c in enumerate(skip):
        spec[f"controlnet_down_blocks.{i}.weight"] = ("zero_w", (c, c, 1, 1))
        spec[f"controlnet_down_blocks.{i}.bias"] = ("zero_b", (c,))
    cm = cfg["block_out_channels"][-1]
    spec["controlnet_mid_block.weight"] = ("zero_w", (cm, cm, 1, 1))
    spec["controlnet_mid_block.bias"] = ("zero_b", (cm,))
    inj = inject_channels(cfg)
    half = [c // 2 for c in inj]
    p = "feature_extractor."
    for side in ("prev", "next"):
        for idx, (ci, co) in zip((0, 2, 4), ((3, 32), (32, 64), (64, 64))):
            spec[f"{p}{side}_pre.{idx}.weight"] = ("w", (co, ci, 3, 3))
            spec[f"{p}{side}_pre.{idx}.bias"] = ("b", (co,))
        cin = 64
        for i in range(4):
            spec[f"{p}{side}_pyramids.{i}.0.weight"] = ("w", (half[i], cin, 3, 3))
            spec[f"{p}{side}_pyramids.{i}.0.bias"] = ("b", (half[i],))
            cin = half[i]
    for i, fe in enumerate((16, 16, 32, 32)):
        spec[f"{p}flow_refiners.{i}.weight"] = ("w", (2, 1, 3, 3))
        spec[f"{p}flow_refiners.{i}.bias"] = ("b", (2,))
        spec[f"{p}flow_feature_encoders.{i}.weight"] = ("w", (fe, 2, 3, 3))        # defined, never used in forward
        spec[f"{p}flow_feature_encoders.{i}.bias"] = ("b", (fe,))
        spec[f"{p}warpers.{i}.metric_net.0.weight"] = ("w", (64, half[i], 3, 3))
        spec[f"{p}warpers.{i}.metric_net.0.bias"] = ("b", (64,))
        spec[f"{p}warpers.{i}.metric_net.2.weight"] = ("w", (1, 64, 3, 3))
        spec[f"{p}warpers.{i}.metric_net.2.bias"] = ("b", (1,))
        spec[f"{p}zero_convs.{i}.weight"] = ("zero_w", (inj[i], half[i], 3, 3))
        spec[f"{p}zero_convs.{i}.bias"] = ("zero_b", (inj[i],))
    q = "warp_extractor."
    chans = [(3, 64), (64, inj[0]), (inj[0], inj[1]), (inj[1], inj[2]), (inj[2], inj[3])]
    for i, (ci, co) in enumerate(chans, start=1):
        spec[f"{q}enc{i}.block.0.weight"] = ("w", (co, ci, 3, 3))
        spec[f"{q}enc{i}.block.0.bias"] = ("b", (co,))
        spec[f"{q}enc{i}.block.2.weight"] = ("w", (co, co, 3, 3))
        spec[f"{q}enc{i}.block.2.bias"] = ("b", (co,))
    for i in range(4):
        spec[f"{q}zero_convs.{i}.weight"] = ("zero_w", (inj[i], inj[i], 3, 3))
        spec[f"{q}zero_convs.{i}.bias"] = ("zero_b", (inj[i],))
    for name, c in zip(("fdn64", "fdn32", "fdn16", "fdn08"), inj):
        for g in ("conv_gamma", "conv_beta"):
            spec[f"{name}.{g}.weight"] = ("w", (c, c, 3, 3))
            spec[f"{name}.{g}.bias"] = ("b", (c,))
    return spec


def vae_spec(cfg=SD15_VAE_CONFIG):
    spec = OrderedDict()
    boc = cfg["block_out_channels"]
    lc = cfg["latent_channels"]

    def attn(p, c):
        spec[p + "group_norm.weight"] = ("norm_w", (c,))
        spec[p + "group_norm.bias"] = ("norm_b", (c,))
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            spec[p + n + ".weight"] = ("w", (c, c))
            spec[p + n + ".bias"] = ("b", (c,))

    # encoder
    spec["encoder.conv_in.weight"] = ("w", (boc[0], cfg["in_channels"], 3, 3))
    spec["encoder.conv_in.bias"] = ("b", (boc[0],))
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            _resnet(spec, f"encoder.down_blocks.{i}.resnets.{j}.", cin, cout, 0)
            cin = cout
        if i != len(boc) - 1:
            spec[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = ("w", (cout, cout, 3, 3))
            spec[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = ("b", (cout,))
    cm = boc[-1]
    _resnet(spec, "encoder.mid_block.resnets.0.", cm, cm, 0)
    attn("encoder.mid_block.attentions.0.", cm)
    _resnet(spec, "encoder.mid_block.resnets.1.", cm, cm, 0)
    spec["encoder.conv_norm_out.weight"] = ("norm_w", (cm,))
    spec["encoder.conv_norm_out.bias"] = ("norm_b", (cm,))
    spec["encoder.conv_out.weight"] = ("w", (2 * lc, cm, 3, 3))
    spec["encoder.conv_out.bias"] = ("b", (2 * lc,))
    spec["quant_conv.weight"] = ("w", (2 * lc, 2 * lc, 1, 1))
    spec["quant_conv.bias"] = ("b", (2 * lc,))
    # decoder
    spec["post_quant_conv.weight"] = ("w", (lc, lc, 1, 1))
    spec["post_quant_conv.bias"] = ("b", (lc,))
    spec["decoder.conv_in.weight"] = ("w", (cm, lc, 3, 3))
    spec["decoder.conv_in.bias"] = ("b", (cm,))
    _resnet(spec, "decoder.mid_block.resnets.0.", cm, cm, 0)
    attn("decoder.mid_block.attentions.0.", cm)
    _resnet(spec, "decoder.mid_block.resnets.1.", cm, cm, 0)
    rev = list(reversed(boc))
    cin = cm
    for i, cout in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            _resnet(spec, f"decoder.up_blocks.{i}.resnets.{j}.", cin, cout, 0)
            cin = cout
        if i != len(boc) - 1:
            spec[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = ("w", (cout, cout, 3, 3))
            spec[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = ("b", (cout,))
    spec["decoder.conv_norm_out.weight"] = ("norm_w", (boc[0],))
    spec["decoder.conv_norm_out.bias"] = ("norm_b", (boc[0],))
    spec["decoder.conv_out.weight"] = ("w", (cfg["out_channels"], boc[0], 3, 3))
    spec["decoder.conv_out.bias"] = ("b", (cfg["out_channels"],))
    return spec


def clip_text_spec(cfg=SD15_CLIP_TEXT_CONFIG):
    """transformers `CLIPTextModel` state-dict layout (`<base>/text_encoder/model.safetensors`, validation.py:31-32):
    ViT-L/14 text tower, 123.06 M parameters at the SD-1.5 config."""
    spec = OrderedDict()
    c, f = cfg["hidden_size"], cfg["intermediate_size"]
    spec["text_model.embeddings.token_embedding.weight"] = ("emb", (cfg["vocab_size"], c))
    spec["text_model.embeddings.position_embedding.weight"] = ("emb", (cfg["max_position_embeddings"], c))
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            spec[p + f"self_attn.{n}.weight"] = ("w", (c, c))
            spec[p + f"self_attn.{n}.bias"] = ("b", (c,))
        for n in ("layer_norm1", "layer_norm2"):
            spec[p + n + ".weight"] = ("norm_w", (c,))
            spec[p + n + ".bias"] = ("norm_b", (c,))
        spec[p + "mlp.fc1.weight"] = ("w", (f, c))
        spec[p + "mlp.fc1.bias"] = ("b", (f,))
        spec[p + "mlp.fc2.weight"] = ("w", (c, f))
        spec[p + "mlp.fc2.bias"] = ("b", (c,))
    spec["text_model.final_layer_norm.weight"] = ("norm_w", (c,))
    spec["text_model.final_layer_norm.bias"] = ("norm_b", (c,))
    return spec


# ------------------------------------------------------------------------------------- synthetic weights
def synthesize(spec, seed=0, bf16_round=True, gain=1.0):
    """Seeded weights of the given spec: W ~ N(0, gain/fan_in), b ~ N(0, 0.02), norm affine near identity,
    zero-convs N(0, 0.02) (a trained checkpoint's are non-zero).  Rounded to bf16 once so that the fp32 CPU
    oracle and the bf16 device path see identical parameters."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, (kind, shape) in spec.items():
        if kind == "w":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * (gain / fan_in) ** 0.5
        elif kind == "b":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "norm_w":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "norm_b":
            t = 0.05 * torch.randn(shape, generator=g)
        elif kind in ("zero_w", "zero_b", "emb"):
            t = torch.randn(shape, generator=g) * 0.02
        else:
            raise KeyError(kind)
        if bf16_round:
            t = t.to(torch.bfloat16).float()
        sd[k] = t
    return sd


def param_count(spec):
    n = 0
    for _, (_, shape) in spec.items():
        m = 1
        for s in shape:
            m *= s
        n += m
    return n


# -------------------------------------------------------------------------------------------- file I/O
def load_safetensors(path):
    from safetensors.torch import load_file
    return load_file(path)


_VAE_ATTN_LEGACY = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def remap_vae_attention_keys(sd):
    """SD-1.5's `vae/diffusion_pytorch_model.safetensors` predates diffusers' Attention refactor: its mid-block attention
    is stored as `*.attentions.0.{query,key,value,proj_attn}.{weight,bias}` (some exports as [C,C,1,1] 1x1-conv weights)
    and diffusers renames them while loading (`AutoencoderKL.from_pretrained`, validation.py:33) [recalled].  Same
    conversion here: -> `{to_q,to_k,to_v,to_out.0}` with [C,C] weights.  New-style checkpoints pass through unchanged."""
    out = OrderedDict()
    for k, v in sd.items():
        parts = k.split(".")
        if len(parts) >= 3 and "attentions" in parts and parts[-2] in _VAE_ATTN_LEGACY:
            k = ".".join(parts[:-2] + [_VAE_ATTN_LEGACY[parts[-2]], parts[-1]])
        if ".attentions." in k and k.endswith(".weight") and v.dim() == 4 and v.shape[2:] == (1, 1) and \
                k.rsplit(".", 2)[-2] in ("to_q", "to_k", "to_v", "0"):
            v = v[:, :, 0, 0]
        out[k] = v
    return out


def load_diffusers_subfolder(base, sub):
    """<base>/<sub>/{config.json, diffusion_pytorch_model.safetensors} (validation.py:33-34)."""
    d = os.path.join(base, sub)
    with open(os.path.join(d, "config.json")) as f:
        cfg = json.load(f)
    sd = load_safetensors(os.path.join(d, "diffusion_pytorch_model.safetensors"))
    return cfg, (remap_vae_attention_keys(sd) if sub == "vae" else sd)


def filter_state_dict(ckpt, spec):
    """`load_state_dict(strict=False)` + the notebook's shape filter (pipeline.ipynb cell 1): keep keys that
    exist in the model with the same shape; report the rest."""
    kept, missing, unexpected, mismatched = OrderedDict(), [], [], []
    for k, (_, shape) in spec.items():
        if k not in ckpt:
            missing.append(k)
        elif tuple(ckpt[k].shape) != tuple(shape):
            mismatched.append(k)
        else:
            kept[k] = ckpt[k]
    for k in ckpt:
        if k not in spec:
            unexpected.append(k)
    return kept, dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def unet_config_from_diffusers(cfg_json):
    """Map a diffusers unet/config.json onto this package's config dict."""
    boc = tuple(cfg_json["block_out_channels"])
    ahd = cfg_json.get("attention_head_dim", 8)
    heads = ahd if isinstance(ahd, int) else ahd[0]     # SD-1.5: "attention_head_dim": 8 is the head COUNT
    dbt = cfg_json.get("down_block_types", ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"])
    return dict(block_out_channels=boc, layers_per_block=cfg_json.get("layers_per_block", 2), num_heads=heads,
                cross_attention_dim=cfg_json.get("cross_attention_dim", 768), in_channels=cfg_json.get("in_channels", 4),
                out_channels=cfg_json.get("out_channels", 4), groups=cfg_json.get("norm_num_groups", 32),
                down_cross=tuple("CrossAttn" in t for t in dbt), time_cond_proj_dim=cfg_json.get("time_cond_proj_dim"))


def vae_config_from_diffusers(cfg_json):
    """Map a diffusers vae/config.json (AutoencoderKL) onto this package's config dict.  SD-1.5's file predates the
    `scaling_factor` entry: diffusers' default 0.18215 applies then (pipeline.py:391 divides the latents by it)."""
    return dict(block_out_channels=tuple(cfg_json["block_out_channels"]), layers_per_block=cfg_json.get("layers_per_block", 2),
                latent_channels=cfg_json.get("latent_channels", 4), in_channels=cfg_json.get("in_channels", 3),
                out_channels=cfg_json.get("out_channels", 3), groups=cfg_json.get("norm_num_groups", 32),
                scaling_factor=cfg_json.get("scaling_factor", 0.18215))
