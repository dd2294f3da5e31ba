"""Reduced-width SD-1.5-shaped operators (same topology, block widths (64,128,256,256)) for `__graft_entry__.smoke()`
and the GPU tests, plus the two error metrics they report.  Nothing here touches `oracle/`: the comparison against the
CPU oracle lives in `__graft_entry__.smoke()` and in `tests/`."""
import math

import torch

from . import weights as W
from .controlnet import HipDualFlowControlNet
from .pipeline import StableDiffusionDualFlowControlNetPipeline
from .scheduler import DDIMScheduler
from .synthetic import synth_controls, synth_latents, synth_text
from .unet import HipUNet2DConditionModel
from .vae import HipAutoencoderKL

SMALL_UNET = dict(W.SD15_UNET_CONFIG, block_out_channels=(64, 128, 256, 256), cross_attention_dim=128)
SMALL_VAE = dict(W.SD15_VAE_CONFIG, block_out_channels=(64, 128, 256, 256))


def small_state_dicts(seed=0):
    return (W.synthesize(W.unet_spec(SMALL_UNET), seed), W.synthesize(W.controlnet_spec(SMALL_UNET), seed + 1),
            W.synthesize(W.vae_spec(SMALL_VAE), seed + 2))


def build_small_pipeline(device="cuda", seed=0):
    usd, csd, vsd = small_state_dicts(seed)
    pipe = StableDiffusionDualFlowControlNetPipeline(
        vae=HipAutoencoderKL(vsd, SMALL_VAE, device), text_encoder=None, tokenizer=None,
        unet=HipUNet2DConditionModel(usd, SMALL_UNET, device), controlnet=HipDualFlowControlNet(csd, SMALL_UNET, device),
        scheduler=DDIMScheduler(), safety_checker=None, feature_extractor=None)
    return pipe, (usd, csd, vsd)


def psnr(a, b, peak=1.0):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10 * math.log10(peak * peak / mse)


def rel_l2(a, b):
    return (torch.linalg.norm((a.double() - b.double()).flatten()) / torch.linalg.norm(b.double().flatten())).item()
