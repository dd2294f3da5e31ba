"""Small end-to-end invocation of the hot path used by `__graft_entry__.smoke()` and the GPU tests:
a reduced-width SD-1.5-shaped model (same topology, block widths (64,128,256,256)) decoding one 256x256 frame,
checked against the CPU oracle on identical seeded weights / inputs.  The oracle is the checker here, never
the thing measured."""
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


def smoke_decode(steps=2, size=256, verbose=True):
    from oracle import pipeline_ref as R        # checker only
    pipe, (usd, csd, vsd) = build_small_pipeline()
    cond, flow = synth_controls(1, size)
    pe, npe = synth_text(1, dim=SMALL_UNET["cross_attention_dim"])
    lat = synth_latents(1, size)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=cond, flow_cond=flow, latents=lat,
               num_inference_steps=steps, guidance_scale=4.5, controlnet_conditioning_scale=1.7, output_type="pt")
    img = out.images.float().cpu()
    ref = R.decode_frame(usd, csd, vsd, SMALL_UNET, SMALL_VAE, cond, flow, pe, npe, lat, num_inference_steps=steps,
                         guidance_scale=4.5, controlnet_conditioning_scale=1.7)
    p = psnr(img, ref)
    if verbose:
        print(f"smoke: {steps}-step decode {size}x{size}: PSNR(hip bf16 vs oracle fp32) = {p:.2f} dB, rel-L2 = {rel_l2(img, ref):.4f}")
    assert torch.isfinite(img).all()
    assert p > 30.0, f"smoke decode diverged from the oracle: PSNR {p:.2f} dB"
    return p
