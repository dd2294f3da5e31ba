"""Full-size parity: one 512x512 frame, 20-step DDIM, CFG 4.5, control scale 1.7, true SD-1.5 widths, device (bf16) vs the
fp32 CPU oracle on identical seeded weights and inputs (developer tool; the oracle takes about a minute on 16 cores)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
from diffcodec_amd import selftest as T, weights as W
from diffcodec_amd.controlnet import HipDualFlowControlNet
from diffcodec_amd.pipeline import StableDiffusionDualFlowControlNetPipeline
from diffcodec_amd.scheduler import DDIMScheduler
from diffcodec_amd.synthetic import synth_controls, synth_latents, synth_text
from diffcodec_amd.unet import HipUNet2DConditionModel
from diffcodec_amd.vae import HipAutoencoderKL
from oracle import pipeline_ref as R

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
usd, csd, vsd = W.synthesize(W.unet_spec(), 0), W.synthesize(W.controlnet_spec(), 1), W.synthesize(W.vae_spec(), 2)
cond, flow = synth_controls(1, 512)
pe, npe = synth_text(1)
lat = synth_latents(1, 512)
kw = dict(num_inference_steps=steps, guidance_scale=4.5, controlnet_conditioning_scale=1.7)
pipe = StableDiffusionDualFlowControlNetPipeline(vae=HipAutoencoderKL(vsd), text_encoder=None, tokenizer=None,
                                                 unet=HipUNet2DConditionModel(usd), controlnet=HipDualFlowControlNet(csd),
                                                 scheduler=DDIMScheduler(), safety_checker=None, feature_extractor=None)
img = pipe(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=cond, flow_cond=flow, latents=lat, output_type="pt", **kw).images.float().cpu()
lat_d = pipe(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=cond, flow_cond=flow, latents=lat, output_type="latent", **kw).images.float().cpu()
t0 = time.time()
ref_img, ref_lat = R.decode_frame(usd, csd, vsd, W.SD15_UNET_CONFIG, W.SD15_VAE_CONFIG, cond, flow, pe, npe, lat, return_latents=True, **kw)
print(f"oracle {time.time() - t0:.1f}s; steps={steps}: image PSNR {T.psnr(img, ref_img):.2f} dB, latent rel-L2 {T.rel_l2(lat_d, ref_lat):.4f}", flush=True)
