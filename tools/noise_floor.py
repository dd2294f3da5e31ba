"""Run-to-run PSNR of the small test pipeline with identical settings (splat atomics make the pyramid non-deterministic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import selftest as T
from diffcodec_amd.synthetic import synth_controls, synth_latents, synth_text
pipe, _ = T.build_small_pipeline()
cond, flow = synth_controls(1, 256)
pe, npe = synth_text(1, dim=T.SMALL_UNET["cross_attention_dim"])
lat = synth_latents(1, 256)
kw = dict(prompt_embeds=pe, negative_prompt_embeds=npe, controlnet_cond=cond, flow_cond=flow, latents=lat,
          num_inference_steps=3, guidance_scale=4.5, controlnet_conditioning_scale=1.7, output_type="pt")
runs = {}
for name, flag in (("off1", False), ("off2", False), ("on1", True), ("on2", True)):
    pipe.enable_cfg_shared_prefix(flag)
    runs[name] = pipe(**kw).images.float().cpu()
for a, b in (("off1", "off2"), ("on1", "on2"), ("off1", "on1"), ("off2", "on2")):
    print(a, b, round(T.psnr(runs[a], runs[b]), 2))
