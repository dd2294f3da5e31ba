"""ORACLE (test infrastructure only) — fp32 CPU restatement of the sampling loop
`StableDiffusionDualFlowControlNetPipeline.__call__` (pipeline.py:144-404) with the DDIM scheduler the
BASELINE configs name (diffusers DDIMScheduler with the SD-1.5 scheduler_config [recalled]:
scaled_linear betas 0.00085..0.012, 1000 train steps, steps_offset=1, set_alpha_to_one=False,
leading spacing, epsilon prediction, eta=0, no clipping).  Parity unpinned (no fixture in the reference).
"""
import numpy as np
import torch

from . import sd15_ref as M


class DDIMRef:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.T = num_train_timesteps
        self.steps_offset = steps_offset

    def set_timesteps(self, n):
        self.n = n
        ratio = self.T // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x):
        t = int(t)
        prev = t - self.T // self.n
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


@torch.no_grad()
def decode_frame(unet_sd, cn_sd, vae_sd, unet_cfg, vae_cfg, controlnet_cond, flow_cond, prompt_embeds,
                 negative_prompt_embeds, latents, num_inference_steps=20, guidance_scale=7.5,
                 controlnet_conditioning_scale=1.0, output_type="pt", hoist=True, return_latents=False,
                 control_guidance_start=0.0, control_guidance_end=1.0):
    """pipeline.py:144-404, `prompt_embeds=` path.  hoist=True computes the step-invariant pyramid once
    (identical values to recomputing it every step as the reference does, flownet.py:78)."""
    do_cfg = guidance_scale is not None and guidance_scale > 1.0            # pipeline.py:202
    sched = DDIMRef()
    sched.set_timesteps(num_inference_steps)
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds], 0) if do_cfg else prompt_embeds   # :234-236
    latents = latents.float() * sched.init_noise_sigma
    from . import control_ref as C
    pyr = C.bi_dir_feature_extractor(cn_sd, "feature_extractor.", controlnet_cond.float(), flow_cond.float()) if hoist else None
    nt = len(sched.timesteps)
    for i, t in enumerate(sched.timesteps):
        keep = 1.0 - float(i / nt < control_guidance_start or (i + 1) / nt > control_guidance_end)  # :292-295
        x_in = torch.cat([latents, latents], 0) if do_cfg else latents     # :313-320
        cc = torch.cat([controlnet_cond] * 2, 0) if do_cfg else controlnet_cond
        fc = torch.cat([flow_cond] * 2, 0) if do_cfg else flow_cond
        p = [torch.cat([q, q], 0) for q in pyr] if (pyr is not None and do_cfg) else pyr
        down, mid = M.dualflow_controlnet_forward(cn_sd, unet_cfg, x_in, t, ctx, cc, fc,
                                                  controlnet_conditioning_scale * keep, pyramid=p)
        eps = M.unet_forward(unet_sd, unet_cfg, x_in, t, ctx, down, mid)   # :358-367
        if do_cfg:
            eu, et = eps.chunk(2)
            eps = eu + guidance_scale * (et - eu)                           # :370-372
        latents = sched.step(eps, t, latents)                               # :375
    if output_type == "latent" or return_latents and vae_sd is None:
        return latents
    img = M.vae_decode(vae_sd, vae_cfg, latents / vae_cfg["scaling_factor"])   # :391
    img = (img / 2 + 0.5).clamp(0, 1)                                       # :397-398 postprocess
    return (img, latents) if return_latents else img
