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

    def step(self, eps, t, x, eta=0.0, noise=None):
        """eta > 0: diffusers' stochastic DDIM [recalled] in its own (unfolded) form — variance of `_get_variance`,
        direction term sqrt(1 - a_prev - std^2) eps, plus std * noise."""
        t = int(t)
        prev = t - self.T // self.n
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        if not eta:
            return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps
        var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        std = eta * var ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * eps + std * noise


class UniPCRef:
    """diffusers UniPCMultistepScheduler (solver_order 2, bh2, predict_x0, lower_order_final, leading spacing,
    steps_offset 1, final sigma 0) as instantiated at validation.py:37 [recalled; parity unpinned].  Written in the
    library's tensor form (D1s / rhos / einsum), independently of the product's folded scalar coefficients."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, T=1000, beta_start=0.00085, beta_end=0.012, solver_order=2):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
        self.ac = torch.cumprod(1.0 - betas, dim=0).double()
        self.T, self.solver_order = T, solver_order

    def set_timesteps(self, n):
        ratio = self.T // (n + 1)
        ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + 1
        sig = (((1 - self.ac) / self.ac) ** 0.5).numpy()
        self.sigmas = torch.from_numpy(np.concatenate([np.interp(ts, np.arange(0, len(sig)), sig), [0.0]]))
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums, self.last_sample, self.step_index, self.this_order = 0, None, 0, 1

    @staticmethod
    def _as(sigma):
        alpha = 1 / (sigma ** 2 + 1) ** 0.5
        return alpha, sigma * alpha

    def _update(self, x, m0, older, sig_t, sig_s0, sig_prev, order, model_t=None):
        a_t, s_t = self._as(sig_t)
        a_s, s_s = self._as(sig_s0)
        lam_t, lam_s = torch.log(a_t) - torch.log(s_t), torch.log(a_s) - torch.log(s_s)
        h = lam_t - lam_s
        rks, d1s = [], []
        for k in range(order - 1):
            a_k, s_k = self._as(sig_prev[k])
            rk = ((torch.log(a_k) - torch.log(s_k)) - lam_s) / h
            rks.append(rk)
            d1s.append((older[k] - m0) / rk)
        rks.append(torch.tensor(1.0, dtype=torch.float64))
        rks = torch.stack(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = torch.expm1(hh)
        rr, bb, fact = [], [], 1
        for i in range(1, order + 1):
            rr.append(rks ** (i - 1))
            bb.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        rr, bb = torch.stack(rr), torch.stack(bb)
        x_t_ = s_t / s_s * x - a_t * h_phi_1 * m0
        if model_t is None:                                   # predictor
            res = 0.5 * d1s[0] if order == 2 else 0.0
            return x_t_ - a_t * b_h * res
        rhos = torch.tensor([0.5], dtype=torch.float64) if order == 1 else torch.linalg.solve(rr, bb)
        res = rhos[0] * d1s[0] if order == 2 else 0.0
        return x_t_ - a_t * b_h * (res + rhos[-1] * (model_t - m0))

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x):
        i = self.step_index
        x, eps = x.double(), eps.double()
        a_i, s_i = self._as(self.sigmas[i])
        m_t = (x - s_i * eps) / a_i
        if i > 0 and self.last_sample is not None:
            o = self.this_order
            x = self._update(self.last_sample, self.model_outputs[-1], [self.model_outputs[-2]] if o == 2 else [], self.sigmas[i],
                             self.sigmas[i - 1], [self.sigmas[i - 2]] if o == 2 else [], o, model_t=m_t)
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = m_t
        self.this_order = min(min(self.solver_order, len(self.timesteps) - i), self.lower_order_nums + 1)
        self.last_sample = x
        o = self.this_order
        if i + 1 == len(self.timesteps):                      # final sigma 0: h = inf, update collapses to the x0 prediction
            prev = m_t
        else:
            prev = self._update(x, m_t, [self.model_outputs[-2]] if o == 2 else [], self.sigmas[i + 1], self.sigmas[i],
                                [self.sigmas[i - 1]] if o == 2 else [], o)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.float()


@torch.no_grad()
def decode_frame(unet_sd, cn_sd, vae_sd, unet_cfg, vae_cfg, controlnet_cond, flow_cond, prompt_embeds,
                 negative_prompt_embeds, latents, num_inference_steps=20, guidance_scale=7.5,
                 controlnet_conditioning_scale=1.0, output_type="pt", hoist=True, return_latents=False,
                 control_guidance_start=0.0, control_guidance_end=1.0, eta=0.0, generator=None, noise_dtype=torch.float32,
                 res_cn_sd=None, warp_cond=None, res_conditioning_scale=None, scheduler="ddim", freeu=None):
    """pipeline.py:144-404, `prompt_embeds=` path.  hoist=True computes the step-invariant pyramid once
    (identical values to recomputing it every step as the reference does, flownet.py:78)."""
    do_cfg = guidance_scale is not None and guidance_scale > 1.0            # pipeline.py:202
    # scheduler="unipc" + freeu=dict(s1, s2, b1, b2): the configuration validation.py:37,106 runs (UniPCRef / apply_freeu above)
    sched = UniPCRef() if scheduler == "unipc" else DDIMRef()
    sched.set_timesteps(num_inference_steps)
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds], 0) if do_cfg else prompt_embeds   # :234-236
    latents = latents.float() * sched.init_noise_sigma
    from . import control_ref as C
    pyr = C.bi_dir_feature_extractor(cn_sd, "feature_extractor.", controlnet_cond.float(), flow_cond.float()) if hoist else None
    # BASELINE config 4 ("dual ControlNet (flow + residual)"): a second, ResControlNet-shaped net (flow_resnet.py:52-144) with
    # its own controls; residuals of the two nets add (this package's rule — the reference never combines them in-repo)
    pyr2 = None
    if res_cn_sd is not None:
        rp = C.bi_dir_residue_extractor(res_cn_sd, "feature_extractor.", controlnet_cond[:, :3].float(), controlnet_cond[:, 3:].float(),
                                        flow_cond[:, :2].float(), flow_cond[:, 2:].float())
        rw = C.warp_extractor(res_cn_sd, "warp_extractor.", warp_cond.float())
        pyr2 = [a_ + b_ for a_, b_ in zip(rp, rw)]
        if res_conditioning_scale is None:
            res_conditioning_scale = controlnet_conditioning_scale
    nt = len(sched.timesteps)
    for i, t in enumerate(sched.timesteps):
        keep = 1.0 - float(i / nt < control_guidance_start or (i + 1) / nt > control_guidance_end)  # :292-295
        x_in = torch.cat([latents, latents], 0) if do_cfg else latents     # :313-320
        cc = torch.cat([controlnet_cond] * 2, 0) if do_cfg else controlnet_cond
        fc = torch.cat([flow_cond] * 2, 0) if do_cfg else flow_cond
        p = [torch.cat([q, q], 0) for q in pyr] if (pyr is not None and do_cfg) else pyr
        down, mid = M.dualflow_controlnet_forward(cn_sd, unet_cfg, x_in, t, ctx, cc, fc,
                                                  controlnet_conditioning_scale * keep, pyramid=p)
        if pyr2 is not None:
            p2 = [torch.cat([q, q], 0) for q in pyr2] if do_cfg else pyr2
            d2, m2 = M.dualflow_controlnet_forward(res_cn_sd, unet_cfg, x_in, t, ctx, cc, fc, res_conditioning_scale * keep,
                                                   pyramid=p2, residual_variant=True)
            down, mid = [a_ + b_ for a_, b_ in zip(down, d2)], mid + m2
        eps = M.unet_forward(unet_sd, unet_cfg, x_in, t, ctx, down, mid, freeu=freeu)   # :358-367
        if do_cfg:
            eu, et = eps.chunk(2)
            eps = eu + guidance_scale * (et - eu)                           # :370-372
        # :289 -> step(eta=, generator=).  diffusers draws the variance noise in the model output's dtype; `noise_dtype` lets a test ask for
        # the stream a bf16 pipeline sees (a CPU generator's bf16 draw is not its fp32 draw rounded)
        noise = torch.randn(tuple(eps.shape), generator=generator, dtype=noise_dtype).float() if eta else None
        latents = sched.step(eps, t, latents).float() if scheduler == "unipc" else sched.step(eps, t, latents, eta, noise)   # :375
    if output_type == "latent" or return_latents and vae_sd is None:
        return latents
    img = M.vae_decode(vae_sd, vae_cfg, latents / vae_cfg["scaling_factor"])   # :391
    img = (img / 2 + 0.5).clamp(0, 1)                                       # :397-398 postprocess
    return (img, latents) if return_latents else img
