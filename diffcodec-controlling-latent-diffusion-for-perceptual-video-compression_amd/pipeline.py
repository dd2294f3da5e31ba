"""`StableDiffusionDualFlowControlNetPipeline` — the decode sampling loop of the reference (pipeline.py:94-404)
behind the same constructor and `__call__` keywords, driving the MI355X-native operators.

Two execution modes, same results:
  * generic  — module-level calls exactly in the reference's order (controlnet(...), unet(...), scheduler.step),
               supports guess_mode, callbacks, any scheduler object with the reference's interface;
  * fused    — taken when the operators are this package's HIP modules with the DDIM scheduler and no callback:
               latents / model input / coefficient tables stay resident on the device, CFG combine + scheduler
               step are one kernel, the step-invariant control pyramid + FDN gamma/beta are computed once per call
               (the reference recomputes them every step), and one denoising step can be captured into a hipGraph
               and replayed (`enable_hip_graphs()`), removing the per-launch host cost.
"""
import inspect
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import ops
from .controlnet import HipDualFlowControlNet
from . import blocks
from .scheduler import DDIMScheduler, UniPCMultistepScheduler, randn_tensor
from .unet import HipUNet2DConditionModel
from .vae import HipAutoencoderKL


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """pipeline.py:19-75 — same dispatch and the same errors."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    if timesteps is not None:
        if "timesteps" not in set(inspect.signature(scheduler.set_timesteps).parameters.keys()):
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" timestep schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
        timesteps = scheduler.timesteps
        num_inference_steps = len(timesteps)
    elif sigmas is not None:
        if "sigmas" not in set(inspect.signature(scheduler.set_timesteps).parameters.keys()):
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" sigmas schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
        timesteps = scheduler.timesteps
        num_inference_steps = len(timesteps)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
        timesteps = scheduler.timesteps
    return timesteps, num_inference_steps


@dataclass
class StableDiffusionPipelineOutput:
    """pipeline.py:77-92"""
    images: Any
    nsfw_content_detected: Optional[List[bool]]


class StableDiffusionDualFlowControlNetPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, controlnet, scheduler, safety_checker=None,
                 feature_extractor=None, image_encoder=None, requires_safety_checker: bool = True):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        # `controlnet` is one DualFlowControlNet as in the reference (pipeline.py:110), or — BASELINE config 4, "dual ControlNet
        # (flow + residual)" — a list/tuple [DualFlowControlNet, ResControlNet] in diffusers' MultiControlNet convention:
        # every net sees the same sample/timestep/text and its own controls, and their residuals ADD.  The reference never
        # combines the two nets in-repo, so this rule is this package's and is unpinned.
        self.unet, self.controlnet, self.scheduler = unet, controlnet, scheduler
        self.safety_checker, self.feature_extractor, self.image_encoder = safety_checker, feature_extractor, image_encoder
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1) if vae is not None else 8
        self._interrupt = False
        self._progress = True
        self._use_graphs = False
        self._dual_stream = False
        self._cfg_shared = True
        self._graph_chunk = 1
        self._graphs = {}
        self._state = {}
        self.device = getattr(unet, "device", torch.device("cuda"))

    # ---- reference conveniences ---------------------------------------------------------------------------
    def to(self, device=None, *a, **k):
        return self

    def set_progress_bar_config(self, **kw):
        self._progress = not kw.get("disable", False)

    def enable_xformers_memory_efficient_attention(self):       # pipeline.py:138-142 — attention is already fused
        return None

    def enable_freeu(self, s1, s2, b1, b2):
        """validation.py:106 `pipe.enable_freeu(s1=0.9, s2=0.2, b1=1.2, b2=1.4)` -> UNet.enable_freeu."""
        if not hasattr(self.unet, "enable_freeu"):
            raise ValueError("The pipeline must have `unet` for using FreeU.")
        self.unet.enable_freeu(s1, s2, b1, b2)
        self._graphs.clear()

    def disable_freeu(self):
        self.unet.disable_freeu()
        self._graphs.clear()

    def enable_hip_graphs(self, flag=True, steps_per_graph=1):
        """Capture denoising steps as hipGraphs and replay them.  steps_per_graph > 1 puts that many consecutive steps
        (same control scale) into one graph: fewer graph launches per frame, which matters at one frame per pass."""
        self._use_graphs = bool(flag)
        self._graph_chunk = max(1, int(steps_per_graph))

    def enable_cfg_shared_prefix(self, flag=True):
        """Compute the layers ahead of the first text cross-attention once for both classifier-free-guidance halves
        (fused loop only; exact: the halves are bitwise identical up to there)."""
        self._cfg_shared = bool(flag)
        self._graphs.clear()

    def enable_dual_stream(self, flag=True):
        """Run the ControlNet and the UNet down path of each step concurrently on two HIP streams (fused loop only)."""
        self._dual_stream = bool(flag)
        self._graphs.clear()

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def _execution_device(self):
        return self.device

    def maybe_free_model_hooks(self):
        return None

    def run_safety_checker(self, image, device, dtype):
        if self.safety_checker is None:
            return image, None
        raise NotImplementedError("safety checker is None on the reference's decode path (validation.py:55)")

    # ---- prompt encoding (pipeline.py:223-236) -----------------------------------------------------------
    def encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                      prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
        def run_clip(texts):
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("no text_encoder/tokenizer: pass prompt_embeds= (and negative_prompt_embeds=)")
            tok = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt")
            with torch.no_grad():
                out = self.text_encoder(tok.input_ids.to(self.text_encoder.device), output_hidden_states=clip_skip is not None)
            if clip_skip is None:
                return out[0]
            return self.text_encoder.text_model.final_layer_norm(out[-1][-(clip_skip + 1)])

        if prompt_embeds is None:
            texts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = run_clip(texts)
        b = prompt_embeds.shape[0]
        if num_images_per_prompt != 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, 0)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            neg = [""] * b if negative_prompt is None else ([negative_prompt] * b if isinstance(negative_prompt, str) else list(negative_prompt))
            negative_prompt_embeds = run_clip(neg)
        if do_classifier_free_guidance and num_images_per_prompt != 1 and negative_prompt_embeds.shape[0] == b:
            negative_prompt_embeds = negative_prompt_embeds.repeat_interleave(num_images_per_prompt, 0)
        return prompt_embeds, negative_prompt_embeds

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = randn_tensor(shape, generator, torch.device("cpu") if generator is None else device, torch.float32)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
        return latents.to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        kw = {}
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    # ---- the call ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, controlnet_cond: torch.Tensor = None,
                 flow_cond: torch.Tensor = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, timesteps: Optional[List[int]] = None, sigmas: Optional[List[float]] = None,
                 guidance_scale: float = 7.5, negative_prompt: Optional[Union[str, List[str]]] = None,
                 num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: str = "pil", return_dict: bool = True, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0, control_guidance_end: Union[float, List[float]] = 1.0,
                 clip_skip: Optional[int] = None, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], **kwargs):
        device = self._execution_device
        self._interrupt = False
        nets = list(self.controlnet) if isinstance(self.controlnet, (list, tuple)) else [self.controlnet]
        warp_cond = kwargs.pop("warp_cond", None)            # [B,3,H,W]: extra control of a ResControlNet (flow_resnet.py:58)
        # 0) checks — pipeline.py:187-192
        if controlnet_cond is None or flow_cond is None:
            raise ValueError("Provide both controlnet_cond [B,6,H,W] and flow_cond [B,4,H,W].")
        if controlnet_cond.ndim != 4 or controlnet_cond.shape[1] != 6:
            raise ValueError(f"controlnet_cond must be [B,6,H,W], got {tuple(controlnet_cond.shape)}")
        if flow_cond.ndim != 4 or flow_cond.shape[1] != 4:
            raise ValueError(f"flow_cond must be [B,4,H,W], got {tuple(flow_cond.shape)}")
        if prompt_embeds is not None:
            base_batch = prompt_embeds.shape[0]
        elif isinstance(prompt, list):
            base_batch = len(prompt)
        else:
            base_batch = 1
        do_cfg = guidance_scale is not None and guidance_scale > 1.0                   # :202
        cgs = control_guidance_start[0] if isinstance(control_guidance_start, list) else control_guidance_start   # :206-214
        cge = control_guidance_end[0] if isinstance(control_guidance_end, list) else control_guidance_end
        # 1) prompts — :223-236
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds,
            None if not cross_attention_kwargs else cross_attention_kwargs.get("scale"), clip_skip)
        # 2) controls — :242-258
        b_ctrl, _, hc, wc = controlnet_cond.shape
        batch_size = base_batch * num_images_per_prompt
        if b_ctrl != batch_size:
            if b_ctrl == 1:
                controlnet_cond = controlnet_cond.expand(batch_size, -1, -1, -1).contiguous()
                flow_cond = flow_cond.expand(batch_size, -1, -1, -1).contiguous()
            else:
                raise ValueError(f"control batch={b_ctrl} vs prompt batch={batch_size} mismatch.")
        if height is None or width is None:
            height, width = hc, wc
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError("height/width must be divisible by 8.")
        # 3) timesteps & latents — :263-278
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, sigmas)
        latents = self.prepare_latents(batch_size, self.unet.config.in_channels, height, width, torch.float32, device, generator, latents)
        if getattr(self.unet.config, "time_cond_proj_dim", None) is not None:
            raise NotImplementedError("time_cond_proj_dim is None for SD-1.5 (pipeline.py:281-286)")
        nt = len(timesteps)
        controlnet_keep = [1.0 - float(i / nt < cgs or (i + 1) / nt > cge) for i in range(nt)]    # :292-295
        if isinstance(controlnet_conditioning_scale, (list, tuple)):           # one net: first entry (:206-214); several: one per net
            sc = [float(v) for v in controlnet_conditioning_scale]
            base_scales = [sc[0]] if len(nets) == 1 else sc
        else:
            base_scales = [float(controlnet_conditioning_scale)] * len(nets)
        if len(base_scales) != len(nets):
            raise ValueError(f"{len(nets)} ControlNets but {len(base_scales)} conditioning scales")
        if warp_cond is not None:                                              # same checks as the other controls (:187-192)
            if warp_cond.ndim != 4 or warp_cond.shape[1] != 3:
                raise ValueError(f"warp_cond must be [B,3,H,W], got {tuple(warp_cond.shape)}")
            if tuple(warp_cond.shape[-2:]) != (hc, wc):
                raise ValueError(f"warp_cond is {tuple(warp_cond.shape[-2:])} but the controls are {(hc, wc)}")
            if warp_cond.shape[0] != batch_size:
                if warp_cond.shape[0] != 1:
                    raise ValueError(f"warp_cond batch={warp_cond.shape[0]} vs prompt batch={batch_size} mismatch.")
                warp_cond = warp_cond.expand(batch_size, -1, -1, -1).contiguous()
        for net in nets:
            if getattr(net, "needs_warp_cond", False) and warp_cond is None:
                raise ValueError("a ResControlNet is registered: pass warp_cond [B,3,H,W] (flow_resnet.py:58).")

        # the fused (graph-replayed) loop knows two schedulers: DDIM with eta = 0 (eta > 0 draws noise per step: generic loop) and the
        # UniPC multistep solver validation.py:37 instantiates (it takes no eta: prepare_extra_step_kwargs drops it)
        sched_ok = (isinstance(self.scheduler, DDIMScheduler) and not eta) or isinstance(self.scheduler, UniPCMultistepScheduler)
        fused = (isinstance(self.unet, HipUNet2DConditionModel) and all(isinstance(nt_, HipDualFlowControlNet) for nt_ in nets)
                 and sched_ok and callback_on_step_end is None and not guess_mode)
        if fused:
            latents = self._denoise_fused(latents, prompt_embeds, negative_prompt_embeds, controlnet_cond, flow_cond, do_cfg,
                                          guidance_scale, base_scales, controlnet_keep, nets, warp_cond)
        else:
            latents = self._denoise_generic(latents, timesteps, prompt_embeds, negative_prompt_embeds, controlnet_cond, flow_cond,
                                            do_cfg, guidance_scale, base_scales, controlnet_keep, guess_mode, eta, generator,
                                            callback_on_step_end, callback_on_step_end_tensor_inputs, nets, warp_cond)
        # 5) decode & postprocess — :390-404
        if output_type == "latent":
            image, has_nsfw = latents, None
        else:
            z = ops.latents_to_model_input(latents.contiguous(), 1.0 / self.vae.config.scaling_factor, 1)
            img = self.vae.decode_nhwc(z) if isinstance(self.vae, HipAutoencoderKL) else \
                self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0].permute(0, 2, 3, 1).float().contiguous()
            has_nsfw = None
            o32, o8 = ops.postprocess_image(img, want_u8=(output_type == "pil"))
            if output_type == "pt":
                image = o32
            elif output_type == "np":
                image = o32.permute(0, 2, 3, 1).cpu().numpy()
            elif output_type == "pil":
                import PIL.Image
                image = [PIL.Image.fromarray(a) for a in o8.cpu().numpy()]
            else:
                raise ValueError(f"unknown output_type {output_type}")
        if not return_dict:
            return (image, has_nsfw)
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=has_nsfw)

    # ---- generic loop: the reference's order of module calls (pipeline.py:308-385) -------------------------
    def _denoise_generic(self, latents, timesteps, pe, npe, cond, flow, do_cfg, guidance, base_scales, keep, guess_mode, eta,
                         generator, callback, cb_inputs, nets, warp=None):
        device = self._execution_device
        if warp is not None:
            warp = warp.to(device)
        warp2 = torch.cat([warp, warp], 0) if (do_cfg and warp is not None) else warp
        text = torch.cat([npe, pe], 0).to(device) if do_cfg else pe.to(device)
        cond, flow = cond.to(device), flow.to(device)
        cond2 = torch.cat([cond, cond], 0) if do_cfg else cond
        flow2 = torch.cat([flow, flow], 0) if do_cfg else flow
        text_cond_only = text.chunk(2)[1].contiguous() if do_cfg else text
        extra = self.prepare_extra_step_kwargs(generator, eta)
        for i, t in enumerate(timesteps):
            if self.interrupt:
                continue
            x_in = torch.cat([latents, latents], 0) if do_cfg else latents
            x_in = self.scheduler.scale_model_input(x_in, t)
            if guess_mode and do_cfg:                                                   # :323-327
                c_in, c_text, cc, fc, wc = self.scheduler.scale_model_input(latents, t), text_cond_only, cond, flow, warp
            else:
                c_in, c_text, cc, fc, wc = x_in, text, cond2, flow2, warp2
            down = mid = None
            for net, base_scale in zip(nets, base_scales):
                extra_c = dict(warp_cond=wc) if getattr(net, "needs_warp_cond", False) else {}
                d_, m_ = net(sample=c_in, timestep=t, encoder_hidden_states=c_text, controlnet_cond=cc, flow_cond=fc,
                             conditioning_scale=base_scale * keep[i], guess_mode=guess_mode, return_dict=False, **extra_c)
                if down is None:
                    down, mid = d_, m_
                else:
                    from .rescontrolnet import combine_residuals
                    down, mid = combine_residuals((down, mid), (d_, m_))
            if guess_mode and do_cfg:                                                   # :353-355
                down = [torch.cat([torch.zeros_like(d), d], 0) for d in down]
                mid = torch.cat([torch.zeros_like(mid), mid], 0)
            noise_pred = self.unet(x_in, t, encoder_hidden_states=text, timestep_cond=None, cross_attention_kwargs=None,
                                   down_block_additional_residuals=down, mid_block_additional_residual=mid, return_dict=False)[0]
            unet_dtype = getattr(self.unet, "dtype", noise_pred.dtype)                  # the model dtype: what diffusers draws eta noise in
            #                                                                             (this U-Net hands its prediction over in fp32)
            if do_cfg:                                                                  # :370-372  eps_u + g (eps_t - eps_u)
                nu, nt_ = (t_.contiguous() for t_ in noise_pred.float().chunk(2, dim=0))
                g32 = np.float32(guidance)                                              # 1 - g in fp32, as the fused kernels form it
                noise_pred = ops.lincomb([(float(np.float32(1.0) - g32), nu), (float(g32), nt_)])
            if "eta" in extra:
                extra["noise_dtype"] = unet_dtype
            latents = self.scheduler.step(noise_pred, t, latents, **extra, return_dict=False)[0]   # :375
            if callback is not None:                                                    # :378-381
                loc = dict(latents=latents, prompt_embeds=pe, noise_pred=noise_pred)
                out = callback(self, i, t, {k: loc[k] for k in cb_inputs if k in loc})
                latents = out.pop("latents", latents)
        return latents

    # ---- fused loop --------------------------------------------------------------------------------------------
    def _denoise_fused(self, latents, pe, npe, cond, flow, do_cfg, guidance, base_scales, keep, nets, warp=None):
        device = self._execution_device
        unet, sched = self.unet, self.scheduler
        b, c, h, w = latents.shape
        st = self._state
        key = (b, c, h, w, do_cfg, tuple(id(n_) for n_ in nets))
        if st.get("key") != key:
            st.clear()
            st["key"] = key
            st["lat"] = torch.empty((b, c, h, w), device=device, dtype=torch.float32)
            st["x_in"] = torch.empty(((2 if do_cfg else 1) * b, h, w, c), device=device, dtype=torch.bfloat16)
            st["step"] = torch.zeros(1, device=device, dtype=torch.int32)
            # multistep state of the UniPC solver (two x0-predictions, the previous predictor's start sample): fp32 like the latents
            st["ms"] = [torch.zeros((b, c, h, w), device=device, dtype=torch.float32) for _ in range(3)]
            self._graphs.clear()
        st["lat"].copy_(latents)
        st["step"].zero_()
        # text context: cached per tensor identity inside the modules (cross-attention K/V are step-invariant)
        if st.get("ctx_src") is None or st["ctx_src"][0] is not pe or st["ctx_src"][1] is not npe:
            ctx = torch.cat([npe, pe], 0) if do_cfg else pe
            st["ctx"] = ctx.to(device=device, dtype=torch.bfloat16).contiguous()
            st["ctx_src"] = (pe, npe)
        unet.set_context(st["ctx"])
        for cn in nets:
            cn.set_context(st["ctx"])
            cn.prepare_controls(cond, flow, warp)             # hoisted: once per call, at batch B (shared by CFG halves)
        coef, ttab = sched.device_tables(device)
        if st.get("buf_epoch") != blocks.BUFFER_EPOCH[0]:
            # a module re-allocated a buffer captured graphs read (FDN gamma/beta, text K/V: another batch size went through
            # the generic loop, another pipe shares the modules, ...): every graph may hold a freed address
            self._graphs.clear()
            st["buf_epoch"] = blocks.BUFFER_EPOCH[0]
        if st.get("tables") != (id(sched), sched.table_version):     # new schedule: graphs captured on the old tables are stale
            self._graphs.clear()
            st["tables"] = (id(sched), sched.table_version)
            st["sched_keepalive"] = sched                            # keeps id(sched) unique while the key is in use
        ops.latents_to_model_input(st["lat"], 1.0, 2 if do_cfg else 1, out=st["x_in"])

        # CFG duplicates the latents (pipeline.py:313-320): the two batch halves only separate at the first text
        # cross-attention, so the layers before it are computed once (same values, see TransformerBlock.__call__)
        shared = bool(do_cfg and self._cfg_shared)
        unipc = isinstance(sched, UniPCMultistepScheduler)

        def run_nets(scales):
            """every ControlNet's features for this step: [(features[12], mid feature, zero convs, zero mid, scale), ...]"""
            out = []
            for cn, sc_ in zip(nets, scales):
                if sc_ == 0.0:
                    continue
                feats, midf = cn.forward_nhwc(st["x_in"], ttab, sc_, step_dev=st["step"], cfg_shared=shared, features_only=True)
                out.append((feats, midf, cn.zero, cn.zero_mid, sc_))
            return out

        def one_step(scales):
            if not any(scales):
                # controlnet_keep = 0 outside [control_guidance_start, end] (pipeline.py:292-295,354): the reference scales
                # every residual by 0, so adding them changes nothing — skip the ControlNet for this step
                eps = unet.forward_nhwc(st["x_in"], ttab, None, None, step_dev=st["step"], cfg_shared=shared)
            elif not self._dual_stream:
                eps = unet.forward_nhwc(st["x_in"], ttab, step_dev=st["step"], cfg_shared=shared, control=run_nets(scales))
            else:
                # The ControlNet and the UNet's down path are independent until the skip additions (flownet.py:83-124 vs
                # pipeline.py:358-367): two HIP streams, joined before the UNet mid block.  Every tensor crossing streams
                # is either persistent or produced before the join; the side stream re-synchronises at each step start.
                main = torch.cuda.current_stream()
                side = st.setdefault("side_stream", torch.cuda.Stream(device=device))
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ctl = run_nets(scales)
                sample, res, temb = unet.encode_nhwc(st["x_in"], ttab, step_dev=st["step"], cfg_shared=shared)
                main.wait_stream(side)
                for feats, midf, _, _, _ in ctl:
                    for t_ in feats + [midf]:
                        t_.record_stream(main)
                eps = unet.decode_nhwc(sample, list(res), temb, control=ctl)
            if unipc:
                ops.cfg_unipc_step(eps, st["lat"], st["ms"][0], st["ms"][1], st["ms"][2], st["x_in"], coef, st["step"],
                                   guidance if do_cfg else 1.0, do_cfg)
            else:
                ops.cfg_ddim_step(eps, st["lat"], st["x_in"], coef, st["step"], guidance if do_cfg else 1.0, do_cfg)

        nsteps = len(sched.timesteps)
        scales = [tuple(float(bs * keep[i]) for bs in base_scales) for i in range(nsteps)]
        i = 0
        while i < nsteps:
            if self.interrupt:
                break
            scale = scales[i]
            if not self._use_graphs:
                one_step(scale)
                i += 1
                continue
            if i == 0 and not st.get("warm"):
                one_step(scale)                               # eager warm-up (lazy kernel attributes) before any capture
                st["warm"] = True
                i += 1
                continue
            # several consecutive steps with the same control scale replay as ONE graph (the step counter lives on the device)
            chunk = 1
            while chunk < self._graph_chunk and i + chunk < nsteps and scales[i + chunk] == scale:
                chunk += 1
            gkey = (scale, float(guidance), shared, self._dual_stream, chunk, unipc)
            g = self._graphs.get(gkey)
            if g is None:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread of a multi-rank job polls events while we capture; in the default
                # global mode that invalidates the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for _ in range(chunk):
                        one_step(scale)
                self._graphs[gkey] = g                        # capture does not execute: replay below runs the steps
            g.replay()
            i += chunk
        return st["lat"].clone()
