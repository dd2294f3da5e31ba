"""DDIM scheduler object with the interface the reference pipeline uses (pipeline.py:263-265,306,320,375):
`set_timesteps`, `timesteps`, `scale_model_input`, `step`, `init_noise_sigma`, `order`.

Arithmetic follows diffusers' DDIMScheduler with the SD-1.5 `scheduler_config.json` (scaled_linear betas
0.00085..0.012 over 1000 steps, steps_offset=1, set_alpha_to_one=False, leading spacing, epsilon prediction,
clip_sample=False) [recalled — SURVEY.md §8 a16].  The host keeps only the tiny coefficient tables; the state
update itself is the device kernel `dc_cfg_ddim_step` (fp32 latents)."""
import numpy as np
import torch

from . import ops


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, set_alpha_to_one=False, clip_sample=False, prediction_type="epsilon", **unused):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon" or clip_sample:
            raise NotImplementedError("only the SD-1.5 DDIM configuration is implemented")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.timesteps = None
        self.num_inference_steps = None
        self._coef_dev = None

    @classmethod
    def from_config(cls, cfg):
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        if num_inference_steps == self.num_inference_steps and self.timesteps is not None:
            return                      # same schedule: keep the device tables (captured hipGraphs point at them)
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)
        self._coef_dev = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coefficients(self):
        """[steps][4] = sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev) in step order."""
        rows = []
        ratio = self.num_train_timesteps // self.num_inference_steps
        for t in self.timesteps.tolist():
            prev = t - ratio
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
            rows.append([(1 - a_t) ** 0.5, a_t ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5])
        return torch.tensor(rows, dtype=torch.float32)

    def device_tables(self, device):
        """(coef [steps,4] fp32, timesteps [steps] fp32) resident on the device for graph-replayed steps."""
        if self._coef_dev is None or self._coef_dev[0].device != torch.device(device):
            self._coef_dev = (self.coefficients().to(device), self.timesteps.float().to(device))
        return self._coef_dev

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True, **kw):
        """Generic (non-fused) entry: model_output/sample logical NCHW device tensors.  eta must be 0."""
        if eta:
            raise NotImplementedError("eta > 0 (stochastic DDIM) is not implemented")
        idx = (self.timesteps == int(timestep)).nonzero()
        if idx.numel() == 0:
            raise ValueError(f"timestep {timestep} is not in the current schedule")
        dev = sample.device
        coef, _ = self.device_tables(dev)
        step = torch.tensor([int(idx[0])], dtype=torch.int32, device=dev)
        lat = sample.float().contiguous().clone()
        eps = model_output.float().permute(0, 2, 3, 1).contiguous()
        b, c, h, w = lat.shape
        scratch = torch.empty((b, h, w, c), device=dev, dtype=torch.bfloat16)
        ops.cfg_ddim_step(eps, lat, scratch, coef, step, 1.0, False)
        return (lat,) if not return_dict else type("DDIMOut", (), {"prev_sample": lat})()
