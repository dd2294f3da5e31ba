"""DDIM scheduler object with the interface the reference pipeline uses (pipeline.py:263-265,306,320,375):
`set_timesteps`, `timesteps`, `scale_model_input`, `step`, `init_noise_sigma`, `order`.

Arithmetic follows diffusers' DDIMScheduler with the SD-1.5 `scheduler_config.json` (scaled_linear betas
0.00085..0.012 over 1000 steps, steps_offset=1, set_alpha_to_one=False, leading spacing, epsilon prediction,
clip_sample=False) [recalled — SURVEY.md §8 a16].  The host keeps only the tiny coefficient tables; the state
update itself is the device kernel `dc_cfg_ddim_step` (fp32 latents)."""
import numpy as np
import torch

from . import ops


def randn_tensor(shape, generator=None, device=None, dtype=torch.float32):
    """diffusers' `randn_tensor` as the reference reaches it (pipeline.py:269-278 `prepare_latents`, and the variance noise
    of DDIM's `step(eta > 0)`) [recalled]: the draw happens on the generator's device (a CPU generator draws on the host,
    the tensor is then moved), in the requested dtype; a LIST of generators draws one batch row per generator."""
    shape = tuple(int(v) for v in shape)
    device = torch.device("cpu") if device is None else torch.device(device)
    if isinstance(generator, (list, tuple)):
        if len(generator) == 1:
            generator = generator[0]
        elif len(generator) != shape[0]:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {shape[0]}. Make sure the batch size matches the length of the generators.")
        else:
            rows = [randn_tensor((1,) + shape[1:], g, device, dtype) for g in generator]
            return torch.cat(rows, 0)
    gdev = generator.device if isinstance(generator, torch.Generator) else device
    return torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)


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

    @staticmethod
    def _same_device(a, b):
        """torch.device('cuda') != torch.device('cuda:0') although both name the current device: compare the type and
        the index with `None` resolved to the current device."""
        a, b = torch.device(a), torch.device(b)
        if a.type != b.type:
            return False
        if a.type != "cuda":
            return True
        cur = torch.cuda.current_device() if (a.index is None or b.index is None) else 0
        return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)

    def device_tables(self, device):
        """(coef [steps,4] fp32, timesteps [steps] fp32) resident on the device for graph-replayed steps."""
        if self._coef_dev is None or not self._same_device(self._coef_dev[0].device, device):
            self._coef_dev = (self.coefficients().to(device), self.timesteps.float().to(device))
            # captured hipGraphs hold raw pointers into these tables: consumers key on this counter, never on id() of the
            # tensors (a freed table's id can be handed to its successor)
            self.table_version = getattr(self, "table_version", 0) + 1
        return self._coef_dev

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True, noise_dtype=None, **kw):
        """Generic (non-fused) entry: model_output/sample logical NCHW device tensors.
        eta > 0 (pipeline.py:289 forwards it through prepare_extra_step_kwargs) is diffusers' stochastic DDIM [recalled]:
        sigma_t = eta * sqrt((1-a_prev)/(1-a_t)) * sqrt(1 - a_t/a_prev); x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev-sigma_t^2) eps
        + sigma_t * noise, the noise drawn with the caller's generator on the generator's device (CPU generator -> CPU draw,
        then moved: `randn_tensor`).  One `dc_lincomb4_f32` launch on fp32 state.
        `noise_dtype`: the dtype the variance noise is DRAWN in.  diffusers draws it in `model_output.dtype`, and in the reference the
        CFG combine stays in the model dtype (pipeline.py:370-372), so that is the U-Net's output dtype (bf16 / fp16 at configs[1]);
        this repo's CFG combine is fp32, so the pipeline passes the U-Net's dtype here explicitly — a CPU generator yields a different
        stream for a bf16 draw than for an fp32 one.  Default: `model_output.dtype`.  Parity unpinned (diffusers not importable)."""
        idx = (self.timesteps == int(timestep)).nonzero()
        if idx.numel() == 0:
            raise ValueError(f"timestep {timestep} is not in the current schedule")
        dev = sample.device
        if eta:
            t = int(timestep)
            prev = t - self.num_train_timesteps // self.num_inference_steps
            a_t = float(self.alphas_cumprod[t])
            a_p = float(self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod)
            var = (1.0 - a_p) / (1.0 - a_t) * (1.0 - a_t / a_p)
            std = float(eta) * var ** 0.5
            noise = randn_tensor(model_output.shape, generator, dev, noise_dtype or model_output.dtype).float()
            x = sample.float().contiguous()
            eps = model_output.float().contiguous()
            c_x = (a_p / a_t) ** 0.5
            c_eps = -(a_p * (1.0 - a_t) / a_t) ** 0.5 + (1.0 - a_p - std * std) ** 0.5
            lat = ops.lincomb([(c_x, x), (c_eps, eps), (std, noise.contiguous())])
            return (lat,) if not return_dict else type("DDIMOut", (), {"prev_sample": lat})()
        coef, _ = self.device_tables(dev)
        step = torch.tensor([int(idx[0])], dtype=torch.int32, device=dev)
        lat = sample.float().contiguous().clone()
        eps = model_output.float().permute(0, 2, 3, 1).contiguous()
        b, c, h, w = lat.shape
        scratch = torch.empty((b, h, w, c), device=dev, dtype=torch.bfloat16)
        ops.cfg_ddim_step(eps, lat, scratch, coef, step, 1.0, False)
        return (lat,) if not return_dict else type("DDIMOut", (), {"prev_sample": lat})()


class UniPCMultistepScheduler:
    """The scheduler the reference actually instantiates (validation.py:37, `UniPCMultistepScheduler.from_pretrained(base,
    subfolder="scheduler")`): diffusers' UniPC with its defaults on top of the SD-1.5 scheduler_config — solver_order 2,
    bh2, predict_x0, lower_order_final, epsilon prediction, leading spacing, steps_offset 1, final sigma 0 [recalled; the
    reference holds no fixture -> parity unpinned].  All coefficients are host scalars; every tensor update is one
    `dc_lincomb4_f32` launch on fp32 state."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 solver_order=2, prediction_type="epsilon", predict_x0=True, solver_type="bh2", lower_order_final=True,
                 timestep_spacing="leading", steps_offset=1, final_sigmas_type="zero", **unused):
        if (beta_schedule, prediction_type, predict_x0, solver_type, timestep_spacing, final_sigmas_type) != \
                ("scaled_linear", "epsilon", True, "bh2", "leading", "zero") or solver_order not in (1, 2):
            raise NotImplementedError("only the configuration the reference runs is implemented")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.solver_order, self.lower_order_final = solver_order, lower_order_final
        self.timesteps = None
        self.num_inference_steps = None
        self._coef_dev = None

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        n = num_inference_steps
        ratio = self.num_train_timesteps // (n + 1)
        ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        ac = self.alphas_cumprod.double().numpy()
        sig = ((1 - ac) / ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]])
        if self.timesteps is None or self.num_inference_steps != n:
            self._coef_dev = None                  # new schedule: new device tables (same one: captured hipGraphs point at them)
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums, self.last_sample, self.step_index, self.this_order = 0, None, None, 1

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha = 1.0 / (sigma * sigma + 1.0) ** 0.5
        return alpha, sigma * alpha

    @staticmethod
    def _lam(alpha, sigma):
        return np.log(alpha) - (np.log(sigma) if sigma > 0 else -np.inf)

    def _rhos_and_h(self, order, lam_t, lam_s0, lam_prev):
        """shared coefficient algebra of uni_p / uni_c: returns (h_phi_1, B_h, rks, R, b)."""
        h = lam_t - lam_s0
        rks = [(lam_prev[i] - lam_s0) / h for i in range(order - 1)] + [1.0]
        hh = -h
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0 if np.isfinite(hh) else -1.0
        b_h = np.expm1(hh)
        rr, bb, fact = [], [], 1.0
        for i in range(1, order + 1):
            rr.append([rk ** (i - 1) for rk in rks])
            bb.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / fact if np.isfinite(hh) else -1.0 / fact
        return h_phi_1, b_h, rks, np.array(rr, dtype=np.float64), np.array(bb, dtype=np.float64)

    UNIPC_ROW = 12          # floats per device table row (DC_UNIPC_ROW of csrc/elementwise.hip)

    def _plan(self, i, prev_order, lower_order_nums):
        """Host scalars of step i given the solver's warm-up state (`this_order` left by the previous step, `lower_order_nums`):
        dict(ca, ce: x0 = ca x + ce eps; corr: coefficients of (last_sample, m0, m_t, m1 | None) or None; order: the order the
        predictor runs at (and the next corrector); pred: coefficients of (x, m_t, m1 | None)).  The ONE place the UniPC algebra
        lives: `step` (generic loop, one dc_lincomb4_f32 launch per update) and `device_tables` (fused loop, dc_cfg_unipc_step)
        both read it, so the two loops apply the same fp32 coefficients in the same order."""
        a_i, s_i = self._alpha_sigma(self.sigmas[i])
        plan = dict(ca=1.0 / a_i, ce=-s_i / a_i, corr=None)
        if i > 0:                                                                 # ---- corrector (multistep_uni_c_bh_update)
            order = prev_order
            a_t, s_t = a_i, s_i
            a_s, s_s = self._alpha_sigma(self.sigmas[i - 1])
            lam_prev = [self._lam(*self._alpha_sigma(self.sigmas[i - (k + 1)])) for k in range(1, order)]
            h_phi_1, b_h, rks, rr, bb = self._rhos_and_h(order, self._lam(a_t, s_t), self._lam(a_s, s_s), lam_prev)
            rhos = np.array([0.5]) if order == 1 else np.linalg.solve(rr, bb)
            t_x, t_m0, t_mt, t_m1 = s_t / s_s, -a_t * h_phi_1, -a_t * b_h * rhos[-1], None
            t_m0 += a_t * b_h * rhos[-1]                                            # D1_t = m_t - m0
            if order == 2:
                c1 = -a_t * b_h * rhos[0] / rks[0]                                  # corr_res = rho * (m1 - m0) / rk
                t_m0 -= c1
                t_m1 = c1
            plan["corr"] = (t_x, t_m0, t_mt, t_m1)
        this_order = min(self.solver_order, len(self.timesteps) - i) if self.lower_order_final else self.solver_order
        order = plan["order"] = min(this_order, lower_order_nums + 1)
        # ---- predictor (multistep_uni_p_bh_update)
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        a_s, s_s = a_i, s_i
        lam_prev = [self._lam(*self._alpha_sigma(self.sigmas[i - k])) for k in range(1, order)]
        h_phi_1, b_h, rks, rr, bb = self._rhos_and_h(order, self._lam(a_t, s_t), self._lam(a_s, s_s), lam_prev)
        c_x, c_m0, c_m1 = s_t / s_s, -a_t * h_phi_1, None
        if order == 2:
            c = -a_t * b_h * 0.5 / rks[0]                                           # rhos_p = [0.5]; D1 = (m1 - m0) / rk
            c_m0 -= c
            c_m1 = c
        plan["pred"] = (c_x, c_m0, c_m1)
        return plan

    def coefficients(self):
        """[steps][UNIPC_ROW] fp32 rows of dc_cfg_unipc_step, the warm-up state machine unrolled over the schedule."""
        rows, prev_order, lower = [], 1, 0
        for i in range(len(self.timesteps)):
            pl = self._plan(i, prev_order, lower)
            corr, pred = pl["corr"], pl["pred"]
            flags = (1 if corr is not None else 0) | (2 if corr is not None and corr[3] is not None else 0) | (4 if pred[2] is not None else 0)
            cz = [0.0 if v is None else float(v) for v in (corr if corr is not None else (0.0, 0.0, 0.0, None))]
            rows.append([pl["ca"], pl["ce"], float(flags)] + cz + [float(pred[0]), float(pred[1]), 0.0 if pred[2] is None else float(pred[2]), 0.0, 0.0])
            prev_order = pl["order"]
            lower = min(lower + 1, self.solver_order)
        return torch.tensor(rows, dtype=torch.float32)

    def device_tables(self, device):
        """(coef [steps, UNIPC_ROW] fp32, timesteps [steps] fp32) resident on the device for graph-replayed steps."""
        if self._coef_dev is None or not DDIMScheduler._same_device(self._coef_dev[0].device, device):
            self._coef_dev = (self.coefficients().to(device), self.timesteps.float().to(device))
            self.table_version = getattr(self, "table_version", 0) + 1     # captured hipGraphs key on this counter
        return self._coef_dev

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        if self.step_index is None:
            idx = (self.timesteps == int(timestep)).nonzero()
            if idx.numel() == 0:
                raise ValueError(f"timestep {timestep} is not in the current schedule")
            self.step_index = int(idx[0])
        i = self.step_index
        x = sample.float().contiguous()
        eps = model_output.float().contiguous()
        if eps.shape != x.shape:
            raise ValueError("model_output and sample must have the same shape")
        pl = self._plan(i, self.this_order, self.lower_order_nums)
        m_t = ops.lincomb([(pl["ca"], x), (pl["ce"], eps)])                         # convert_model_output: x0 prediction
        if pl["corr"] is not None and self.last_sample is not None:
            t_x, t_m0, t_mt, t_m1 = pl["corr"]
            m1 = self.model_outputs[-2] if t_m1 is not None else None
            x = ops.lincomb([(t_x, self.last_sample), (t_m0, self.model_outputs[-1]), (t_mt, m_t), (t_m1 if m1 is not None else 0.0, m1)])
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = m_t
        self.this_order = pl["order"]
        self.last_sample = x
        c_x, c_m0, c_m1 = pl["pred"]
        m1 = self.model_outputs[-2] if c_m1 is not None else None
        prev = ops.lincomb([(c_x, x), (c_m0, m_t), (c_m1 if m1 is not None else 0.0, m1)])
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return (prev,) if not return_dict else type("UniPCOut", (), {"prev_sample": prev})()
