"""`DualFlowControlNet` operator (controlnet/flownet.py) and its control pyramid (controlnet/extractors.py,
controlnet/control_utils.py), MI355X-native.

Call surface kept from the reference (pipeline.py:341-350):
    controlnet(sample=, timestep=, encoder_hidden_states=, controlnet_cond=[B,6,H,W], flow_cond=[B,4,H,W],
               conditioning_scale=, guess_mode=, return_dict=False) -> (list of 12 residuals, mid residual)

MI355X-first restructuring (results-preserving): the pyramid and the FDN gamma/beta convolutions depend only on
(controlnet_cond, flow_cond) — flownet.py:78 and control_utils.py:31-32 — so they are computed once per control
pair (`prepare_controls`) and reused by every denoising step and by both CFG halves; the reference recomputes
them every step.  The pyramid/splat stage runs in fp32 NCHW like the reference (softsplat.py:279); its output
is cast to bf16 NHWC once.  `holes.any()` (extractors.py:308, a device->host sync per scale) is replaced by an
unconditional select with the same result."""
from types import SimpleNamespace

import torch

from . import blocks, ops, weights
from .blocks import EncoderHalf
from .ops import PackedConv, PackedConvF32
from .unet import as_nchw, to_nhwc_bf16


class BiDirFeatureExtractor:
    """Bi_Dir_FeatureExtractor.forward — controlnet/extractors.py:264-316."""

    def __init__(self, sd, p, device):
        def c(k):
            return PackedConvF32(sd[p + k + ".weight"], sd[p + k + ".bias"], device)

        self.pre = {side: [c(f"{side}_pre_extractor.{i}") for i in (0, 2, 4, 6, 8)] for side in ("first", "last")}
        self.ext = {side: [c(f"extractors_{side}.{i}.0") for i in range(4)] for side in ("first", "last")}
        self.metric = [(c(f"wrapper.{i}.metric_net.0"), c(f"wrapper.{i}.metric_net.2")) for i in range(4)]
        self.zero = [c(f"zero_convs.{i}") for i in range(4)]

    def _pre(self, side, x):
        l = self.pre[side]
        x = ops.conv3x3_nchw_f32(x, l[0], 1, True)
        x = ops.conv3x3_nchw_f32(x, l[1], 2, True)
        x = ops.conv3x3_nchw_f32(x, l[2], 1, True)
        x = ops.conv3x3_nchw_f32(x, l[3], 2, True)
        return ops.conv3x3_nchw_f32(x, l[4], 1, True)

    def _warp(self, idx, feat, flow, occ):
        """FeatureWarperSoftsplat.forward — control_utils.py:49-72."""
        m = ops.conv3x3_nchw_f32(feat, self.metric[idx][0], 1, True)
        metric = ops.conv3x3_nchw_f32(m, self.metric[idx][1], 1, False)
        return ops.splat_soft(feat, flow, metric, mask=occ), metric

    def __call__(self, cond, flow):
        """cond [B,6,H,W], flow [B,4,H,W] fp32 device -> 4 pyramid levels, NCHW fp32."""
        h = cond.shape[-2]
        first, last = cond[:, 3:], cond[:, :3]                  # extractors.py:266-267 (slot swap kept)
        ffwd, fbwd = flow[:, :2], flow[:, 2:]
        f, l = self._pre("first", first), self._pre("last", last)
        outs = []
        for idx, r in enumerate((h // 8, h // 16, h // 32, h // 64)):
            f = ops.conv3x3_nchw_f32(f, self.ext["first"][idx], 2, True)
            l = ops.conv3x3_nchw_f32(l, self.ext["last"][idx], 2, True)
            flow_f = ops.flow_resize_normalize(ffwd, r, r)      # :286-287
            flow_b = ops.flow_resize_normalize(fbwd, r, r)
            occ_f = ops.occlusion_mask(flow_f, flow_b)          # :290-291
            occ_b = ops.occlusion_mask(flow_b, flow_f)
            wf, cf = self._warp(idx, f, flow_f, occ_f)          # :294-295
            wl, cb = self._warp(idx, l, flow_b, occ_b)
            fused = ops.fuse_warped(wf, wl, cf, cb, occ_f, occ_b)   # :297-310
            outs.append(ops.conv3x3_nchw_f32(fused, self.zero[idx], 1, False))
        return outs


class HipDualFlowControlNet:
    needs_warp_cond = False                # ResControlNet (flow_resnet.py:58) overrides

    def __init__(self, state_dict, config=None, device="cuda"):
        cfg = dict(weights.SD15_UNET_CONFIG if config is None else config)
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.config = SimpleNamespace(global_pool_conditions=False, cross_attention_dim=cfg["cross_attention_dim"],
                                      block_out_channels=cfg["block_out_channels"])
        sd = state_dict
        self.enc = EncoderHalf(sd, cfg, device)
        self.feature_extractor = self._make_extractor(sd, device)
        self.fdn = []
        for name in ("fdn64", "fdn32", "fdn16", "fdn08"):
            self.fdn.append((PackedConv(sd[name + ".conv_gamma.weight"], sd[name + ".conv_gamma.bias"], device),
                             PackedConv(sd[name + ".conv_beta.weight"], sd[name + ".conv_beta.bias"], device)))
        n_res = 1 + sum(cfg["layers_per_block"] + (1 if i != len(cfg["block_out_channels"]) - 1 else 0)
                        for i in range(len(cfg["block_out_channels"])))
        self.zero = [PackedConv(sd[f"controlnet_down_blocks.{i}.weight"], sd[f"controlnet_down_blocks.{i}.bias"], device)
                     for i in range(n_res)]
        self.zero_mid = PackedConv(sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"], device)
        self._ctx_key = None
        self._ctrl_key = None
        self.gamma_beta = None

    def _make_extractor(self, sd, device):
        return BiDirFeatureExtractor(sd, "feature_extractor.", device)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def set_context(self, ctx):
        key = (ctx.data_ptr(), tuple(ctx.shape), ctx._version)
        if key == self._ctx_key:
            return
        cb = ctx.to(device=self.device, dtype=torch.bfloat16).contiguous()
        for t in self.enc.transformers():
            t.set_context(cb)
        self._ctx_key = key
        self._ctx_keepalive = ctx

    # ---- step-invariant part ------------------------------------------------------------------------------
    def compute_pyramid(self, controlnet_cond, flow_cond, warp_cond=None):
        """flownet.py:78 -> (P64,P32,P16,P08) NCHW fp32 (exposed for parity tests)."""
        cond = controlnet_cond.to(device=self.device, dtype=torch.float32).contiguous()
        flow = flow_cond.to(device=self.device, dtype=torch.float32).contiguous()   # kept fp32: see module docstring
        return self.feature_extractor(cond, flow)

    def prepare_controls(self, controlnet_cond, flow_cond, warp_cond=None):
        """Pyramid + FDN gamma/beta (control_utils.py:31-32) at the controls' own batch size; cached by identity
        (pointer, shape and in-place version of EVERY control tensor, `warp_cond` of the ResControlNet included).  The
        gamma/beta buffers keep their addresses while the batch shape stays the same (captured hipGraphs read them); a
        re-allocation bumps `blocks.BUFFER_EPOCH`, which makes the pipeline drop its graphs."""
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in (controlnet_cond, flow_cond, warp_cond) if t is not None)
        if key == self._ctrl_key:
            return self.gamma_beta
        pyr = self.compute_pyramid(controlnet_cond, flow_cond, warp_cond)
        gb = []
        for lvl, (cg, cb) in zip(pyr, self.fdn):
            p = ops.nchw_f32_to_nhwc_bf16(lvl)
            gb.append((ops.conv(p, cg), ops.conv(p, cb)))
        old = self.gamma_beta
        if old is not None and all(o[0].shape == n[0].shape for o, n in zip(old, gb)):
            for o, n in zip(old, gb):       # keep addresses stable for captured hipGraphs
                o[0].copy_(n[0])
                o[1].copy_(n[1])
        else:
            self.gamma_beta = gb
            blocks.BUFFER_EPOCH[0] += 1
        self._ctrl_key = key
        self._ctrl_keepalive = (controlnet_cond, flow_cond, warp_cond)
        return self.gamma_beta

    # ---- per-step part ---------------------------------------------------------------------------------------
    def _fdn(self, sample, level):
        gamma, beta = self.gamma_beta[level]
        assert gamma.shape[1:3] == sample.shape[1:3], "FDN spatial mismatch (control_utils.py:30)"
        ab = ops.group_norm_ab(sample, None, None, self.cfg["groups"], 1e-5)
        return ops.fdn_modulate(sample, ab, gamma, beta)

    def forward_nhwc(self, x, t_dev, conditioning_scale=1.0, step_dev=None, cfg_shared=False, features_only=False):
        """x NHWC bf16 [n,h,w,4] -> (12 residuals, mid) NHWC bf16.  Needs set_context + prepare_controls first."""
        enc = self.enc
        temb = enc.temb(t_dev, x.shape[0], step_dev)
        sample = ops.conv(x, enc.conv_in)                                   # flownet.py:83
        sample = self._fdn(sample, 0)                                       # :84

        def hook(i, s):                                                     # :98-106 — fdn08 twice, residuals pre-FDN
            return self._fdn(s, min(i + 1, 3))

        sample, res = enc.run_down(sample, temb, after_block=hook, cfg_shared=cfg_shared)
        sample = enc.run_mid(sample, temb)                                  # :112-118
        if features_only:
            # the caller applies the zero-convs itself with the UNet skip as the GEMM's residual operand
            # (HipUNet2DConditionModel.decode_nhwc(control=...)): zero_conv(f) * scale + skip in one epilogue
            return res, sample
        down = [ops.conv(r, z, out_scale=conditioning_scale) for r, z in zip(res, self.zero)]   # :120-128
        mid = ops.conv(sample, self.zero_mid, out_scale=conditioning_scale)
        return down, mid

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond=None, flow_cond=None,
                conditioning_scale=1.0, guess_mode=False, return_dict=True, **kw):
        if controlnet_cond is None or flow_cond is None:
            raise ValueError("controlnet_cond [B,6,H,W] and flow_cond [B,4,H,W] are required")
        self.set_context(encoder_hidden_states)
        self.prepare_controls(controlnet_cond, flow_cond)
        p64 = self.gamma_beta[0][0]
        assert p64.shape[2] * 8 == controlnet_cond.shape[-1], "pyramid width mismatch (flownet.py:79)"
        t_dev = torch.as_tensor(timestep).to(device=self.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        x = to_nhwc_bf16(sample.to(self.device))
        down, mid = self.forward_nhwc(x, t_dev, float(conditioning_scale))
        down = [as_nchw(d) for d in down]
        mid = as_nchw(mid)
        if self.config.global_pool_conditions:                                  # flownet.py:130-132
            raise NotImplementedError("global_pool_conditions is False for this model")
        return (down, mid)

    __call__ = forward
