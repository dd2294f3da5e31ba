"""`UNet2DConditionModel` operator of the decode path (call site pipeline.py:358-367), MI355X-native.

Same call surface as the diffusers module the reference passes to its pipeline:
    unet(sample, timestep, encoder_hidden_states=..., timestep_cond=None, cross_attention_kwargs=None,
         down_block_additional_residuals=[12 tensors], mid_block_additional_residual=tensor, return_dict=False)[0]
plus `.config.in_channels`, `.config.time_cond_proj_dim`, `.dtype`, `.device`.
Tensors cross this boundary as logical NCHW torch tensors; inside, everything is NHWC bf16 and every op is a
C-ABI launch (`ops`)."""
from types import SimpleNamespace

import torch

from . import ops, weights
from .blocks import EncoderHalf, ResnetBlock, TransformerBlock
from .ops import PackedConv


def to_nhwc_bf16(x):
    """boundary conversion: logical NCHW tensor (any float dtype / memory format) -> NHWC bf16 device tensor."""
    if x.dtype == torch.bfloat16 and x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous():
        return x.permute(0, 2, 3, 1)
    return ops.nchw_f32_to_nhwc_bf16(x.float().contiguous())


def as_nchw(x_nhwc):
    """zero-copy logical-NCHW view (channels_last memory) of an NHWC tensor."""
    return x_nhwc.permute(0, 3, 1, 2)


class HipUNet2DConditionModel:
    def __init__(self, state_dict, config=None, device="cuda"):
        cfg = dict(weights.SD15_UNET_CONFIG if config is None else config)
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.config = SimpleNamespace(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                                      time_cond_proj_dim=cfg.get("time_cond_proj_dim"),
                                      block_out_channels=cfg["block_out_channels"], cross_attention_dim=cfg["cross_attention_dim"])
        sd, g, boc = state_dict, cfg["groups"], cfg["block_out_channels"]
        nb = len(boc)
        up_res = [f"up_blocks.{i}.resnets.{j}." for i in range(nb) for j in range(cfg["layers_per_block"] + 1)]
        self.enc = EncoderHalf(sd, cfg, device, extra_resnets=up_res)
        self.up = []
        for i in range(nb):
            cross = cfg["down_cross"][nb - 1 - i]
            blk = dict(resnets=[], attns=[], up=None)
            for j in range(cfg["layers_per_block"] + 1):
                blk["resnets"].append(ResnetBlock(sd, f"up_blocks.{i}.resnets.{j}.", device, g, 1e-5))
                blk["attns"].append(TransformerBlock(sd, f"up_blocks.{i}.attentions.{j}.", device, cfg["num_heads"], g) if cross else None)
            if i != nb - 1:
                k = f"up_blocks.{i}.upsamplers.0.conv"
                blk["up"] = PackedConv(sd[k + ".weight"], sd[k + ".bias"], device)
            self.up.append(blk)
        self.norm_out = (sd["conv_norm_out.weight"].float().to(device), sd["conv_norm_out.bias"].float().to(device))
        self.conv_out = PackedConv(sd["conv_out.weight"], sd["conv_out.bias"], device, mfma_small_cout=True)
        self._ctx_key = None
        self.freeu = None                      # dict(s1, s2, b1, b2) when enabled (diffusers UNet.enable_freeu)

    def enable_freeu(self, s1, s2, b1, b2):
        self.freeu = dict(s1=float(s1), s2=float(s2), b1=float(b1), b2=float(b2))

    def disable_freeu(self):
        self.freeu = None

    # ---- reference-compatible niceties
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def transformers(self):
        yield from self.enc.transformers()
        for blk in self.up:
            for a in blk["attns"]:
                if a is not None:
                    yield a

    def set_context(self, ctx):
        """Cache the cross-attention K/V of a text embedding [B,77,768] (identity-keyed; step-invariant)."""
        key = (ctx.data_ptr(), tuple(ctx.shape), ctx._version)
        if key == self._ctx_key:
            return
        cb = ctx.to(device=self.device, dtype=torch.bfloat16).contiguous()
        for t in self.transformers():
            t.set_context(cb)
        self._ctx_key = key
        self._ctx_keepalive = ctx

    def encode_nhwc(self, x, t_dev, step_dev=None, cfg_shared=False):
        """First half (time embedding, conv_in, down blocks): independent of the ControlNet, so the fused pipeline can run
        it on one stream while the ControlNet runs on another.  Returns (sample, skips, temb)."""
        enc = self.enc
        temb = enc.temb(t_dev, x.shape[0], step_dev)
        sample = ops.conv(x, enc.conv_in)
        sample, res = enc.run_down(sample, temb, cfg_shared=cfg_shared)
        return sample, res, temb

    def decode_nhwc(self, sample, res, temb, down_res=None, mid_res=None, control=None):
        """control = (features[12], mid_feature, zero_convs[12], zero_mid, scale), or a list of such tuples, from
        `HipDualFlowControlNet.forward_nhwc(features_only=True)`: the ControlNet's zero-convs (flownet.py:120-128) run here
        with the skip tensor as the GEMM's residual operand — zero_conv(f) * scale + skip in one epilogue instead of a
        conv and a separate add (pipeline.py:364-365)."""
        enc = self.enc
        controls = [] if control is None else ([control] if isinstance(control, tuple) else list(control))
        if controls:                           # several ControlNets (config 4): their residuals add, one GEMM epilogue each
            assert down_res is None and mid_res is None
            for feats, _, zero, _, scale in controls:
                assert len(feats) == len(res) == len(zero)
                res = [ops.conv(f, z, out_scale=scale, residual=r, gn_part=True) for f, z, r in zip(feats, zero, res)]
        elif down_res is not None:
            assert len(down_res) == len(res)
            res = [ops.add_bf16(a, b) for a, b in zip(res, down_res)]          # pipeline.py:364 residual injection
        sample = enc.run_mid(sample, temb)
        if controls:
            for _, mid_feat, _, zero_mid, scale in controls:
                sample = ops.conv(mid_feat, zero_mid, out_scale=scale, residual=sample, gn_part=True)
        elif mid_res is not None:
            sample = ops.add_bf16(sample, mid_res)
        for bi, blk in enumerate(self.up):
            for r, a in zip(blk["resnets"], blk["attns"]):
                skip = res.pop()
                if self.freeu is not None and bi < 2:                          # apply_freeu: resolution_idx 0 and 1 only
                    sample = ops.freeu_backbone(sample, self.freeu["b1" if bi == 0 else "b2"])
                    skip = ops.freeu_lowfreq(skip, self.freeu["s1" if bi == 0 else "s2"])
                sample = r(sample, temb, x2=skip)                              # cat[sample, skip] read in place
                if a is not None:
                    sample = a(sample)
            if blk["up"] is not None:
                sample = ops.conv(sample, blk["up"], upsample=True, gn_part=True)   # nearest-2x fused into the conv load
        ab = ops.group_norm_ab(sample, self.norm_out[0], self.norm_out[1], self.cfg["groups"], 1e-5)
        return ops.conv(sample, self.conv_out, gn_ab=ab, gn_silu=True, out_f32=True)

    def forward_nhwc(self, x, t_dev, down_res=None, mid_res=None, step_dev=None, cfg_shared=False, control=None):
        """x NHWC bf16 [n,h,w,4]; t_dev fp32 device scalar (or table indexed by step_dev); residuals NHWC bf16.
        cfg_shared: the caller guarantees x[:n/2] == x[n/2:] (see TransformerBlock).  Returns eps NHWC fp32 [n,h,w,4]."""
        sample, res, temb = self.encode_nhwc(x, t_dev, step_dev, cfg_shared)
        return self.decode_nhwc(sample, list(res), temb, down_res, mid_res, control)

    def forward(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict=False, **kw):
        if timestep_cond is not None or cross_attention_kwargs:
            raise NotImplementedError("timestep_cond / cross_attention_kwargs are None for SD-1.5 (pipeline.py:281-286)")
        self.set_context(encoder_hidden_states)
        t_dev = torch.as_tensor(timestep).to(device=self.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        x = to_nhwc_bf16(sample.to(self.device))
        down = None if down_block_additional_residuals is None else [to_nhwc_bf16(d) for d in down_block_additional_residuals]
        mid = None if mid_block_additional_residual is None else to_nhwc_bf16(mid_block_additional_residual)
        eps = as_nchw(self.forward_nhwc(x, t_dev, down, mid))
        return (eps,) if not return_dict else SimpleNamespace(sample=eps)

    __call__ = forward
