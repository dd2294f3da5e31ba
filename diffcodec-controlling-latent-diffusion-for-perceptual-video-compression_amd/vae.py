"""`AutoencoderKL` operator of the decode path, MI355X-native: `vae.decode(z, return_dict=False)[0]`
(pipeline.py:391) and `vae.encode(x).latent_dist.sample()` (train_controlnet.py:1081, pipeline.ipynb cell 7),
plus `.config.scaling_factor` / `.config.block_out_channels` (pipeline.py:129,391).

Inside: NHWC bf16, GroupNorm+SiLU folded into conv loads, nearest-2x upsample folded into the following conv,
the single-head d=512 mid attention as GEMM -> row softmax -> GEMM on the same MFMA kernel."""
from types import SimpleNamespace

import torch

from . import ops, weights
from .blocks import ResnetBlock
from .ops import PackedConv
from .unet import as_nchw, to_nhwc_bf16


class _RawWeight:
    """Activation matrix used as the weight operand of the GEMM ([rows][K] bf16) — attention scores / PV."""

    def __init__(self, t):
        self.w, self.bias, self.cout, self.cin, self.ksize, self.geglu, self.kind = t, None, t.shape[0], t.shape[1], 1, False, "igemm"
        self.ln_eps, self.colsum = None, None


class VaeAttention:
    """diffusers Attention (1 head, residual_connection=True) inside UNetMidBlock2D of the VAE."""

    def __init__(self, sd, p, device, groups):
        self.groups = groups
        self.norm = (sd[p + "group_norm.weight"].float().to(device), sd[p + "group_norm.bias"].float().to(device))
        self.q = PackedConv(sd[p + "to_q.weight"], sd[p + "to_q.bias"], device)
        self.k = PackedConv(sd[p + "to_k.weight"], sd[p + "to_k.bias"], device)
        self.v = PackedConv(sd[p + "to_v.weight"], sd[p + "to_v.bias"], device)
        self.o = PackedConv(sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"], device)

    def __call__(self, x):
        n, h, w, c = x.shape
        ab = ops.group_norm_ab(x, self.norm[0], self.norm[1], self.groups, 1e-6)
        y = ops.gn_apply(x, ab, silu=False).reshape(n, h * w, c)
        q, k, v = ops.linear(y, self.q), ops.linear(y, self.k), ops.linear(y, self.v)
        vt = ops.transpose_bf16(v)                                    # [n, c, L]
        outs = []
        for b in range(n):
            s = ops.linear(q[b], _RawWeight(k[b]), out_f32=True)      # [L, L] fp32 scores
            p = ops.softmax_rows(s, c ** -0.5)
            outs.append(ops.linear(p, _RawWeight(vt[b])))             # [L, c]
        a = outs[0].reshape(1, h * w, c) if n == 1 else torch.stack(outs, 0)
        return ops.linear(a, self.o, residual=x.reshape(n, h * w, c)).reshape(n, h, w, c)


class HipAutoencoderKL:
    def __init__(self, state_dict, config=None, device="cuda"):
        cfg = dict(weights.SD15_VAE_CONFIG if config is None else config)
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        state_dict = weights.remap_vae_attention_keys(state_dict)      # legacy query/key/value/proj_attn names
        self.config = SimpleNamespace(scaling_factor=cfg["scaling_factor"], block_out_channels=list(cfg["block_out_channels"]),
                                      latent_channels=cfg["latent_channels"])
        sd, g, boc = state_dict, cfg["groups"], cfg["block_out_channels"]

        def pc(k):
            return PackedConv(sd[k + ".weight"], sd[k + ".bias"], device)

        def res(p):
            return ResnetBlock(sd, p, device, g, 1e-6)

        nb = len(boc)
        # decoder
        self.post_quant = pc("post_quant_conv")
        self.d_conv_in = pc("decoder.conv_in")
        self.d_mid = (res("decoder.mid_block.resnets.0."), VaeAttention(sd, "decoder.mid_block.attentions.0.", device, g),
                      res("decoder.mid_block.resnets.1."))
        self.d_up = []
        for i in range(nb):
            self.d_up.append(dict(resnets=[res(f"decoder.up_blocks.{i}.resnets.{j}.") for j in range(cfg["layers_per_block"] + 1)],
                                  up=pc(f"decoder.up_blocks.{i}.upsamplers.0.conv") if i != nb - 1 else None))
        self.d_norm = (sd["decoder.conv_norm_out.weight"].float().to(device), sd["decoder.conv_norm_out.bias"].float().to(device))
        # conv_out (128 -> 3 at full resolution): on the MFMA tile kernel with a zero 4th output channel when the input width
        # allows it (the direct VALU conv takes 6.9 ms per 16 frames at 512x512, the tile kernel ~2); callers see channels 0-2
        wo, bo = sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"]
        self.d_out_channels = wo.shape[0]
        if wo.shape[0] == 3 and wo.shape[1] % 64 == 0:
            wo = torch.cat([wo, torch.zeros_like(wo[:1])], 0)
            bo = torch.cat([bo, torch.zeros_like(bo[:1])], 0)
            self.d_conv_out = PackedConv(wo, bo, device, mfma_small_cout=True)
        else:
            self.d_conv_out = pc("decoder.conv_out")
        # encoder (present in every SD-1.5 vae checkpoint; optional here)
        self.has_encoder = "encoder.conv_in.weight" in sd
        if self.has_encoder:
            self.e_conv_in = pc("encoder.conv_in")
            self.e_down = []
            for i in range(nb):
                self.e_down.append(dict(resnets=[res(f"encoder.down_blocks.{i}.resnets.{j}.") for j in range(cfg["layers_per_block"])],
                                        down=pc(f"encoder.down_blocks.{i}.downsamplers.0.conv") if i != nb - 1 else None))
            self.e_mid = (res("encoder.mid_block.resnets.0."), VaeAttention(sd, "encoder.mid_block.attentions.0.", device, g),
                          res("encoder.mid_block.resnets.1."))
            self.e_norm = (sd["encoder.conv_norm_out.weight"].float().to(device), sd["encoder.conv_norm_out.bias"].float().to(device))
            self.e_conv_out = pc("encoder.conv_out")
            self.quant = pc("quant_conv")

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # ---- decode ------------------------------------------------------------------------------------------
    def decode_nhwc(self, z):
        """z NHWC bf16 [n,h,w,4] (already divided by scaling_factor) -> image NHWC fp32 [n,8h,8w,3] in [-1,1]."""
        x = ops.conv(z, self.post_quant)
        x = ops.conv(x, self.d_conv_in)
        x = self.d_mid[0](x)
        x = self.d_mid[1](x)
        x = self.d_mid[2](x)
        for blk in self.d_up:
            for r in blk["resnets"]:
                x = r(x)
            if blk["up"] is not None:
                x = ops.conv(x, blk["up"], upsample=True, gn_part=True)
        ab = ops.group_norm_ab(x, self.d_norm[0], self.d_norm[1], self.cfg["groups"], 1e-6)
        return ops.conv(x, self.d_conv_out, gn_ab=ab, gn_silu=True, out_f32=True)[..., : self.d_out_channels]

    def decode(self, z, return_dict=True, generator=None):
        img = as_nchw(self.decode_nhwc(to_nhwc_bf16(z.to(self.device))))
        return (img,) if not return_dict else SimpleNamespace(sample=img)

    # ---- encode ------------------------------------------------------------------------------------------
    def encode_moments_nhwc(self, x):
        """x NHWC bf16 [n,H,W,3] in [-1,1] -> moments NHWC fp32 [n,H/8,W/8,2*latent] (mean | logvar)."""
        if not self.has_encoder:
            raise RuntimeError("this checkpoint carries no VAE encoder weights")
        x = ops.conv(x, self.e_conv_in)
        for blk in self.e_down:
            for r in blk["resnets"]:
                x = r(x)
            if blk["down"] is not None:
                x = ops.conv(x, blk["down"], stride=2, pad=0)            # F.pad(0,1,0,1) + stride-2 conv
        x = self.e_mid[0](x)
        x = self.e_mid[1](x)
        x = self.e_mid[2](x)
        ab = ops.group_norm_ab(x, self.e_norm[0], self.e_norm[1], self.cfg["groups"], 1e-6)
        m = ops.conv(x, self.e_conv_out, gn_ab=ab, gn_silu=True)         # bf16 [n,h,w,8]
        q = ops.conv(m, self.quant)                                      # 1x1, small-cin path, bf16
        return q

    def encode(self, x, return_dict=True):
        moments = self.encode_moments_nhwc(to_nhwc_bf16(x.to(self.device)))
        vae = self

        class _Dist:
            def __init__(self):
                self.moments_nhwc = moments

            def sample(self, generator=None):
                n, h, w, c2 = moments.shape
                noise = torch.randn((n, c2 // 2, h, w), generator=generator,
                                    device=generator.device if generator is not None else "cpu").to(vae.device)
                return ops.vae_sample_latents(ops_f32(moments), noise.float().contiguous(), 1.0)

            def mode(self):
                n, h, w, c2 = moments.shape
                zero = torch.zeros((n, c2 // 2, h, w), device=vae.device)
                return ops.vae_sample_latents(ops_f32(moments), zero, 1.0)

        d = _Dist()
        return (d,) if not return_dict else SimpleNamespace(latent_dist=d)


def ops_f32(t_bf16):
    """bf16 -> fp32 copy for the tiny moments tensor (plumbing, [n,h,w,8])."""
    return t_bf16.float().contiguous()
