"""Device building blocks shared by the UNet, the ControlNet encoder and the VAE: the diffusers modules the
reference instantiates (ResnetBlock2D, Transformer2DModel/BasicTransformerBlock, Down/Upsample2D, time embedding),
re-expressed as sequences of C-ABI launches on NHWC bf16 activations.  Host code only: no arithmetic here.

Weight keys are the diffusers state-dict keys; topology follows the published SD-1.5 modules
as called from controlnet/flownet.py:74-124 and pipeline.py:358-367."""
import torch

from . import ops
from .ops import PackedConv


# Transformer2DModel.norm + proj_in: GroupNorm is applied by its own HBM-bound pass and the 1x1 runs on the LDS-DMA GEMM, whose
# epilogue also writes the LayerNorm row statistics norm1 needs.  The alternative (GroupNorm affine inside the gather GEMM's load
# stage + a separate row-statistics pass) was the round-1 choice above 65,536 rows; measured again in round 2 on one box it costs
# +5.5 ms per 16-frame step (igemm 26 -> 14 ms against +3.5 ms gn_apply and +5.8 ms GEMM).  Module constant (tools assign it for an A/B);
# nothing here reads the environment.
PROJ_IN_FUSE_MIN_ROWS = 1 << 62
# ... and the round-3 form of the same idea that does pay: where the K = 320 row-panel GEMM takes proj_in (64x64 maps, >= 65,536 rows),
# the affine is applied to the activation panel the kernel already holds in registers — no gather GEMM, statistics epilogue kept.
PROJ_IN_GN_ON_LOAD = True
# LayerNorm folded into the following linear's weights + epilogue (ops.PackedConv(ln=...)): removes the three standalone
# LayerNorm passes of every BasicTransformerBlock.  False keeps the separate dc_layernorm_bf16 launches (A/B from tools/).
LN_FOLD = True


# Captured hipGraphs hold raw addresses of the step-invariant buffers the modules cache (cross-attention K/V of the text, FDN
# gamma/beta).  The modules refill those buffers IN PLACE while the shape stays the same; whenever one has to be re-allocated
# (another batch size) this epoch moves on, and the pipeline drops its graphs when it sees a new value.
BUFFER_EPOCH = [0]


def _f32(sd, key, device):
    return sd[key].detach().float().contiguous().to(device)


class TimeEmbedding:
    """time_proj + time_embedding MLP (flownet.py:74-75) and ALL ResnetBlock2D.time_emb_proj projections of a model
    batched into one GEMM: out[n, sum(cout)] fp32, sliced per block (each slice is a `row_add` of that block's conv1).
    Only SiLU(emb) is ever consumed in SD-1.5, so the SiLU is fused into linear_2's epilogue."""

    def __init__(self, sd, device, ch0, resnet_prefixes):
        self.ch0 = ch0
        self.l1 = PackedConv(sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"], device)
        self.l2 = PackedConv(sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"], device)
        ws, bs, self.slices, off = [], [], {}, 0
        for p in resnet_prefixes:
            w = sd[p + "time_emb_proj.weight"]
            ws.append(w)
            bs.append(sd[p + "time_emb_proj.bias"])
            self.slices[p] = (off, off + w.shape[0])
            off += w.shape[0]
        self.proj = PackedConv(torch.cat(ws, 0), torch.cat(bs, 0), device)

    def __call__(self, t_dev, n, step_dev=None):
        e = ops.timestep_embedding(t_dev, n, self.ch0, step_dev)
        e = ops.f32_to_bf16(e)
        e = ops.linear(e, self.l1, act=1)
        e = ops.linear(e, self.l2, act=1)                       # SiLU(emb)
        self.all = ops.linear(e, self.proj, out_f32=True)       # [n, total]
        return self

    def slice(self, prefix, rows=None):
        a, b = self.slices[prefix]
        return self.all[:rows, a:b]


class ResnetBlock:
    """diffusers ResnetBlock2D: GN -> SiLU -> conv3x3 (+temb) -> GN -> SiLU -> conv3x3, + (1x1 shortcut of) input.
    Both GroupNorm+SiLU are folded into the following conv's load stage; the channel concat of a UNet skip
    connection (x2) is read in place by the GN statistics, conv1 and the shortcut."""

    def __init__(self, sd, p, device, groups, eps):
        self.p, self.groups, self.eps = p, groups, eps
        self.n1 = (_f32(sd, p + "norm1.weight", device), _f32(sd, p + "norm1.bias", device))
        self.n2 = (_f32(sd, p + "norm2.weight", device), _f32(sd, p + "norm2.bias", device))
        self.conv1 = PackedConv(sd[p + "conv1.weight"], sd[p + "conv1.bias"], device)
        self.conv2 = PackedConv(sd[p + "conv2.weight"], sd[p + "conv2.bias"], device)
        self.shortcut = None
        if p + "conv_shortcut.weight" in sd:
            self.shortcut = PackedConv(sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"], device)

    def __call__(self, x, temb=None, x2=None):
        ab1 = ops.group_norm_ab(x, self.n1[0], self.n1[1], self.groups, self.eps, x2=x2)
        h = ops.conv_gn_silu(x, self.conv1, ab1, x2=x2, row_add=None if temb is None else temb.slice(self.p, x.shape[0]),
                             gn_part=True)                                  # norm2 reads the statistics from conv1's epilogue
        ab2 = ops.group_norm_ab(h, self.n2[0], self.n2[1], self.groups, self.eps)
        if self.shortcut is not None:
            sc = ops.conv(x, self.shortcut, x2=x2)
        else:
            assert x2 is None
            sc = x
        return ops.conv_gn_silu(h, self.conv2, ab2, residual=sc, gn_part=True)   # the next block's GroupNorm input


class TransformerBlock:
    """diffusers Transformer2DModel with one BasicTransformerBlock (SD-1.5: conv proj_in/out, GEGLU feed-forward).
    to_q/to_k/to_v of the self-attention are one fused GEMM; the cross-attention K/V of the (step-invariant) text
    embedding are computed once per `set_context` instead of every step."""

    def __init__(self, sd, p, device, heads, groups):
        self.heads, self.groups = heads, groups
        self.norm = (_f32(sd, p + "norm.weight", device), _f32(sd, p + "norm.bias", device))
        self.proj_in = PackedConv(sd[p + "proj_in.weight"], sd[p + "proj_in.bias"], device)
        self.proj_out = PackedConv(sd[p + "proj_out.weight"], sd[p + "proj_out.bias"], device)
        q = p + "transformer_blocks.0."
        self.ln = [(_f32(sd, q + f"norm{i}.weight", device), _f32(sd, q + f"norm{i}.bias", device)) for i in (1, 2, 3)]
        self.fold = LN_FOLD

        def ln(i):                           # nn.LayerNorm(eps=1e-5) parameters folded into the consumer's weights
            return (sd[q + f"norm{i}.weight"], sd[q + f"norm{i}.bias"], 1e-5) if self.fold else None

        self.qkv1 = PackedConv(torch.cat([sd[q + "attn1.to_q.weight"], sd[q + "attn1.to_k.weight"], sd[q + "attn1.to_v.weight"]], 0), None, device, ln=ln(1))
        self.out1 = PackedConv(sd[q + "attn1.to_out.0.weight"], sd[q + "attn1.to_out.0.bias"], device)
        self.q2 = PackedConv(sd[q + "attn2.to_q.weight"], None, device, ln=ln(2))
        self.kv2 = PackedConv(torch.cat([sd[q + "attn2.to_k.weight"], sd[q + "attn2.to_v.weight"]], 0), None, device)
        self.out2 = PackedConv(sd[q + "attn2.to_out.0.weight"], sd[q + "attn2.to_out.0.bias"], device)
        self.ff1 = PackedConv(sd[q + "ff.net.0.proj.weight"], sd[q + "ff.net.0.proj.bias"], device, geglu=True, ln=ln(3))
        self.ff2 = PackedConv(sd[q + "ff.net.2.weight"], sd[q + "ff.net.2.bias"], device)
        self.c = self.proj_in.cout
        self.kv_ctx = None

    def set_context(self, ctx_bf16):
        """ctx [B,77,768] bf16 -> cached fused K|V [B,77,2C]."""
        new = ops.linear(ctx_bf16, self.kv2)
        if self.kv_ctx is not None and self.kv_ctx.shape == new.shape:
            self.kv_ctx.copy_(new)          # keep the address stable for captured hipGraphs
        else:
            self.kv_ctx = new
            BUFFER_EPOCH[0] += 1

    def __call__(self, x, cfg_shared=False):
        """cfg_shared: x is ONE half [B] of a classifier-free-guidance batch whose two halves are identical up to here
        (same latents, same timestep; pipeline.py:313-320 duplicates them).  Everything before the text cross-attention
        is then the same for both halves and is computed once; the halves separate at attn2's K/V (the contexts
        [uncond | cond] of `set_context`).  Returns the full [2B] batch.  Same values as running the duplicated batch."""
        n, h, w, c = x.shape
        fold = self.fold
        parts = ops.row_stats_parts(c) if fold else 0

        def stats_buf(rows):                 # row statistics (sum, sum of squares) written by the producing GEMM's epilogue
            return torch.empty((rows, parts, 2), device=x.device, dtype=torch.float32) if fold else None

        ab = ops.group_norm_ab(x, self.norm[0], self.norm[1], self.groups, 1e-6)
        if PROJ_IN_GN_ON_LOAD and self.proj_in.bias is not None and ops.rowpanel_takes(n * h * w, h * w, c, self.proj_in.cout):
            # 64x64 maps at decode batch sizes: the row-panel GEMM applies the GroupNorm affine to its register-resident activation panel
            # (same arithmetic as the standalone pass): the normalized tensor is never written or re-read, the statistics epilogue stays
            st = stats_buf(n * h * w)
            t = ops.conv(x, self.proj_in, gn_ab=ab, gn_silu=False, stats_out=st).reshape(n, h * w, c)
        elif n * h * w > PROJ_IN_FUSE_MIN_ROWS:                  # big maps: GroupNorm applied inside the GEMM's load stage
            t = ops.conv(x, self.proj_in, gn_ab=ab, gn_silu=False).reshape(n, h * w, c)
            st = ops.row_stats(t) if fold else None              # that kernel has no statistics epilogue: one read-only pass
        else:                                                    # small maps: a separate pass + the LDS-DMA GEMM is faster
            st = stats_buf(n * h * w)
            t = ops.conv(ops.gn_apply(x, ab), self.proj_in, stats_out=st).reshape(n, h * w, c)
        # self-attention (norm1 folded into to_q/k/v)
        qkv = ops.linear(t, self.qkv1, ln_partials=(st, 1e-5)) if fold else ops.linear(ops.layer_norm(t, *self.ln[0]), self.qkv1)
        a = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.heads)
        st = stats_buf(n * h * w)
        t = ops.linear(a, self.out1, residual=t, stats_out=st)
        # cross-attention (norm2 folded into to_q)
        q = ops.linear(t, self.q2, ln_partials=(st, 1e-5)) if fold else ops.linear(ops.layer_norm(t, *self.ln[1]), self.q2)
        kv = self.kv_ctx
        if not cfg_shared:
            a = ops.attention(q, kv[..., :c], kv[..., c:], self.heads)
            st = stats_buf(n * h * w)
            t = ops.linear(a, self.out2, residual=t, stats_out=st)
        else:
            assert kv.shape[0] == 2 * n, "cfg_shared needs the [uncond | cond] context batch"
            a = torch.empty((2 * n, h * w, c), device=x.device, dtype=x.dtype)
            t2 = torch.empty_like(a)
            st = stats_buf(2 * n * h * w)
            for half in (0, 1):                                  # same queries and residual, that half's text keys/values
                sl = slice(half * n, (half + 1) * n)
                ops.attention(q, kv[sl, :, :c], kv[sl, :, c:], self.heads, out=a[sl])
                ops.linear(a[sl], self.out2, residual=t, out=t2[sl],
                           stats_out=st[half * n * h * w:(half + 1) * n * h * w] if fold else None)
            t = t2
        # GEGLU feed-forward (norm3 folded into ff.net.0.proj)
        f = ops.linear(t, self.ff1, ln_partials=(st, 1e-5)) if fold else ops.linear(ops.layer_norm(t, *self.ln[2]), self.ff1)
        t = ops.linear(f, self.ff2, residual=t)
        if not cfg_shared:
            return ops.conv(t.reshape(n, h, w, c), self.proj_out, residual=x, gn_part=True)
        out = torch.empty((2 * n, h, w, c), device=x.device, dtype=x.dtype)
        parts = []
        for half in (0, 1):
            sl = slice(half * n, (half + 1) * n)
            y = ops.conv(t[sl].reshape(n, h, w, c), self.proj_out, residual=x, out=out[sl], gn_part=True)
            parts.append(getattr(y, "gn_part", None))
        # the GroupNorm partials of the epilogue ([chunks, N, C, 2], per sample) travel with the tensor exactly as on the unshared
        # path — without them the next ResnetBlock would measure the STORED bf16 tensor instead (statistics of the rounded
        # values: a different last place, found by tests/test_gpu_round3.py's bit-exactness check of the shared prefix)
        if parts[0] is not None and parts[1] is not None and parts[0].shape[0] == parts[1].shape[0]:
            out.gn_part = torch.cat(parts, 1)
        return out


class EncoderHalf:
    """conv_in + time embedding + down blocks + mid block shared by UNet2DConditionModel and ControlNetModel."""

    def __init__(self, sd, cfg, device, extra_resnets=()):
        self.cfg = cfg
        boc, g = cfg["block_out_channels"], cfg["groups"]
        self.conv_in = PackedConv(sd["conv_in.weight"], sd["conv_in.bias"], device)
        self.down = []
        res_prefixes = []
        for i in range(len(boc)):
            blk = dict(resnets=[], attns=[], down=None)
            for j in range(cfg["layers_per_block"]):
                p = f"down_blocks.{i}.resnets.{j}."
                res_prefixes.append(p)
                blk["resnets"].append(ResnetBlock(sd, p, device, g, 1e-5))
                blk["attns"].append(TransformerBlock(sd, f"down_blocks.{i}.attentions.{j}.", device, cfg["num_heads"], g)
                                    if cfg["down_cross"][i] else None)
            if i != len(boc) - 1:
                k = f"down_blocks.{i}.downsamplers.0.conv"
                blk["down"] = PackedConv(sd[k + ".weight"], sd[k + ".bias"], device)
            self.down.append(blk)
        self.mid_res0 = ResnetBlock(sd, "mid_block.resnets.0.", device, g, 1e-5)
        self.mid_attn = TransformerBlock(sd, "mid_block.attentions.0.", device, cfg["num_heads"], g)
        self.mid_res1 = ResnetBlock(sd, "mid_block.resnets.1.", device, g, 1e-5)
        res_prefixes += ["mid_block.resnets.0.", "mid_block.resnets.1."] + list(extra_resnets)
        self.temb = TimeEmbedding(sd, device, boc[0], res_prefixes)

    def transformers(self):
        for blk in self.down:
            for a in blk["attns"]:
                if a is not None:
                    yield a
        yield self.mid_attn

    def run_down(self, sample, temb, after_block=None, cfg_shared=False):
        """cfg_shared: the two batch halves of `sample` are identical (classifier-free guidance: duplicated latents, one
        timestep): the first resnet and the first transformer up to its text cross-attention run on one half only."""
        res = [sample]
        shared = cfg_shared and self.down[0]["attns"][0] is not None and sample.shape[0] % 2 == 0
        for i, blk in enumerate(self.down):
            for j, (r, a) in enumerate(zip(blk["resnets"], blk["attns"])):
                if shared and i == 0 and j == 0:
                    sample = a(r(sample[: sample.shape[0] // 2], temb), cfg_shared=True)
                    res.append(sample)
                    continue
                sample = r(sample, temb)
                if a is not None:
                    sample = a(sample)
                res.append(sample)
            if blk["down"] is not None:
                sample = ops.conv(sample, blk["down"], stride=2, gn_part=True)
                res.append(sample)
            if after_block is not None:
                sample = after_block(i, sample)
        return sample, res

    def run_mid(self, sample, temb):
        sample = self.mid_res0(sample, temb)
        sample = self.mid_attn(sample)
        return self.mid_res1(sample, temb)
