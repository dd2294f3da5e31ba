"""`CLIPTextModel` operator behind `encode_prompt` (reference pipeline.py:223-236; built at validation.py:31-32),
MI355X-native: the ViT-L/14 text tower as C-ABI launches (`ops`).  Runs once per prompt, outside the sampling loop.

Same call surface as the transformers module the reference passes to its pipeline:
    out = text_encoder(input_ids, attention_mask=None, output_hidden_states=False)
    out[0] / out.last_hidden_state [B,77,768]; out.pooler_output; out[-1] / out.hidden_states when requested
plus `.device`, `.dtype`, `.config`, `.text_model.final_layer_norm(x)` (the clip_skip path).
Layers (transformers `CLIPEncoderLayer`): x += out_proj(causal_attn(LN1(x))); x += fc2(quick_gelu(fc1(LN2(x))))."""
from types import SimpleNamespace

import torch

from . import ops, weights
from .ops import PackedConv


class TextEncoderOutput(tuple):
    """tuple-compatible (`out[0]`, `out[-1]`) with the attribute names of `BaseModelOutputWithPooling`."""

    def __new__(cls, last_hidden_state, pooler_output, hidden_states=None):
        items = (last_hidden_state, pooler_output) + ((hidden_states,) if hidden_states is not None else ())
        self = super().__new__(cls, items)
        self.last_hidden_state, self.pooler_output, self.hidden_states = last_hidden_state, pooler_output, hidden_states
        return self


class _FinalNorm:
    def __init__(self, g, b, eps):
        self.g, self.b, self.eps = g, b, eps

    def __call__(self, x):
        return ops.layer_norm(x.to(torch.bfloat16).contiguous(), self.g, self.b, self.eps)


class HipCLIPTextModel:
    def __init__(self, state_dict, config=None, device="cuda"):
        cfg = dict(weights.SD15_CLIP_TEXT_CONFIG if config is None else config)
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.config = SimpleNamespace(**cfg, use_attention_mask=False)
        sd, dev = state_dict, self.device
        e = "text_model.embeddings."
        self.tok = sd[e + "token_embedding.weight"].to(dev, torch.bfloat16).contiguous()
        self.pos = sd[e + "position_embedding.weight"].to(dev, torch.bfloat16).contiguous()
        f32 = lambda k: sd[k].float().to(dev).contiguous()
        self.layers = []
        for i in range(cfg["num_hidden_layers"]):
            p = f"text_model.encoder.layers.{i}."
            a = p + "self_attn."
            qkv_w = torch.cat([sd[a + n + ".weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)       # one [3C,C] GEMM
            qkv_b = torch.cat([sd[a + n + ".bias"] for n in ("q_proj", "k_proj", "v_proj")], 0)
            self.layers.append(dict(
                ln1=(f32(p + "layer_norm1.weight"), f32(p + "layer_norm1.bias")),
                ln2=(f32(p + "layer_norm2.weight"), f32(p + "layer_norm2.bias")),
                qkv=PackedConv(qkv_w, qkv_b, dev), out=PackedConv(sd[a + "out_proj.weight"], sd[a + "out_proj.bias"], dev),
                fc1=PackedConv(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], dev),
                fc2=PackedConv(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], dev)))
        fn = _FinalNorm(f32("text_model.final_layer_norm.weight"), f32("text_model.final_layer_norm.bias"), cfg["layer_norm_eps"])
        self.text_model = SimpleNamespace(final_layer_norm=fn)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def forward(self, input_ids, attention_mask=None, output_hidden_states=False, return_dict=True, **kw):
        if attention_mask is not None:
            raise NotImplementedError("SD-1.5's text encoder config has no use_attention_mask: the reference passes None")
        cfg = self.cfg
        ids = torch.as_tensor(input_ids).to(self.device, torch.int64).contiguous()
        if ids.dim() != 2 or ids.shape[1] > cfg["max_position_embeddings"]:
            raise ValueError(f"input_ids must be [B, T<={cfg['max_position_embeddings']}]")
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= cfg["vocab_size"]):
            raise IndexError("token id outside the vocabulary")
        heads, eps, c = cfg["num_attention_heads"], cfg["layer_norm_eps"], cfg["hidden_size"]
        x = ops.embed_tokens(ids, self.tok, self.pos)
        hidden = [x] if output_hidden_states else None
        for L in self.layers:
            h = ops.layer_norm(x, *L["ln1"], eps)
            qkv = ops.linear(h, L["qkv"])
            a = ops.attention_causal(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], heads)
            x = ops.linear(a, L["out"], residual=x)
            h = ops.layer_norm(x, *L["ln2"], eps)
            h = ops.linear(h, L["fc1"], act=2)                       # quick-GELU fused into the GEMM epilogue
            x = ops.linear(h, L["fc2"], residual=x)
            if output_hidden_states:
                hidden.append(x)
        last = self.text_model.final_layer_norm(x)
        # pooled = the hidden state at the EOS token (CLIPTextTransformer.forward): legacy configs (eos_token_id == 2)
        # take argmax(ids), newer ones the first position holding eos_token_id
        if cfg.get("eos_token_id", 2) == 2:
            pos = ids.argmax(-1)
        else:
            pos = (ids == cfg["eos_token_id"]).int().argmax(-1)
        pooled = last[torch.arange(ids.shape[0], device=self.device), pos]
        return TextEncoderOutput(last, pooled, tuple(hidden) if output_hidden_states else None)

    __call__ = forward


def load_text_encoder(base, device="cuda"):
    """<base>/text_encoder/{config.json, model.safetensors} (validation.py:31-32)."""
    import json
    import os
    with open(os.path.join(base, "text_encoder", "config.json")) as f:
        j = json.load(f)
    cfg = {k: j.get(k, v) for k, v in weights.SD15_CLIP_TEXT_CONFIG.items()}
    if j.get("hidden_act", "quick_gelu") != "quick_gelu":
        raise ValueError("only the quick_gelu CLIP text tower of SD-1.5 is implemented")
    sd = weights.load_safetensors(os.path.join(base, "text_encoder", "model.safetensors"))
    kept, report = weights.filter_state_dict(sd, weights.clip_text_spec(cfg))
    if report["missing"] or report["mismatched"]:
        raise KeyError(f"text encoder checkpoint does not match the config: {report['missing'][:3]} {report['mismatched'][:3]}")
    return HipCLIPTextModel(kept, cfg, device)
