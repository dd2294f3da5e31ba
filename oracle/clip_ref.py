"""ORACLE (test infrastructure only; never imported by the product path) — fp32 CPU restatement of the text tower
behind `encode_prompt` (reference pipeline.py:223-236): transformers `CLIPTextModel.forward` for the SD-1.5 config
(quick_gelu MLP, causal mask, pre-LN layers, final LayerNorm; pooled = hidden state at the EOS position).

The arithmetic lives in a third-party dependency of the reference (`transformers`, unpinned in
controlnet/requirements.txt); that library IS importable in this image (5.15.0), so this restatement is pinned
against the library itself: tests/test_oracle_clip.py loads the same seeded state dict into
`transformers.CLIPTextModel` and requires agreement to 1e-5."""
import torch
import torch.nn.functional as F


@torch.no_grad()
def clip_text_forward(sd, cfg, input_ids, output_hidden_states=False):
    c, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    b, t = input_ids.shape
    e = "text_model.embeddings."
    x = sd[e + "token_embedding.weight"].float()[input_ids] + sd[e + "position_embedding.weight"].float()[:t]
    causal = torch.full((t, t), float("-inf")).triu(1)
    hidden = [x]
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}."
        lin = lambda v, n: F.linear(v, sd[p + n + ".weight"].float(), sd[p + n + ".bias"].float())
        h = F.layer_norm(x, (c,), sd[p + "layer_norm1.weight"].float(), sd[p + "layer_norm1.bias"].float(), eps)
        split = lambda v: v.reshape(b, t, heads, c // heads).transpose(1, 2)
        q, k, v = split(lin(h, "self_attn.q_proj")), split(lin(h, "self_attn.k_proj")), split(lin(h, "self_attn.v_proj"))
        w = (q @ k.transpose(-1, -2)) * (c // heads) ** -0.5 + causal
        a = (w.softmax(-1) @ v).transpose(1, 2).reshape(b, t, c)
        x = x + lin(a, "self_attn.out_proj")
        h = F.layer_norm(x, (c,), sd[p + "layer_norm2.weight"].float(), sd[p + "layer_norm2.bias"].float(), eps)
        h = lin(h, "mlp.fc1")
        h = h * torch.sigmoid(1.702 * h)                              # quick_gelu
        x = x + lin(h, "mlp.fc2")
        hidden.append(x)
    last = F.layer_norm(x, (c,), sd["text_model.final_layer_norm.weight"].float(), sd["text_model.final_layer_norm.bias"].float(), eps)
    if cfg.get("eos_token_id", 2) == 2:
        pos = input_ids.argmax(-1)
    else:
        pos = (input_ids == cfg["eos_token_id"]).int().argmax(-1)
    pooled = last[torch.arange(b), pos]
    return (last, pooled, tuple(hidden)) if output_hidden_states else (last, pooled)
