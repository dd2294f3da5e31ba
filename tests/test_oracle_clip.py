"""Pins oracle/clip_ref.py against the third-party library the reference actually calls (transformers CLIPTextModel,
validation.py:31-32) on a seeded random-init model: same state dict, same ids, fp32 CPU."""
import pytest
import torch

from diffcodec_amd import weights
from oracle import clip_ref

SMALL = dict(hidden_size=128, num_hidden_layers=3, num_attention_heads=4, intermediate_size=256, vocab_size=1000,
             max_position_embeddings=77, layer_norm_eps=1e-5, eos_token_id=2)


def _hf_model(cfg, sd):
    tr = pytest.importorskip("transformers")
    hf_cfg = tr.CLIPTextConfig(hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                               num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                               vocab_size=cfg["vocab_size"], max_position_embeddings=cfg["max_position_embeddings"],
                               layer_norm_eps=cfg["layer_norm_eps"], hidden_act="quick_gelu", eos_token_id=cfg["eos_token_id"],
                               bos_token_id=0, pad_token_id=1)
    m = tr.CLIPTextModel(hf_cfg).eval()
    if not any(k.startswith("text_model.") for k in m.state_dict()):          # transformers >= 5 flattened the module tree;
        sd = {k.removeprefix("text_model."): v for k, v in sd.items()}         # checkpoints on disk keep the 4.x prefix
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def _ids(cfg, b=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfg["vocab_size"] - 1, (b, cfg["max_position_embeddings"]), generator=g)
    ids[:, 0] = 0
    for i in range(b):                      # an EOS (highest id) somewhere, padding after it — the tokenizer's layout
        e = 5 + 7 * i
        ids[i, e] = cfg["vocab_size"] - 1
        ids[i, e + 1:] = cfg["vocab_size"] - 1
    return ids


def test_spec_matches_transformers_state_dict():
    tr = pytest.importorskip("transformers")
    spec = weights.clip_text_spec(SMALL)
    m = _hf_model(SMALL, weights.synthesize(spec, seed=1))
    hf = {k.removeprefix("text_model."): tuple(v.shape) for k, v in m.state_dict().items() if "position_ids" not in k}
    assert hf == {k.removeprefix("text_model."): tuple(s) for k, (_, s) in spec.items()}
    assert weights.param_count(weights.clip_text_spec()) == 123_060_480          # ViT-L/14 text tower


def test_restatement_equals_transformers():
    sd = weights.synthesize(weights.clip_text_spec(SMALL), seed=1, gain=2.0)
    ids = _ids(SMALL)
    m = _hf_model(SMALL, sd)
    with torch.no_grad():
        ref = m(input_ids=ids, output_hidden_states=True)
    last, pooled, hidden = clip_ref.clip_text_forward(sd, SMALL, ids, output_hidden_states=True)
    assert torch.allclose(last, ref.last_hidden_state, atol=1e-5, rtol=1e-5)
    assert torch.allclose(pooled, ref.pooler_output, atol=1e-5, rtol=1e-5)
    assert len(hidden) == len(ref.hidden_states)
    for a, b in zip(hidden, ref.hidden_states):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
