"""CPU: host-side logic that needs no GPU — scheduler tables vs the oracle's DDIM, checkpoint specs, timestep
retrieval errors, GOP / shard bookkeeping."""
import pytest
import torch

from diffcodec_amd import sharding, weights as W
from diffcodec_amd.pipeline import retrieve_timesteps
from diffcodec_amd.scheduler import DDIMScheduler


def test_ddim_tables_match_oracle():
    from oracle.pipeline_ref import DDIMRef
    s, r = DDIMScheduler(), DDIMRef()
    for n in (4, 20, 50):
        s.set_timesteps(n)
        r.set_timesteps(n)
        assert torch.equal(s.timesteps, r.timesteps)
        coef = s.coefficients()
        x = torch.randn(2, 4, 8, 8)
        e = torch.randn(2, 4, 8, 8)
        for i, t in enumerate(s.timesteps.tolist()):
            c = coef[i]
            mine = c[2] * ((x - c[0] * e) / c[1]) + c[3] * e
            torch.testing.assert_close(mine, r.step(e, t, x), rtol=1e-6, atol=1e-6)
    s.set_timesteps(20)
    assert s.timesteps[0].item() == 951 and s.timesteps[-1].item() == 1 and s.order == 1 and s.init_noise_sigma == 1.0


def test_retrieve_timesteps_errors():
    s = DDIMScheduler()
    with pytest.raises(ValueError):
        retrieve_timesteps(s, 10, None, timesteps=[1, 2], sigmas=[0.1])
    with pytest.raises(ValueError):
        retrieve_timesteps(s, None, None, timesteps=[999, 500])      # DDIM here has no custom-timestep support
    with pytest.raises(ValueError):
        retrieve_timesteps(s, None, None, sigmas=[1.0, 0.5])
    ts, n = retrieve_timesteps(s, 20, None)
    assert n == 20 and len(ts) == 20


def test_checkpoint_specs_match_sd15_parameter_counts():
    assert W.param_count(W.unet_spec()) == 859_520_964
    assert W.param_count(W.vae_spec()) == 83_653_863
    cn = W.controlnet_spec()
    custom = {k: v for k, v in cn.items() if k.startswith("feature_extractor.") or k.startswith("fdn")}
    assert len(custom) == 76      # 60 extractor + 16 FDN tensors (SURVEY.md §8(b) says 74; its own parameter totals, checked below, need 76)
    assert W.param_count({k: v for k, v in cn.items() if k.startswith("feature_extractor.")}) == 16_275_204
    assert W.param_count({k: v for k, v in cn.items() if k.startswith("fdn")}) == 40_555_520
    assert cn["fdn16.conv_gamma.weight"][1] == (640, 640, 3, 3)
    assert cn["feature_extractor.zero_convs.3.weight"][1] == (1280, 640, 3, 3)


def test_filter_state_dict_is_strict_false_plus_shape_filter():
    spec = W.controlnet_spec(dict(W.SD15_UNET_CONFIG, block_out_channels=(32, 64, 128, 128), cross_attention_dim=64))
    sd = W.synthesize(spec, 0)
    sd.pop("feature_extractor.wrapper.0.metric_net.0.weight")           # older checkpoints lack metric_net (pipeline.ipynb)
    sd["fdn08.conv_beta.bias"] = torch.zeros(3)
    sd["stray.key"] = torch.zeros(1)
    kept, rep = W.filter_state_dict(sd, spec)
    assert rep["missing"] == ["feature_extractor.wrapper.0.metric_net.0.weight"]
    assert rep["mismatched"] == ["fdn08.conv_beta.bias"] and rep["unexpected"] == ["stray.key"]
    assert "fdn08.conv_beta.bias" not in kept


def test_gop_and_shards():
    units = sharding.gop_inter_frames(97, 12)            # 96-frame UVG clip + closing anchor
    assert len(units) == 88 and units[0] == (1, 0, 12) and units[-1] == (95, 84, 96)
    assert all(f % 12 for f, _, _ in units)
    for world in (1, 2, 3, 8):
        got = sorted(sum((sharding.shard_units(len(units), r, world) for r in range(world)), []))
        assert got == list(range(len(units)))             # every unit exactly once
    assert len(sharding.shard_units(88, 0, 8)) == 11      # 88 = 8 x 11
    with pytest.raises(ValueError):
        sharding.shard_units(4, 4, 4)


def test_pipeline_validation_errors_need_no_gpu():
    from types import SimpleNamespace
    from diffcodec_amd.pipeline import StableDiffusionDualFlowControlNetPipeline as P
    vae = SimpleNamespace(config=SimpleNamespace(block_out_channels=[1, 2, 3, 4], scaling_factor=0.18215))
    unet = SimpleNamespace(config=SimpleNamespace(in_channels=4, time_cond_proj_dim=None), device=torch.device("cpu"))
    pipe = P(vae, None, None, unet, None, DDIMScheduler(), None, None)
    assert pipe.vae_scale_factor == 8
    pe = torch.zeros(1, 77, 768)
    with pytest.raises(ValueError, match="Provide both"):
        pipe(prompt_embeds=pe)
    with pytest.raises(ValueError, match=r"controlnet_cond must be \[B,6,H,W\]"):
        pipe(prompt_embeds=pe, controlnet_cond=torch.zeros(1, 5, 64, 64), flow_cond=torch.zeros(1, 4, 64, 64))
    with pytest.raises(ValueError, match=r"flow_cond must be \[B,4,H,W\]"):
        pipe(prompt_embeds=pe, controlnet_cond=torch.zeros(1, 6, 64, 64), flow_cond=torch.zeros(1, 2, 64, 64))
    with pytest.raises(ValueError, match="mismatch"):
        pipe(prompt_embeds=torch.zeros(3, 77, 768), negative_prompt_embeds=torch.zeros(3, 77, 768),
             controlnet_cond=torch.zeros(2, 6, 64, 64), flow_cond=torch.zeros(2, 4, 64, 64))
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe(prompt_embeds=pe, negative_prompt_embeds=pe, controlnet_cond=torch.zeros(1, 6, 60, 60), flow_cond=torch.zeros(1, 4, 60, 60))


def test_device_tables_are_stable_for_equivalent_device_spellings():
    """torch.device('cuda') != torch.device('cuda:0') made every call rebuild the tables (and drop the captured
    hipGraphs).  The comparison resolves an index-less CUDA device to the current one; on CPU the same logic is exercised
    with the spellings 'cpu' / torch.device('cpu')."""
    s = DDIMScheduler()
    s.set_timesteps(20)
    a = s.device_tables("cpu")
    v = s.table_version
    b = s.device_tables(torch.device("cpu"))
    assert a[0] is b[0] and a[1] is b[1] and s.table_version == v
    s.set_timesteps(20)                                  # same schedule: tables kept
    assert s.device_tables("cpu")[0] is a[0] and s.table_version == v
    s.set_timesteps(10)                                  # new schedule: rebuilt, version bumped
    assert s.device_tables("cpu")[0] is not a[0] and s.table_version == v + 1
    same = DDIMScheduler._same_device
    assert same("cpu", torch.device("cpu")) and not same("cpu", "meta")
    assert same(torch.device("cuda", 1), "cuda:1") and not same("cuda:0", "cuda:1")      # explicit indices never touch the runtime


def test_vae_legacy_attention_keys_are_remapped():
    """SD-1.5's vae safetensors stores the mid-block attention as query/key/value/proj_attn (some exports as 1x1-conv
    weights); diffusers converts on load (validation.py:33).  Same conversion in `remap_vae_attention_keys`."""
    cfg = dict(W.SD15_VAE_CONFIG, block_out_channels=(32, 64, 64, 64))
    sd = W.synthesize(W.vae_spec(cfg), 0)
    legacy = {}
    names = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in sd.items():
        for new, old in names.items():
            if f".attentions.0.{new}." in k:
                k = k.replace(f".{new}.", f".{old}.")
                if v.dim() == 2 and old != "proj_attn":
                    v = v[:, :, None, None]              # conv-style export of the linear weight
                break
        legacy[k] = v
    assert sum("query" in k for k in legacy) == 4 and not any(".to_q." in k for k in legacy)
    back = W.remap_vae_attention_keys(legacy)
    assert list(back) == list(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    again = W.remap_vae_attention_keys(sd)               # new-style checkpoints pass through
    assert all(again[k] is sd[k] for k in sd)
    kept, rep = W.filter_state_dict(back, W.vae_spec(cfg))
    assert not rep["missing"] and not rep["mismatched"] and not rep["unexpected"]


def test_ddim_eta_matches_oracle_on_cpu_tensors():
    """eta > 0 (pipeline.py:289): the folded coefficients of `DDIMScheduler.step` equal the oracle's unfolded form."""
    from oracle.pipeline_ref import DDIMRef
    s, r = DDIMScheduler(), DDIMRef()
    s.set_timesteps(10)
    r.set_timesteps(10)
    x, e, nz = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    ratio = s.num_train_timesteps // s.num_inference_steps
    for t in s.timesteps.tolist():
        prev = t - ratio
        a_t = float(s.alphas_cumprod[t])
        a_p = float(s.alphas_cumprod[prev] if prev >= 0 else s.final_alpha_cumprod)
        std = 0.7 * ((1 - a_p) / (1 - a_t) * (1 - a_t / a_p)) ** 0.5
        mine = (a_p / a_t) ** 0.5 * x + (-(a_p * (1 - a_t) / a_t) ** 0.5 + (1 - a_p - std * std) ** 0.5) * e + std * nz
        torch.testing.assert_close(mine, r.step(e, t, x, 0.7, nz), rtol=1e-5, atol=1e-5)


def test_warp_cond_is_validated_like_the_other_controls():
    """warp_cond [B,3,H,W] (flow_resnet.py:58) gets the checks pipeline.py:187-192,249 apply to the other controls."""
    from types import SimpleNamespace
    from diffcodec_amd.pipeline import StableDiffusionDualFlowControlNetPipeline as P
    vae = SimpleNamespace(config=SimpleNamespace(block_out_channels=[1, 2, 3, 4], scaling_factor=0.18215))
    unet = SimpleNamespace(config=SimpleNamespace(in_channels=4, time_cond_proj_dim=None), device=torch.device("cpu"))
    pipe = P(vae, None, None, unet, None, DDIMScheduler(), None, None)
    pe = torch.zeros(2, 77, 768)
    ok = dict(prompt_embeds=pe, negative_prompt_embeds=pe, controlnet_cond=torch.zeros(2, 6, 64, 64), flow_cond=torch.zeros(2, 4, 64, 64))
    with pytest.raises(ValueError, match=r"warp_cond must be \[B,3,H,W\]"):
        pipe(**ok, warp_cond=torch.zeros(2, 4, 64, 64))
    with pytest.raises(ValueError, match="warp_cond is"):
        pipe(**ok, warp_cond=torch.zeros(2, 3, 32, 64))
    with pytest.raises(ValueError, match="warp_cond batch=3"):
        pipe(**ok, warp_cond=torch.zeros(3, 3, 64, 64))


def test_randn_tensor_follows_diffusers_generator_rules():
    """prepare_latents / DDIM variance noise (pipeline.py:269-278, :289): CPU generator -> host draw; a list of generators draws
    one batch row each; a list of the wrong length is the library's ValueError."""
    from diffcodec_amd.scheduler import randn_tensor
    a = randn_tensor((2, 4, 8, 8), torch.Generator().manual_seed(3), "cpu")
    assert torch.equal(a, torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(3)))
    gens = [torch.Generator().manual_seed(k) for k in (5, 6)]
    b = randn_tensor((2, 4, 8, 8), gens, "cpu")
    want = torch.cat([torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(k)) for k in (5, 6)], 0)
    assert torch.equal(b, want)
    assert torch.equal(randn_tensor((2, 4, 8, 8), [torch.Generator().manual_seed(3)], "cpu"), a)      # a one-element list is that generator
    with pytest.raises(ValueError, match="list of generators of length 3"):
        randn_tensor((2, 4, 8, 8), gens + gens[:1], "cpu")
    assert randn_tensor((1, 4, 8, 8), None, "cpu", torch.bfloat16).dtype == torch.bfloat16


def test_unipc_device_table_reproduces_the_tensor_form_restatement():
    """The coefficient rows the fused loop hands to dc_cfg_unipc_step (scheduler.UniPCMultistepScheduler.coefficients: the order
    warm-up state machine unrolled, flags + corrector / predictor coefficients per step), applied with the kernel's update rule
    in float64 on the host, against the oracle's tensor-form UniPC (written independently, in the library's D1s / rhos form):
    every latent of 2-, 5-, 20- and 40-step schedules.  No GPU: this pins the TABLE; the kernel's arithmetic is pinned to the
    generic scheduler step bit for bit by tests/test_gpu_round3.py."""
    from diffcodec_amd.scheduler import UniPCMultistepScheduler
    from oracle.pipeline_ref import UniPCRef
    for n in (2, 5, 20, 40):
        s, r = UniPCMultistepScheduler(), UniPCRef()
        s.set_timesteps(n)
        r.set_timesteps(n)
        rows = s.coefficients().double()
        assert rows.shape == (n, UniPCMultistepScheduler.UNIPC_ROW)
        flags = rows[:, 2].long().tolist()
        assert flags[0] == (0 if n == 1 else 0) and all(f & 1 for f in flags[1:])          # no corrector on the first step only
        assert flags[-1] & 4 == 0                                                            # lower_order_final: last predictor is order 1
        g = torch.Generator().manual_seed(n)
        x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
        xr = x.clone()
        m0 = m1 = last = torch.zeros_like(x)
        for i, t in enumerate(s.timesteps.tolist()):
            eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
            c = rows[i]
            f = int(c[2])
            mt = c[0] * x + c[1] * eps
            xc = x
            if f & 1:
                xc = c[3] * last + c[4] * m0 + c[5] * mt + (c[6] * m1 if f & 2 else 0.0)
            last, m1, m0 = xc, m0, mt
            x = c[7] * xc + c[8] * mt + (c[9] * m1 if f & 4 else 0.0)
            xr = r.step(eps, t, xr)
            torch.testing.assert_close(x, xr.double(), rtol=2e-5, atol=2e-5)     # the rows are fp32
