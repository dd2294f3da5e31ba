"""CPU: the oracle's restatement of the control pyramid against goldens captured from the imported reference
(oracle/make_goldens.py).  fp32; tolerances cover only conv summation-order differences."""
import os

import numpy as np
import torch

from oracle import control_ref as C
from oracle import splat as S
from diffcodec_amd.synthetic import synth_controls


def _sd(z, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_soft_splat_wrapper_and_flow_normalise(golden_dir):
    z = np.load(os.path.join(golden_dir, "control_splat_small.npz"))
    x, fl, me = (torch.from_numpy(z[k]) for k in ("x", "flow", "metric"))
    out = S.softsplat(x, fl, me, "soft")
    # same C splat loop; exp() is libm here vs torch in the capture -> last-ulp differences only
    torch.testing.assert_close(out, torch.from_numpy(z["soft"]), rtol=2e-6, atol=1e-6)
    out_t = S.softsplat_torch(x, fl, me, "soft")
    torch.testing.assert_close(out_t, torch.from_numpy(z["soft"]), rtol=1e-5, atol=1e-5)
    fb = torch.from_numpy(z["flow_big"])
    assert torch.equal(C.resize_and_normalize_flow(fb, 16, 16), torch.from_numpy(z["rn16"]))
    assert torch.equal(C.resize_and_normalize_flow(fb, 8, 8), torch.from_numpy(z["rn8"]))
    m = C.compute_mask(torch.from_numpy(z["mask_a"]), torch.from_numpy(z["mask_b"]))
    assert torch.equal(m, torch.from_numpy(z["mask"]))
    assert 0.05 < m.mean() < 0.95                                   # both sides of the 0.3 threshold exercised


def test_fdn(golden_dir):
    z = np.load(os.path.join(golden_dir, "control_fdn.npz"))
    sd = _sd(z)
    y = C.fdn(sd, "", torch.from_numpy(z["x"]), torch.from_numpy(z["local"]))
    torch.testing.assert_close(y, torch.from_numpy(z["y"]), rtol=1e-5, atol=1e-5)


def test_feature_warper(golden_dir):
    z = np.load(os.path.join(golden_dir, "control_warper.npz"))
    sd = _sd(z)
    w, m = C.feature_warper(sd, "", torch.from_numpy(z["feat"]), torch.from_numpy(z["flow"]), torch.from_numpy(z["mask"]))
    torch.testing.assert_close(m, torch.from_numpy(z["metric"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(w, torch.from_numpy(z["warped"]), rtol=1e-4, atol=1e-5)


def test_bi_dir_feature_extractor_512(golden_dir):
    z = np.load(os.path.join(golden_dir, "control_extractor512.npz"))
    sd = _sd(z)
    cond, flow = synth_controls(1, 512, seed=1234)
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6 * abs(float(z["cond_sum"]))
    assert abs(flow.double().sum().item() - float(z["flow_sum"])) < 1e-6 * max(1.0, abs(float(z["flow_sum"])))
    with torch.no_grad():
        outs, aux = C.bi_dir_feature_extractor(sd, "", cond, flow, return_aux=True)
    for i, o in enumerate(outs):
        ref = torch.from_numpy(z[f"p{i}"])
        assert o.shape == ref.shape
        torch.testing.assert_close(o, ref, rtol=2e-4, atol=2e-5)
    # the fixture exercises both mask values and the double-hole branch at some scale
    assert any(0.02 < a["occ_f"].mean() < 0.98 for a in aux)
    assert any(((a["occ_f"] + a["occ_b"]) > 1.5).any() for a in aux)


def test_bi_dir_residue_extractor_512(golden_dir):
    z = np.load(os.path.join(golden_dir, "control_residue512.npz"))
    sd = _sd(z)
    cond, flow = synth_controls(1, 512, seed=1234)
    with torch.no_grad():
        outs = C.bi_dir_residue_extractor(sd, "", cond[:, :3], cond[:, 3:], flow[:, :2], flow[:, 2:])
    for i, o in enumerate(outs):
        torch.testing.assert_close(o, torch.from_numpy(z[f"p{i}"]), rtol=2e-4, atol=2e-5)


def test_warp_extractor_512(golden_dir):
    """oracle.control_ref.warp_extractor vs the reference module's outputs (extractors.py:26-65); weights regenerate from the
    package's seeded synthesis, pinned by a checksum in the fixture."""
    from diffcodec_amd import weights as W
    from oracle.make_goldens import synth_warp_image
    z = np.load(os.path.join(golden_dir, "control_warp512.npz"))
    spec = {k[len("warp_extractor."):]: v for k, v in W.rescontrolnet_spec().items() if k.startswith("warp_extractor.")}
    sd = W.synthesize(spec, seed=31, bf16_round=False)
    assert abs(sum(v.double().abs().sum().item() for v in sd.values()) - float(z["w_sum"])) < 1e-9 * float(z["w_sum"])
    x = synth_warp_image()
    assert abs(x.double().sum().item() - float(z["x_sum"])) < 1e-9 * float(z["x_sum"])
    with torch.no_grad():
        outs = C.warp_extractor(sd, "", x)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(z[f"p{i}_shape"])
        torch.testing.assert_close(o[:, :32], torch.from_numpy(z[f"p{i}"]), rtol=2e-4, atol=2e-5)
        assert abs(o.double().sum().item() - float(z[f"p{i}_sum"])) < 1e-5 * float(z[f"p{i}_abs"])


def test_pyramid_hoist_equals_per_step_recompute():
    """flownet.py:78 recomputes the pyramid in every forward; the oracle's default (and the product) computes it once per
    frame.  Same latents either way — the pyramid takes neither the sample nor the timestep.  Reduced widths, 2 steps."""
    from diffcodec_amd import selftest as T
    from diffcodec_amd.synthetic import synth_latents, synth_text
    from oracle import pipeline_ref as R
    usd, csd, _ = T.small_state_dicts()
    cond, flow = synth_controls(1, 512, seed=99)                    # the reference asserts the 64/32/16/8 pyramid: 512 px
    cond, flow = cond[:, :, :128, :128].contiguous(), flow[:, :, :128, :128].contiguous()
    pe, npe = synth_text(1, dim=T.SMALL_UNET["cross_attention_dim"])
    lat = synth_latents(1, 128)
    kw = dict(num_inference_steps=2, guidance_scale=4.5, controlnet_conditioning_scale=1.7, output_type="latent")
    a = R.decode_frame(usd, csd, None, T.SMALL_UNET, T.SMALL_VAE, cond, flow, pe, npe, lat, hoist=True, **kw)
    b = R.decode_frame(usd, csd, None, T.SMALL_UNET, T.SMALL_VAE, cond, flow, pe, npe, lat, hoist=False, **kw)
    # equal up to fp32 summation order: the hoisted pyramid is computed at batch 1, the per-step one at the CFG batch 2, and
    # the host conv picks a different blocking per batch size (measured 2e-5 max abs on latents of O(1))
    torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4)
