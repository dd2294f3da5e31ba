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
