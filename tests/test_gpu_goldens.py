"""GPU: every fixture under tests/golden/ (captured from the imported reference modules by oracle/make_goldens.py) fed to
the HIP kernels directly, through the C-ABI.  fp32 stage: tolerances cover conv / splat summation order only.
    control_splat_small.npz  softsplat 'soft' wrapper (softsplat.py:232-274), resize_and_normalize_flow_batched
                             (control_utils.py:74-97), compute_mask (control_utils.py:11-17)
    control_fdn.npz          FDN.forward (control_utils.py:19-34)
    control_warper.npz       FeatureWarperSoftsplat.forward (control_utils.py:49-72)
    control_warp512.npz      WarpExtractor.forward (extractors.py:26-65)
(control_extractor512 / control_residue512 are consumed by tests/test_gpu_models.py.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffcodec_amd import ops as o
    return o


def _t(z, k):
    return torch.from_numpy(z[k])


def test_splat_wrapper_flow_normalise_and_mask_goldens(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "control_splat_small.npz"))
    x, fl, me = (_t(z, k).to(DEV) for k in ("x", "flow", "metric"))        # flow holds an inf and a nan target (skipped)
    out = ops.splat_soft(x, fl, me).cpu()
    torch.testing.assert_close(out, _t(z, "soft"), rtol=1e-4, atol=1e-5)
    fb = _t(z, "flow_big").to(DEV)
    torch.testing.assert_close(ops.flow_resize_normalize(fb, 16, 16).cpu(), _t(z, "rn16"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ops.flow_resize_normalize(fb, 8, 8).cpu(), _t(z, "rn8"), rtol=1e-5, atol=1e-6)
    m = ops.occlusion_mask(_t(z, "mask_a").to(DEV), _t(z, "mask_b").to(DEV)).cpu()
    ref = _t(z, "mask")
    assert m.shape == ref.shape and set(m.unique().tolist()) <= {0.0, 1.0}
    assert (m != ref).float().mean().item() < 5e-3                       # threshold compare at 0.3: summation-order flips only
    assert 0.05 < ref.mean() < 0.95


def test_fdn_golden(ops, golden_dir):
    """GroupNorm(32, affine=False)(x) * (1 + conv_gamma(P)) + conv_beta(P): the device path runs the gamma/beta convs on the
    bf16 MFMA tile kernel and the modulation in bf16, so the bar is the bf16 one (2e-2 on O(1) values)."""
    from diffcodec_amd.ops import PackedConv
    z = np.load(os.path.join(golden_dir, "control_fdn.npz"))
    x, local, y = _t(z, "x"), _t(z, "local"), _t(z, "y")
    cg = PackedConv(_t(z, "w.conv_gamma.weight"), _t(z, "w.conv_gamma.bias"), DEV)
    cb = PackedConv(_t(z, "w.conv_beta.weight"), _t(z, "w.conv_beta.bias"), DEV)
    p = ops.nchw_f32_to_nhwc_bf16(local.to(DEV))
    gamma, beta = ops.conv(p, cg), ops.conv(p, cb)
    xs = ops.nchw_f32_to_nhwc_bf16(x.to(DEV))
    ab = ops.group_norm_ab(xs, None, None, 32, 1e-5)
    out = ops.fdn_modulate(xs, ab, gamma, beta).float().cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(out, y, rtol=3e-2, atol=3e-2)
    assert (out - y).abs().mean().item() < 6e-3


def test_feature_warper_golden(ops, golden_dir):
    from diffcodec_amd.ops import PackedConvF32
    z = np.load(os.path.join(golden_dir, "control_warper.npz"))
    feat, flow, mask = (_t(z, k).to(DEV) for k in ("feat", "flow", "mask"))
    m0 = PackedConvF32(_t(z, "w.metric_net.0.weight"), _t(z, "w.metric_net.0.bias"), DEV)
    m2 = PackedConvF32(_t(z, "w.metric_net.2.weight"), _t(z, "w.metric_net.2.bias"), DEV)
    metric = ops.conv3x3_nchw_f32(ops.conv3x3_nchw_f32(feat, m0, 1, True), m2, 1, False)
    torch.testing.assert_close(metric.cpu(), _t(z, "metric"), rtol=1e-4, atol=1e-5)
    warped = ops.splat_soft(feat, flow, metric, mask=mask)
    torch.testing.assert_close(warped.cpu(), _t(z, "warped"), rtol=1e-3, atol=1e-4)


def test_warp_extractor_golden_512(golden_dir):
    """WarpExtractor at its native widths: the weights regenerate from the package's seeded synthesis (checksum pinned),
    the outputs come from the reference module (oracle/make_goldens.py section 6)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffcodec_amd import weights as W
    from diffcodec_amd.rescontrolnet import WarpExtractor
    from oracle.make_goldens import synth_warp_image
    z = np.load(os.path.join(golden_dir, "control_warp512.npz"))
    spec = {k: v for k, v in W.rescontrolnet_spec().items() if k.startswith("warp_extractor.")}
    sd = W.synthesize({k[len("warp_extractor."):]: v for k, v in spec.items()}, seed=31, bf16_round=False)
    assert abs(sum(v.double().abs().sum().item() for v in sd.values()) - float(z["w_sum"])) < 1e-9 * float(z["w_sum"])
    x = synth_warp_image()
    assert abs(x.double().sum().item() - float(z["x_sum"])) < 1e-9 * float(z["x_sum"])
    outs = WarpExtractor({"warp_extractor." + k: v for k, v in sd.items()}, "warp_extractor.", DEV)(x.to(DEV))
    for i, o in enumerate(outs):
        o = o.cpu()
        assert tuple(o.shape) == tuple(z[f"p{i}_shape"])
        torch.testing.assert_close(o[:, :32], _t(z, f"p{i}"), rtol=1e-3, atol=1e-4)
        assert abs(o.double().sum().item() - float(z[f"p{i}_sum"])) < 1e-4 * float(z[f"p{i}_abs"])
        assert abs(o.double().abs().sum().item() - float(z[f"p{i}_abs"])) < 1e-4 * float(z[f"p{i}_abs"])
