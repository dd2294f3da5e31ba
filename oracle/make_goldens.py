"""ORACLE (test infrastructure only) — golden-vector capture.  Runs ONLY in the build container.

Imports the importable part of the reference (/root/reference/controlnet/{control_utils,extractors}.py)
and records seeded inputs / outputs as small .npz fixtures under tests/golden/.  Fixtures are data only
(inputs, weights, expected outputs); no reference source travels.

Two substitutions are needed to import on a CPU-only box, both recorded in DESIGN.md:
  * `cupy` is absent -> a stub module providing the three names softsplat.py touches at import time
    (int32, float32, memoize).  No cupy function is ever executed.
  * the reference's forward-splat kernel is CUDA-only (softsplat.py:347-348 asserts on CPU) ->
    `softsplat_func.apply` is replaced by oracle.splat.splat_sum (the plain-C restatement of
    softsplat.py:285-335).  Everything else (wrapper math, masks, warper, fusion, FDN, flow
    normalisation) is the reference's own Python executing.

Usage:  python -m oracle.make_goldens          (from the repo root)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    stub = types.ModuleType("cupy")
    stub.int32 = np.int32
    stub.float32 = np.float32
    stub.memoize = lambda **kw: (lambda f: f)
    sys.modules.setdefault("cupy", stub)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import controlnet.softsplat as ss            # noqa: E402
    from oracle.splat import splat_sum

    class _CpuSplat:
        @staticmethod
        def apply(tenIn, tenFlow):
            return splat_sum(tenIn.float(), tenFlow.float())

    ss.softsplat_func = _CpuSplat
    import controlnet.control_utils as cu        # noqa: E402
    import controlnet.extractors as ex           # noqa: E402
    return ss, cu, ex


def synth_controls(b, size, seed):
    """Smooth images in [0,1] and smooth bidirectional flow in pixel units."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(b, 6, size, size, generator=g)
    img = torch.nn.functional.avg_pool2d(img, 9, 1, 4, count_include_pad=False)
    lo = torch.randn(b, 2, size // 16, size // 16, generator=g) * 8.0
    fwd = torch.nn.functional.interpolate(lo, size=(size, size), mode="bilinear", align_corners=False)
    bwd = -fwd + 0.5 * torch.nn.functional.interpolate(
        torch.randn(b, 2, size // 16, size // 16, generator=g), size=(size, size), mode="bilinear", align_corners=False)
    return img, torch.cat([fwd, bwd], 1)


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def main():
    ss, cu, ex = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # --- 1. softsplat wrapper ('soft') + compute_mask + resize_and_normalize -------------------------
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 12, 16, generator=g)
    fl = torch.randn(2, 2, 12, 16, generator=g) * 3.0
    fl[0, 0, 0, 0] = float("inf")          # non-finite target is skipped (softsplat.py:301-302)
    fl[1, 1, 3, 4] = float("nan")
    me = torch.randn(2, 1, 12, 16, generator=g)
    soft = ss.softsplat(tenIn=x, tenFlow=fl, tenMetric=me, strMode="soft")
    flow_big = torch.randn(2, 2, 64, 64, generator=g) * 6.0
    rn16 = cu.resize_and_normalize_flow_batched(flow_big, 16, 16)
    rn8 = cu.resize_and_normalize_flow_batched(flow_big, 8, 8)
    fa = torch.randn(2, 2, 16, 16, generator=g) * 0.6
    fb = -fa + 0.25 * torch.randn(2, 2, 16, 16, generator=g)
    mask = cu.compute_mask(fa, fb)
    np.savez_compressed(os.path.join(OUT, "control_splat_small.npz"),
                        x=x.numpy(), flow=fl.numpy(), metric=me.numpy(), soft=soft.numpy(),
                        flow_big=flow_big.numpy(), rn16=rn16.numpy(), rn8=rn8.numpy(),
                        mask_a=fa.numpy(), mask_b=fb.numpy(), mask=mask.numpy())

    # --- 2. FDN -------------------------------------------------------------------------------------
    fdn = cu.FDN(norm_nc=64, label_nc=64)
    xs = torch.randn(2, 64, 8, 8, generator=g)
    lf = torch.randn(2, 64, 8, 8, generator=g)
    with torch.no_grad():
        y = fdn(xs, lf)
    d = {"w." + k: v for k, v in _np(fdn.state_dict()).items()}
    np.savez_compressed(os.path.join(OUT, "control_fdn.npz"), x=xs.numpy(), local=lf.numpy(), y=y.numpy(), **d)

    # --- 3. FeatureWarperSoftsplat -----------------------------------------------------------------
    fw = cu.FeatureWarperSoftsplat(with_learnable_metric=True, in_channels=16)
    feat = torch.randn(2, 16, 16, 16, generator=g)
    with torch.no_grad():
        warped, metric = fw(feat, fa, mask=mask)
    d = {"w." + k: v for k, v in _np(fw.state_dict()).items()}
    np.savez_compressed(os.path.join(OUT, "control_warper.npz"), feat=feat.numpy(), flow=fa.numpy(),
                        mask=mask.numpy(), warped=warped.numpy(), metric=metric.numpy(), **d)

    # --- 4. Bi_Dir_FeatureExtractor at the reference's only legal size (512x512), narrow channels ----
    # inject_channels is a constructor argument (extractors.py:211); (32,32,64,128) keeps the fixture small.
    inj = (32, 32, 64, 128)
    fe = ex.Bi_Dir_FeatureExtractor(inject_channels=inj)
    gen = torch.Generator().manual_seed(5)
    for zc in fe.zero_convs:                      # zero-init would make every output 0 (extractors.py:257-262)
        zc.weight.data = torch.randn(zc.weight.shape, generator=gen) * 0.02
        zc.bias.data = torch.randn(zc.bias.shape, generator=gen) * 0.02
    cond, flow = synth_controls(1, 512, seed=1234)
    with torch.no_grad():
        outs = fe(cond, flow)
    d = {"w." + k: v.astype(np.float32) for k, v in _np(fe.state_dict()).items()}
    # inputs regenerate from the seed (synth_controls is duplicated in tests/); store a checksum + outputs
    np.savez_compressed(os.path.join(OUT, "control_extractor512.npz"),
                        cond_sum=np.float64(cond.double().sum().item()), flow_sum=np.float64(flow.double().sum().item()),
                        cond_lo=torch.nn.functional.avg_pool2d(cond, 16).numpy(),
                        p0=outs[0].numpy(), p1=outs[1].numpy(), p2=outs[2].numpy(), p3=outs[3].numpy(),
                        inject=np.array(inj), **d)

    # --- 5. Bi_Dir_ResidueExtractor + WarpExtractor (config-4 operator), narrow channels -------------
    import contextlib
    import io
    re_ = ex.Bi_Dir_ResidueExtractor(inject_channels=list(inj))
    for zc in re_.zero_convs:
        zc.weight.data = torch.randn(zc.weight.shape, generator=gen) * 0.02
        zc.bias.data = torch.randn(zc.bias.shape, generator=gen) * 0.02
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):      # extractors.py:174 prints
        routs = re_(cond[:, :3], cond[:, 3:], flow[:, :2], flow[:, 2:])
    d = {"w." + k: v.astype(np.float32) for k, v in _np(re_.state_dict()).items()}
    np.savez_compressed(os.path.join(OUT, "control_residue512.npz"),
                        p0=routs[0].numpy(), p1=routs[1].numpy(), p2=routs[2].numpy(), p3=routs[3].numpy(), **d)

    print("wrote goldens to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(" ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
