"""ORACLE (test infrastructure only) — Python face of ``splat_oracle.c``.

``softsplat(tenIn, tenFlow, tenMetric, strMode)`` mirrors the reference wrapper
(controlnet/softsplat.py:232-274) for the modes the hot path uses ('soft', plus 'sum' for tests); the
forward kernel itself (softsplat.py:285-335) is the plain-C loop in ``splat_oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsplat_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        fp = ctypes.POINTER(ctypes.c_float)
        _LIB.splat_sum_f32.argtypes = [fp, fp, fp] + [ctypes.c_int] * 4
        _LIB.splat_soft_f32.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 4
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def splat_sum(ten_in: torch.Tensor, ten_flow: torch.Tensor) -> torch.Tensor:
    a = np.ascontiguousarray(ten_in.detach().cpu().float().numpy())
    f = np.ascontiguousarray(ten_flow.detach().cpu().float().numpy())
    n, c, h, w = a.shape
    assert f.shape == (n, 2, h, w)
    out = np.empty_like(a)
    _lib().splat_sum_f32(_p(a), _p(f), _p(out), n, c, h, w)
    return torch.from_numpy(out)


def softsplat(tenIn, tenFlow, tenMetric, strMode):
    """softsplat.py:232-274 — 'sum' and 'soft' only (the only modes reached from control_utils.py:14,66)."""
    if strMode == "sum":
        assert tenMetric is None
        return splat_sum(tenIn, tenFlow)
    assert strMode == "soft" and tenMetric is not None
    a = np.ascontiguousarray(tenIn.detach().cpu().float().numpy())
    f = np.ascontiguousarray(tenFlow.detach().cpu().float().numpy())
    m = np.ascontiguousarray(tenMetric.detach().cpu().float().numpy())
    n, c, h, w = a.shape
    assert f.shape == (n, 2, h, w) and m.shape == (n, 1, h, w)
    out = np.empty_like(a)
    _lib().splat_soft_f32(_p(a), _p(f), _p(m), _p(out), n, c, h, w)
    return torch.from_numpy(out)


def softsplat_torch(tenIn, tenFlow, tenMetric, strMode="soft"):
    """Same arithmetic in vectorised torch (index_add_); used as the bulk CPU baseline / cross-check of the
    C loop.  Summation order differs from the C loop (per-corner passes), so agreement is to fp32 rounding."""
    x = tenIn.float()
    if strMode == "soft":
        e = tenMetric.float().exp()
        x = torch.cat([x * e, e], 1)
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    fx = gx[None] + tenFlow[:, 0].float()
    fy = gy[None] + tenFlow[:, 1].float()
    fin = torch.isfinite(fx) & torch.isfinite(fy)
    fx = torch.where(fin, fx, torch.zeros_like(fx))
    fy = torch.where(fin, fy, torch.zeros_like(fy))
    x0 = torch.floor(fx)
    y0 = torch.floor(fy)
    out = torch.zeros(n, c, h * w)
    xf = x.reshape(n, c, h * w)
    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
        cx = x0 + dx
        cy = y0 + dy
        # weights exactly as softsplat.py:315-318
        wx = (x0 + 1 - fx) if dx == 0 else (fx - x0)
        wy = (y0 + 1 - fy) if dy == 0 else (fy - y0)
        ok = fin & (cx >= 0) & (cx < w) & (cy >= 0) & (cy < h)
        wgt = torch.where(ok, wx * wy, torch.zeros_like(wx)).reshape(n, 1, h * w)
        idx = torch.where(ok, cy * w + cx, torch.zeros_like(cx)).long().reshape(n, h * w)
        for b in range(n):
            out[b].index_add_(1, idx[b], xf[b] * wgt[b])
    out = out.reshape(n, c, h, w)
    if strMode == "soft":
        out = out[:, :-1] / (out[:, -1:] + 0.0000001)
    return out
