"""MI355X-native decode hot path of DiffCodec (ControlNet-conditioned SD-1.5 sampling + VAE).

Importable as ``diffcodec_amd`` (the directory name carries hyphens; ``diffcodec_amd/__init__.py`` aliases it).
Compute lives in ``csrc/`` (hand-written HIP for gfx950 behind the C-ABI of ``include/diffcodec_hip.h``);
the Python modules here mirror the reference's operator interface (pipeline.py, controlnet/flownet.py).
"""
__all__ = ["weights", "synthetic"]
