"""Import alias: ``import diffcodec_amd`` loads the package directory
``diffcodec-controlling-latent-diffusion-for-perceptual-video-compression_amd/`` (hyphens are not importable
with the ``import`` statement)."""
import importlib.util
import os
import sys

_NAME = "diffcodec-controlling-latent-diffusion-for-perceptual-video-compression_amd"
_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), _NAME)
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
