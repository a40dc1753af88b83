"""vit-gan_amd: MI355X-native (gfx950) engine for the ViTGAN G/D training hot path.

Python keeps the reference's nn.Module surface (src/v2/modules.py) and drives
hand-written HIP kernels through a C ABI (include/vitgan_hip.h).  Import as
``vit_gan_amd`` (alias module at the repo root).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
from .config import Config  # noqa: E402,F401
