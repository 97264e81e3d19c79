"""
pixel_nerf_multiscale_amd — MI355X-native pixelNeRF volume-render hot path behind the reference's
PixelNeRFNet / NeRFRenderer Python API.  Kernels: csrc/*.hip -> lib/libpnr_hip.so (C ABI: include/pnr.h).
"""
from . import util
from .model import PixelNeRFNet, make_model
from .render import NeRFRenderer

__all__ = ["PixelNeRFNet", "NeRFRenderer", "make_model", "util"]
