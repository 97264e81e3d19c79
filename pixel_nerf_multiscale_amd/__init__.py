"""
pixel_nerf_multiscale_amd — MI355X-native pixelNeRF volume-render hot path behind the reference's
PixelNeRFNet / NeRFRenderer Python API.  Kernels: csrc/*.hip -> lib/libpnr_hip.so (C ABI: include/pnr.h).

The public names are resolved on first use, so that `python -m pixel_nerf_multiscale_amd.build_native` can run
before the library exists; touching any of them without the built library raises (there is no fallback path).
"""
import importlib

__all__ = ["PixelNeRFNet", "NeRFRenderer", "make_model", "util"]

_LAZY = {"PixelNeRFNet": ".model", "make_model": ".model", "NeRFRenderer": ".render", "util": ".util",
         "model": ".model", "render": ".render", "parallel": ".parallel", "evalio": ".evalio", "_native": "._native"}


def __getattr__(name):
    target = _LAZY.get(name)
    if target is None:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
    mod = importlib.import_module(target, __name__)
    value = mod if target == "." + name else getattr(mod, name)
    globals()[name] = value
    return value
