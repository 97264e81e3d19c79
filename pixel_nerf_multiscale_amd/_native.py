"""
ctypes binding of libpnr_hip.so (include/pnr.h).  The library is REQUIRED: importing this module
raises if it has not been built (python -m pixel_nerf_multiscale_amd.build_native) — there is no
CPU or PyTorch fallback for the render path.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNR_LIB") or os.path.join(HERE, "lib", "libpnr_hip.so")   # PNR_LIB: diagnostic builds (tools/dev)

PNR_MAX_LEVELS = 5
PNR_MAX_BLOCKS = 8
PNR_F32, PNR_BF16, PNR_F16 = 0, 1, 2
PNR_BF16X3 = 3     # training entry points only
PRECISIONS = {"fp32": PNR_F32, "f32": PNR_F32, "bf16": PNR_BF16, "fp16": PNR_F16, "f16": PNR_F16, "bf16x3": PNR_BF16X3}
COMBINE = {"average": 0, "max": 1}

_fp = C.c_void_p  # device pointers travel as integers


class pnr_mlp(C.Structure):
    _fields_ = [
        ("d_in", C.c_int32), ("d_latent", C.c_int32), ("d_hidden", C.c_int32), ("d_out", C.c_int32),
        ("n_blocks", C.c_int32), ("combine_layer", C.c_int32), ("combine_type", C.c_int32), ("packed_objs", C.c_int32),
        ("lin_in_w", _fp), ("lin_in_b", _fp),
        ("lin_z_w", _fp * PNR_MAX_BLOCKS), ("lin_z_b", _fp * PNR_MAX_BLOCKS),
        ("fc0_w", _fp * PNR_MAX_BLOCKS), ("fc0_b", _fp * PNR_MAX_BLOCKS),
        ("fc1_w", _fp * PNR_MAX_BLOCKS), ("fc1_b", _fp * PNR_MAX_BLOCKS),
        ("lin_out_w", _fp), ("lin_out_b", _fp),
        ("packed", _fp), ("packed_bytes", C.c_uint64), ("packed_dtype", C.c_int32), ("packed_texels", C.c_int32),
    ]


class pnr_views(C.Structure):
    _fields_ = [
        ("n_objs", C.c_int32), ("n_views", C.c_int32),
        ("w2c", _fp), ("focal", _fp), ("c", _fp),
        ("n_focal", C.c_int32), ("n_c", C.c_int32), ("n_levels", C.c_int32), ("reserved0", C.c_int32),
        ("latent", _fp * PNR_MAX_LEVELS),
        ("lat_c", C.c_int32 * PNR_MAX_LEVELS), ("lat_h", C.c_int32 * PNR_MAX_LEVELS), ("lat_w", C.c_int32 * PNR_MAX_LEVELS),
        ("latent_packed", _fp * PNR_MAX_LEVELS), ("packed_dtype", C.c_int32), ("reserved1", C.c_int32),
        ("uv_scale_x", C.c_float * PNR_MAX_LEVELS), ("uv_scale_y", C.c_float * PNR_MAX_LEVELS),
    ]


class pnr_params(C.Structure):
    _fields_ = [
        ("n_coarse", C.c_int32), ("n_fine", C.c_int32), ("n_fine_depth", C.c_int32), ("white_bkgd", C.c_int32),
        ("lindisp", C.c_int32), ("use_code_viewdirs", C.c_int32), ("num_freqs", C.c_int32), ("precision", C.c_int32),
        ("depth_std", C.c_float), ("freq_factor", C.c_float), ("train_tape_fp32", C.c_int32), ("park_fp32", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class pnr_mlp_grads(C.Structure):
    _fields_ = [
        ("lin_in_w", _fp), ("lin_in_b", _fp),
        ("lin_z_w", _fp * PNR_MAX_BLOCKS), ("lin_z_b", _fp * PNR_MAX_BLOCKS),
        ("fc0_w", _fp * PNR_MAX_BLOCKS), ("fc0_b", _fp * PNR_MAX_BLOCKS),
        ("fc1_w", _fp * PNR_MAX_BLOCKS), ("fc1_b", _fp * PNR_MAX_BLOCKS),
        ("lin_out_w", _fp), ("lin_out_b", _fp),
    ]


class pnr_noise(C.Structure):
    _fields_ = [("noise_c", _fp), ("u", _fp), ("r", _fp), ("g", _fp), ("ray_index_obj_stride", C.c_int64)]


class pnr_outputs(C.Structure):
    _fields_ = [("coarse_rgb", _fp), ("coarse_depth", _fp), ("coarse_weights", _fp), ("fine_rgb", _fp),
                ("fine_depth", _fp), ("fine_weights", _fp), ("z_coarse", _fp), ("z_fine", _fp),
                ("ev_point_begin", _fp), ("ev_point_end", _fp), ("rgb_stride", C.c_int32), ("depth_stride", C.c_int32),
                ("coarse_weights_stride", C.c_int32), ("fine_weights_stride", C.c_int32)]


# every symbol include/pnr.h declares: name -> (restype, argtypes)
_i32, _i64, _u64, _f = C.c_int32, C.c_int64, C.c_uint64, C.c_float
PROTOTYPES = {
    "pnr_version": (_i32, []),
    "pnr_error_string": (C.c_char_p, [_i32]),
    "pnr_packed_mlp_bytes": (_u64, [C.POINTER(pnr_mlp)]),
    "pnr_pack_mlp": (_i32, [C.POINTER(pnr_mlp), _i32, _fp, _u64, _fp]),
    "pnr_packed_mlp_projected_bytes": (_u64, [C.POINTER(pnr_mlp), C.POINTER(pnr_views)]),
    "pnr_pack_mlp_projected": (_i32, [C.POINTER(pnr_mlp), C.POINTER(pnr_views), _i32, _fp, _u64, _fp]),
    "pnr_packed_latent_bytes": (_u64, [C.POINTER(pnr_views)]),
    "pnr_pack_latents": (_i32, [C.POINTER(pnr_views), _i32, _fp, _u64, C.POINTER(C.c_uint64), _fp]),
    "pnr_sample_coarse": (_i32, [_fp, _i64, _i32, _i32, _fp, _u64, _i64, _fp, _fp]),
    "pnr_composite": (_i32, [_fp, _fp, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _fp]),
    "pnr_sample_fine": (_i32, [_fp, _fp, _fp, _fp, _i64, _i32, _i32, _i32, _f, _i32, _fp, _fp, _fp, _u64, _i64, _fp, _fp]),
    "pnr_point_mlp": (_i32, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_views), _fp, _fp, _i32, _fp, _fp,
                             _i64, _i64, _fp, _fp, _u64, _fp]),
    "pnr_workspace_bytes": (_u64, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_views), _i64]),
    "pnr_render": (_i32, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_mlp), C.POINTER(pnr_views), _fp, _i64,
                          _i64, C.POINTER(pnr_noise), _u64, _i64, C.POINTER(pnr_outputs), _fp, _u64, _fp]),
    "pnr_resnetfc_workspace_bytes": (_u64, [C.POINTER(pnr_mlp), _i32]),
    "pnr_resnetfc_forward": (_i32, [C.POINTER(pnr_mlp), _fp, _i64, _i32, _i64, _fp, _fp, _u64, _fp]),
    "pnr_index_latent": (_i32, [C.POINTER(pnr_views), _fp, _i64, _i32, _fp, _fp]),
    "pnr_render_camera": (_i32, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_mlp), C.POINTER(pnr_views),
                                 C.POINTER(C.c_float), _i32, _i32, _f, _f, _f, _f, _f, _f, _i64, _i64, C.POINTER(pnr_noise), _u64,
                                 _i64, C.POINTER(pnr_outputs), _fp, _u64, _fp]),
    "pnr_train_tape_bytes": (_u64, [C.POINTER(pnr_mlp), C.POINTER(pnr_views), _i64]),
    "pnr_train_tape_bytes_for": (_u64, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_views), _i64]),
    "pnr_train_bwd_workspace_bytes": (_u64, [C.POINTER(pnr_mlp), C.POINTER(pnr_views), _i64]),
    "pnr_point_mlp_train_fwd": (_i32, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_views), _fp, _fp, _i32,
                                       _fp, _fp, _i64, _i64, _fp, _fp, _u64, _fp]),
    "pnr_point_mlp_bwd": (_i32, [C.POINTER(pnr_params), C.POINTER(pnr_mlp), C.POINTER(pnr_views), _fp, _fp, _i32, _fp, _fp,
                                 _i64, _i64, _fp, _fp, _fp, _u64, C.POINTER(pnr_mlp_grads), C.POINTER(C.c_void_p), _fp, _fp,
                                 _fp, _u64, _fp]),
    "pnr_composite_bwd": (_i32, [_fp, _fp, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
    "pnr_sample_fine_bwd": (_i32, [_fp, _fp, _i64, _i32, _i32, _i32, _f, _fp, _u64, _i64, _fp, _fp, _fp, _fp]),
    "pnr_gen_rays": (_i32, [C.POINTER(C.c_float), _i32, _i32, _f, _f, _f, _f, _f, _f, _i64, _i64, _fp, _fp]),
    "pnr_event_create": (_i32, [C.POINTER(C.c_void_p)]),
    "pnr_event_record": (_i32, [_fp, _fp]),
    "pnr_event_elapsed_ms": (_i32, [_fp, _fp, C.POINTER(C.c_float)]),
    "pnr_event_destroy": (_i32, [_fp]),
    "pnr_debug_gemm_grid": (C.c_int64, [_i32, _i32, _i32, _i32, _i32]),
    "pnr_debug_gemm_tile": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_int32)]),
}


class Stream:
    """HIP stream handle + the ordinal of the device it belongs to.  Travels through the C ABI as the plain handle
    (`_as_parameter_`); the ordinal is what the launch guard below needs — handle 0 (the default stream) does not say
    which device it means."""
    __slots__ = ("_as_parameter_", "device")

    def __init__(self, handle, device):
        self._as_parameter_ = handle
        self.device = device


def _guarded(fn):
    """The library launches on the calling thread's CURRENT HIP device (it never calls hipSetDevice: include/pnr.h
    'Threading').  The reference picks its GPU with util.get_cuda(gpu_id) and no set_device (eval/eval.py:94), so a
    tensor may live on cuda:N while the current device is 0: every entry point that takes a stream runs under the
    stream's device."""
    def call(*args):
        s = args[-1] if args else None
        if isinstance(s, Stream) and s.device != torch.cuda.current_device():
            with torch.cuda.device(s.device):
                return fn(*args)
        return fn(*args)
    call.__name__ = getattr(fn, "__name__", "pnr_fn")
    return call


class _Lib:
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m pixel_nerf_multiscale_amd.build_native` "
            "(hipcc --offload-arch=gfx950).  The render path has no CPU/PyTorch fallback.")
    cdll = C.CDLL(LIB_PATH)
    lib = _Lib()
    lib._cdll = cdll
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(cdll, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
        takes_stream = bool(args) and args[-1] is _fp and name not in ("pnr_event_destroy",)
        setattr(lib, name, _guarded(fn) if takes_stream else fn)
    return lib


lib = _load()


class PnrError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = lib.pnr_error_string(rc).decode()
        if -6 <= rc < 0:
            raise ValueError(f"{what}: {msg} (code {rc})")
        raise PnrError(f"{what}: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor for the C ABI (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libpnr_hip needs tensors on a HIP device (cuda:N); got a CPU tensor")
    if t.dtype != torch.float32 and t.dtype != torch.uint8:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def f32c(t, device=None):
    """float32 + contiguous (+ device) view/copy of t."""
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def current_stream(device):
    """torch's current stream on `device` for the C ABI (see Stream)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return Stream(torch.cuda.current_stream(idx).cuda_stream, idx)


def same_device(*tensors):
    """The one device all the given tensors live on; raises when they differ (the library takes raw pointers and
    cannot tell)."""
    devs = {t.device for t in tensors if t is not None}
    if len(devs) != 1:
        raise ValueError(f"tensors of one native call must share a device, got {sorted(str(d) for d in devs)}")
    dev = devs.pop()
    if dev.type != "cuda":
        raise RuntimeError("libpnr_hip needs tensors on a HIP device (cuda:N); got a CPU tensor")
    return dev
