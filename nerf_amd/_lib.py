"""ctypes binding of libnerfx.so (the C ABI declared in include/nerfx.h).

There is no CPU fallback: if the library is missing the import fails loudly.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- first: libnerfx.so must bind to the HIP runtime torch has loaded, so that
#                               torch's streams and device pointers are valid inside the library

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERFX_LIB", os.path.join(_HERE, "libnerfx.so"))   # override: diagnostic builds

ENC_POSITIONAL, ENC_FOURIER = 0, 1
ACT_RELU, ACT_LEAKY_RELU, ACT_SILU = 0, 1, 2
PREC_FP32, PREC_BF16 = 0, 1


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("enc_x_kind", "enc_x_n", "enc_d_kind", "enc_d_n",
                                         "width", "depth", "activ", "resid")]


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("offset", C.c_uint64)]


class DeviceInfo(C.Structure):
    _fields_ = [("compute_units", C.c_int32), ("wavefront", C.c_int32),
                ("lds_bytes_per_cu", C.c_int32), ("clock_khz", C.c_int32),
                ("hbm_bytes", C.c_int64), ("arch", C.c_char * 32)]


_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes): exactly the declarations of include/nerfx.h
SIGNATURES = {
    "nerfx_version": (_i, []),
    "nerfx_last_error": (C.c_char_p, []),
    "nerfx_device_info": (_i, [_i, C.POINTER(DeviceInfo)]),
    "nerfx_model_create": (_i, [C.POINTER(ModelDesc), C.POINTER(_p), _i, _p, C.POINTER(_p)]),
    "nerfx_model_refresh": (_i, [_p, C.POINTER(_p), _i, _p]),
    "nerfx_model_refresh_precision": (_i, [_p, C.POINTER(_p), _i, _i, _p]),
    "nerfx_model_destroy": (None, [_p]),
    "nerfx_sample_encode_fwd": (_i, [_p, _p, _l, _f, _f, _i, _p, _i, _i, _p, _i, _i, _p,
                                     _p, _p, _p, _p, _p, _p]),
    "nerfx_encode_fwd": (_i, [_p, _l, _i, _i, _p, _p, _p]),
    "nerfx_mlp_fwd_points": (_i, [_p, _i, _p, _p, _l, _p, _p, _p]),
    "nerfx_mlp_fwd_rays": (_i, [_p, _i, _p, _p, _p, _l, _i, _p, _p, _p]),
    "nerfx_mlp_train_workspace_bytes": (_l, [_p, _i, _l]),
    "nerfx_mlp_fwd_train_points": (_i, [_p, _i, _p, _p, _l, _p, _p, _p, _l, _p]),
    "nerfx_mlp_fwd_train_rays": (_i, [_p, _i, _p, _p, _p, _l, _i, _p, _p, _p, _l, _p]),
    "nerfx_mlp_bwd": (_i, [_p, _i, _l, _p, _p, _p, _p, _p, _l, C.POINTER(_p), _i, _p]),
    "nerfx_mlp_bwd_checked": (_i, [_p, _i, _l, _p, _p, _p, _p, _p, _l, C.POINTER(_p), _i, _p, _p]),
    "nerfx_composite_fwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p]),
    "nerfx_composite_bwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p]),
    "nerfx_segment_lengths": (_i, [_p, _p, _l, _i, _p, _p]),
    "nerfx_pdf_sample": (_i, [_p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p]),
    "nerfx_coarse_pdf_sample": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _p, _p]),
    "nerfx_searchsorted_right": (_i, [_p, _p, _l, _i, _i, _p, _p]),
    "nerfx_orbit_poses": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _p, _p]),
    "nerfx_pinhole_directions": (_i, [_i, _i, _f, _p, _p]),
    "nerfx_project_rays": (_i, [_p, _l, _p, _p, _p, _p]),
    "nerfx_camera_rays": (_i, [_p, _i, _i, _f, _l, _l, _p, _p, _p]),
    "nerfx_frame_workspace_bytes": (_l, []),
    "nerfx_frame_finish": (_i, [_p, _p, _l, _p, _l, _p, _p, _p, _p, _p]),
    "nerfx_gather_rays": (_i, [_p, _p, _p, _p, _l, _l, _l, _p, _p, _p, _p]),
    "nerfx_mse_loss": (_i, [_p, _p, _l, _p, _p, _p]),
    "nerfx_mse_loss_scaled": (_i, [_p, _p, _l, _p, _p, _p, _p, _p]),
    "nerfx_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    "nerfx_adam_step_scaled": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    "nerfx_voxel_sample": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, C.c_uint64, C.c_uint64, _p, _p, _p, _p]),
    "nerfx_distill_loss": (_i, [_p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p]),
    "nerfx_debug_set_stamps": (None, [_p]),
    "nerfx_debug_set_m16": (None, [_i]),
    "nerfx_debug_set_k4_fast": (None, [_i]),
    "nerfx_debug_set_narrow": (None, [_i]),
    "nerfx_debug_folded_head": (_i, [_p, _p, _p, _p]),
    "nerfx_debug_unfold_grads": (_i, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "nerfx_render_workspace_bytes": (_l, [_l, _i, _i]),
    "nerfx_render_fwd": (_i, [_p, _p, _i, _p, _p, _l, _f, _f, _i, _i, _p, _p, _p, _l,
                              _p, _p, _p, _p, _p]),
    "nerfx_render_fwd_rng": (_i, [_p, _p, _i, _p, _p, _l, _f, _f, _i, _i, _p, _p, C.POINTER(Rng), _p, _l,
                                  _p, _p, _p, _p, _p]),
    "nerfx_rng_uniform": (_i, [C.c_uint64, C.c_uint64, _i, _l, _p, _p]),
    "nerfx_render_train_workspace_bytes": (_l, [_p, _i, _l, _i, _i]),
    "nerfx_render_train_fwd": (_i, [_p, _p, _i, _p, _p, _l, _f, _f, _i, _i, _p, _p, C.POINTER(Rng), _p, _l,
                                    _p, _p, _p, _p, _p]),
    "nerfx_render_bwd": (_i, [_p, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _l, C.POINTER(_p), _i, _p]),
    "nerfx_render_bwd_checked": (_i, [_p, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _l, C.POINTER(_p), _i, _p, _p]),
    "nerfx_render_static_max_rays": (_l, [_i, _i]),
    "nerfx_render_fwd_static": (_i, [_p, _p, _i, _p, _p, _l, _f, _f, _i, _i, _p, _p,
                                     _p, _p, _p, _p, _p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C nerf_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). nerf_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class NerfxError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise NerfxError(f"nerfx error {rc}: {lib.nerfx_last_error().decode()}")
