"""ctypes binding of libsin3dm_hip.so (C ABI: include/sin3dm_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, every compute
entry point raises.  torch is used by the callers only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsin3dm_hip.so")

c_fp = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)

ABI_VERSION = 4
TAB_ROWS = ("sqrt_recip", "sqrt_recipm1", "coef1", "coef2", "logvar", "acp", "acp_prev")
CARRY_OUT, CARRY_IN = 1, 2          # s3d_unet_step_film_carry flags
STEP_DDPM, STEP_DDIM, STEP_MEAN_ONLY = 0, 1, 2
CARRY_OUT, CARRY_IN = 1, 2               # s3d_unet_step_film_carry flags (include/sin3dm_hip.h)
MEAN_START_X, MEAN_EPSILON = 0, 1
ERR_INVALID, ERR_MISSING, ERR_HIP, ERR_UNSUPPORTED = -1, -2, -3, -4
MAX_LANES = 16


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("model_channels", C.c_int32), ("out_channels", C.c_int32),
                ("num_res_blocks", C.c_int32), ("n_levels", C.c_int32), ("channel_mult", C.c_int32 * 8),
                ("use_scale_shift_norm", C.c_int32), ("is_rollout", C.c_int32)]


class SamplerArgs(C.Structure):
    _fields_ = [("mode", C.c_int32), ("mean_type", C.c_int32), ("clip_denoised", C.c_int32),
                ("is_mask_t0", C.c_int32), ("eta", C.c_float), ("T", C.c_int32), ("batch", C.c_int64),
                ("per_sample", C.c_int64), ("model_out", C.c_void_p), ("x", C.c_void_p), ("noise", C.c_void_p),
                ("t", C.c_void_p), ("tables", C.c_void_p), ("y0", C.c_void_p), ("mask", C.c_void_p),
                ("sample", C.c_void_p), ("pred_xstart", C.c_void_p), ("mean", C.c_void_p)]


PROF_CLASSES = 4


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * PROF_CLASSES), ("flops", C.c_double * PROF_CLASSES), ("launches", C.c_int64 * PROF_CLASSES),
                ("forwards", C.c_int64), ("mfma_flops", C.c_double * PROF_CLASSES)]


class AeLossCfg(C.Structure):
    _fields_ = [("sdf_loss", C.c_int32), ("tex_loss", C.c_int32), ("sdf_threshold", C.c_float),
                ("tex_threshold_ratio", C.c_float), ("tex_weight", C.c_float)]


class DecoderCfg(C.Structure):
    _fields_ = [("geo_feat_channels", C.c_int32), ("tex_feat_channels", C.c_int32), ("feat_channel_up", C.c_int32),
                ("mlp_hidden_channels", C.c_int32), ("mlp_hidden_layers", C.c_int32), ("tex_channels", C.c_int32)]


# name -> (restype, argtypes); every symbol include/sin3dm_hip.h declares
SIGNATURES = {
    "s3d_abi_version": (C.c_int, []),
    "s3d_last_error": (C.c_char_p, []),
    "s3d_device_count": (C.c_int, []),
    "s3d_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "s3d_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "s3d_unet_create": (C.c_int, [C.POINTER(UNetCfg), C.POINTER(C.c_void_p)]),
    "s3d_unet_destroy": (None, [C.c_void_p]),
    "s3d_unet_num_params": (C.c_int, [C.c_void_p]),
    "s3d_unet_param_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), c_i64p, C.POINTER(C.c_int)]),
    "s3d_unet_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64p, C.c_int]),
    "s3d_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "s3d_unet_film_width": (C.c_int, [C.c_void_p]),
    "s3d_unet_film": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "s3d_unet_select_lane": (C.c_int, [C.c_void_p, C.c_int]),
    "s3d_unet_current_lane": (C.c_int, [C.c_void_p]),
    "s3d_unet_forward_film": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "s3d_unet_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "s3d_unet_profile_classes": (C.c_int, [C.c_void_p, C.c_int]),
    "s3d_unet_profile_read": (C.c_int, [C.c_void_p, C.POINTER(Profile)]),
    "s3d_unet_profile_kernel": (C.c_char_p, [C.c_void_p, C.c_int]),
    "s3d_sampler_step": (C.c_int, [C.POINTER(SamplerArgs), C.c_void_p]),
    "s3d_unet_step_film": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(SamplerArgs), C.c_void_p, C.c_void_p]),
    "s3d_unet_step_film_carry": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(SamplerArgs), C.c_void_p, C.c_void_p, C.c_int]),
    "s3d_op_triplane_conv": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.c_void_p]),
    "s3d_op_triplane_norm_silu": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.c_void_p]),
    "s3d_op_triplane_resample": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.c_int), C.c_int, C.c_void_p]),
    "s3d_op_timestep_embed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "s3d_op_triplane_resblock": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "s3d_decoder_create": (C.c_int, [C.POINTER(DecoderCfg), C.POINTER(C.c_void_p)]),
    "s3d_decoder_destroy": (None, [C.c_void_p]),
    "s3d_decoder_num_params": (C.c_int, [C.c_void_p]),
    "s3d_decoder_param_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), c_i64p, C.POINTER(C.c_int)]),
    "s3d_decoder_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64p, C.c_int]),
    "s3d_decoder_prepare_triplane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p]),
    "s3d_decoder_decode_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_fp, C.c_int, C.c_void_p,
                                            C.c_void_p]),
    "s3d_decoder_grid_dims": (C.c_int, [c_fp, C.c_int, C.POINTER(C.c_int)]),
    "s3d_decoder_decode_grid": (C.c_int, [C.c_void_p, C.c_int, c_fp, C.c_void_p, C.c_void_p]),
    # iso-surface extraction
    "s3d_mc_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "s3d_mc_destroy": (None, [C.c_void_p]),
    "s3d_mc_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                               c_i64p, c_i64p, C.c_void_p]),
    "s3d_mc_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "s3d_mesh_components": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    # auto-encoder training tier
    "s3d_ae_create": (C.c_int, [C.POINTER(DecoderCfg), C.POINTER(C.c_void_p)]),
    "s3d_ae_destroy": (None, [C.c_void_p]),
    "s3d_ae_num_params": (C.c_int, [C.c_void_p]),
    "s3d_ae_param_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), c_i64p, C.POINTER(C.c_int), c_i64p]),
    "s3d_ae_param_numel": (C.c_int64, [C.c_void_p, c_i64p]),
    "s3d_ae_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "s3d_ae_repack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "s3d_ae_set_volume": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "s3d_ae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3d_ae_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_fp, C.c_void_p, C.c_void_p]),
    "s3d_ae_loss_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_fp, C.POINTER(AeLossCfg),
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    # training tier
    "s3d_unet_param_numel": (C.c_int64, [C.c_void_p]),
    "s3d_unet_param_offset": (C.c_int, [C.c_void_p, C.c_int, c_i64p]),
    "s3d_unet_train_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "s3d_unet_repack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "s3d_unet_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]),
    "s3d_unet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3d_unet_backward_marked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "s3d_train_q_sample": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                     C.c_void_p, C.c_void_p]),
    "s3d_train_mse_terms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "s3d_train_mse_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "s3d_train_adamw_ema": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), c_fp, C.c_int,
                                      C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                      C.c_void_p]),
}

_lib = None


class Sin3DMHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (no GPU needed just to load and inspect symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C sin3dm_amd/csrc`. sin3dm_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.s3d_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"libsin3dm_hip.so ABI {v} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def last_error():
    return (load().s3d_last_error() or b"").decode(errors="replace")


def check(rc):
    """Map a status code to the exception type the reference would raise for the same mistake."""
    if rc == 0:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise AssertionError(msg)
    if rc == ERR_MISSING:
        raise RuntimeError("Error(s) in loading state_dict: " + msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise Sin3DMHipError(msg)


def require_gpu(t=None):
    """The product path runs on MI355X only."""
    import torch
    if t is not None and not t.is_cuda:
        raise Sin3DMHipError("sin3dm_amd runs on an MI355X: tensor is on %s and there is no CPU fallback" % t.device)
    if not torch.cuda.is_available():
        raise Sin3DMHipError("sin3dm_amd needs a visible MI355X (torch.cuda.is_available() is False); "
                             "there is no CPU fallback")


def set_option(name, value):
    """Select a kernel form process-wide (include/sin3dm_hip.h: s3d_set_option); value None = the library's own choice."""
    check(load().s3d_set_option(str(name).encode(), None if value is None else str(value).encode()))


def get_option(name):
    v = C.c_int(0)
    check(load().s3d_get_option(str(name).encode(), C.byref(v)))
    return None if v.value == -1 else v.value


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def ptr3(ts):
    return (C.c_void_p * 3)(*[C.c_void_p(t.data_ptr()) for t in ts])
