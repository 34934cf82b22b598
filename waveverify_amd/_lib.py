"""ctypes binding of libwaveverify_hip.so (include/waveverify_hip.h).

There is no CPU fallback: if the shared library is missing, loading fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libwaveverify_hip.so")
WV_MAX_STRIDES = 8

WV_KIND = {"generator": 0, "detector": 1, "locator": 2}


class WvConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dimension", C.c_int32), ("msg_dimension", C.c_int32),
        ("channels_enc", C.c_int32), ("channels_dec", C.c_int32), ("n_fft_base", C.c_int32),
        ("n_residual_enc", C.c_int32), ("n_residual_dec", C.c_int32), ("n_strides", C.c_int32),
        ("strides", C.c_int32 * WV_MAX_STRIDES),
        ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32),
        ("residual_kernel_size", C.c_int32), ("dilation_base", C.c_int32),
        ("zero_init", C.c_int32), ("nbits", C.c_int32), ("output_dim", C.c_int32),
        ("embedding_dim", C.c_int32), ("embedding_layers", C.c_int32), ("freq_bands", C.c_int32),
        ("res_scale_enc", C.c_float), ("res_scale_dec", C.c_float), ("wav_std", C.c_float),
        ("spec_means", C.c_float * (WV_MAX_STRIDES + 1)),
        ("spec_stds", C.c_float * (WV_MAX_STRIDES + 1)),
    ]


_FP = C.POINTER(C.c_float)
_VP = C.c_void_p

# name -> (restype, argtypes); the complete export list of include/waveverify_hip.h
SIGNATURES = {
    "wv_last_error": (C.c_char_p, []),
    "wv_version": (C.c_char_p, []),
    "wv_config_default": (C.c_int, [C.c_int, C.POINTER(WvConfig)]),
    "wv_model_create": (C.c_int, [C.POINTER(WvConfig), C.POINTER(_VP)]),
    "wv_model_destroy": (None, [_VP]),
    "wv_model_num_params": (C.c_int, [_VP]),
    "wv_model_param_info": (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "wv_model_set_param": (C.c_int, [_VP, C.c_char_p, _VP, C.c_int64]),
    "wv_model_set_param_wn": (C.c_int, [_VP, C.c_char_p, _VP, C.c_int64, _VP, C.c_int64]),
    "wv_model_set_stft_basis": (C.c_int, [_VP, C.c_char_p, _VP, C.c_int64]),
    "wv_model_finalize": (C.c_int, [_VP]),
    "wv_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_generator_forward": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, C.c_int, C.c_int, C.c_int,
                                       _VP, C.c_size_t, _VP]),
    "wv_detector_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_locator_forward": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_encoder_forward": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, C.c_int, C.c_int, _VP,
                                     C.c_size_t, _VP]),
    "wv_model_film": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, _VP]),
    "wv_train_unit_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_unit_destroy": (None, [_VP]),
    "wv_train_unit_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_unit_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_int, _VP, C.c_int, C.c_int, _VP]),
    "wv_train_unit_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                         C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_convpre_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_convpre_destroy": (None, [_VP]),
    "wv_train_convpre_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_convpre_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_float, _VP, C.c_int, C.c_int, _VP]),
    "wv_train_convpre_backward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_spec_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_spec_destroy": (None, [_VP]),
    "wv_train_spec_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_spec_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_float, _VP, C.c_int, C.c_int, _VP]),
    "wv_train_spec_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_convpost_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_convpost_destroy": (None, [_VP]),
    "wv_train_convpost_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_convpost_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP, C.c_int, C.c_int, _VP]),
    "wv_train_convpost_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int,
                                             _VP, C.c_size_t, _VP]),
    "wv_train_head_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_head_destroy": (None, [_VP]),
    "wv_train_head_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_head_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_head_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_stft_plan_create": (C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    "wv_stft_plan_destroy": (None, [_VP]),
    "wv_stft_plan_logmag": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _VP]),
    "wv_stft_plan_backward_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int, C.c_int]),
    "wv_stft_plan_backward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _VP, C.c_size_t, _VP]),
    "wv_train_up_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_up_destroy": (None, [_VP]),
    "wv_train_up_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_up_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_int, _VP, C.c_int, C.c_int, _VP]),
    "wv_train_up_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int,
                                       _VP, C.c_size_t, _VP]),
    "wv_train_tail_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_VP)]),
    "wv_train_tail_destroy": (None, [_VP]),
    "wv_train_tail_workspace_bytes": (C.c_size_t, [_VP, C.c_int]),
    "wv_train_tail_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_train_tail_backward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int,
                                         _VP, C.c_size_t, _VP]),
    "wv_train_film_param_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "wv_train_film_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "wv_train_film_forward": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_film_backward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_film_apply": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_train_film_apply_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_block_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "wv_train_block_destroy": (None, [_VP]),
    "wv_train_block_saved_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_block_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_block_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, C.c_size_t, C.c_int, C.c_int, _VP]),
    "wv_train_block_backward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int,
                                          _VP, C.c_size_t, _VP]),
    "wv_train_bce_workspace_bytes": (C.c_size_t, []),
    "wv_train_bce_logits": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_float, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_l1": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, C.c_size_t, _VP, C.c_size_t, _VP]),
    "wv_train_sumsq": (C.c_int, [_VP, C.c_size_t, _VP, _VP, C.c_size_t, _VP]),
    "wv_train_adamw": (C.c_int, [_VP, _VP, _VP, _VP, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                 _VP, C.c_float, _VP]),
    "wv_aug_localize_sequence": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP,
                                           _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_aug_backward": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_aug_sequence": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP,
                                  C.c_int, C.c_int, C.c_int, _VP]),
    "wv_fx_resample": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_fx_fir_bank": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_fx_fold_replicate": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_fx_resample_adjoint": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_profile_enable": (C.c_int, [C.c_int]),
    "wv_profile_reset": (C.c_int, []),
    "wv_profile_collect": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    "wv_op_pw_dw": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                              C.c_int, _VP, C.c_float, _VP]),
    "wv_train_half_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "wv_train_half_destroy": (None, [_VP]),
    "wv_train_half_workspace_bytes": (C.c_size_t, [_VP, C.c_int, C.c_int]),
    "wv_train_half_forward": (C.c_int, [_VP] * 7 + [C.c_float, _VP, C.c_int, C.c_int, _VP]),
    "wv_train_half_backward": (C.c_int, [_VP] * 6 + [C.c_float] + [_VP] * 7 + [C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_train_last_error": (C.c_char_p, []),
    "wv_train_fold_weight": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP]),
    "wv_op_spec_block": (C.c_int, [_VP] * 6 + [C.c_int] * 5 + [C.c_float] * 4 + [_VP]),
    "wv_h16_round_host": (C.c_int, [_VP, _VP, C.c_int64]),
    "wv_h16_from_f32": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _VP]),
    "wv_h16_to_f32": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_h16_conv_pre": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _VP]),
    "wv_h16_resblock": (C.c_int, [_VP, C.c_float] + [_VP] * 8 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _VP]),
    "wv_h16_conv": (C.c_int, [_VP] * 8 + [C.c_int] * 7 + [C.c_float, C.c_float, _VP]),
    "wv_h16_spec_block": (C.c_int, [_VP] * 6 + [C.c_int] * 5 + [C.c_float] * 4 + [_VP]),
    "wv_detector_forward_f16": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_locator_forward_f16": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_generator_forward_f16": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, C.c_int, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    "wv_h16_upsample": (C.c_int, [_VP] * 6 + [C.c_int] * 5 + [C.c_float, _VP]),
    "wv_h16_tail": (C.c_int, [_VP] * 5 + [C.c_int] * 5 + [C.c_float, _VP]),
    "wv_h16_l2norm": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "wv_h16_conv_film": (C.c_int, [_VP] * 5 + [C.c_int] + [_VP] * 2 + [C.c_int] * 7 + [C.c_float, _VP]),
    "wv_op_resblock": (C.c_int, [_VP, C.c_float] + [_VP] * 8 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _VP]),
    "wv_op_dw_pw": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, _VP, C.c_float, _VP]),
    "wv_op_stft_logmag": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_float, _VP]),
    "wv_op_conv_pre": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, _VP]),
    "wv_op_tail": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_float, C.c_float, _VP]),
    "wv_op_head": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_int, _VP]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library (once). Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"waveverify_amd: HIP extension not found at {LIB_PATH}; build it with "
            "`python -m waveverify_amd.build` (there is no CPU fallback)")
    # torch first: its bundled libamdhip64 (soname libamdhip64.so.7) must be THE HIP runtime of
    # the process, so that torch's device pointers and streams are valid inside this library.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().wv_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'waveverify_hip'} failed (code {rc}): {msg}")
