"""ctypes binding of liblmufft_hip.so (C ABI declared in include/lmufft.h).

There is no fallback: if the shared library is missing this module raises at
import time, and every entry point raises RuntimeError when the library
reports an error.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LMUFFT_LIBRARY selects another build of the library (kernel experiments)
LIB_PATH = os.environ.get("LMUFFT_LIBRARY") or os.path.join(_HERE, "csrc", "liblmufft_hip.so")

F32 = 0
BF16 = 1
CONV_AUTO = 0
CONV_TOEPLITZ = 1
CONV_FFT = 2
CONV_SCAN = 3
NOALLOC_SCRATCH_BYTES = 8 << 20
ABI_VERSION = 31


class Shape(ctypes.Structure):
    """Mirror of `lmufft_shape` (include/lmufft.h)."""

    _fields_ = [
        ("B", ctypes.c_int32),
        ("T", ctypes.c_int32),
        ("D", ctypes.c_int32),
        ("M", ctypes.c_int32),
        ("Hd", ctypes.c_int32),
        ("ldH", ctypes.c_int32),
    ]


class LnSumItem(ctypes.Structure):
    """Mirror of `lmufft_ln_sum_item` (include/lmufft.h)."""

    _fields_ = [
        ("rows", ctypes.c_int64),
        ("C", ctypes.c_int32),
        ("dgamma", ctypes.c_void_p),
        ("dbeta", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
    ]


# options of lmufft_backward_flags_bf16 (include/lmufft.h)
BWD_GH_MASKED = 1
BWD_MASK_GX = 2
BWD_PARTIAL_SUMS = 4


class QItem(ctypes.Structure):
    """Mirror of `lmufft_q_item` (include/lmufft.h)."""

    _fields_ = [
        ("shape", Shape),
        ("hprep", ctypes.c_void_p),
        ("wprep", ctypes.c_void_p),
        ("qt", ctypes.c_void_p),
    ]


class BwdExtras(ctypes.Structure):
    """Mirror of `lmufft_bwd_extras` (include/lmufft.h)."""

    _fields_ = [
        ("x_mask", ctypes.c_void_p),
        ("qt", ctypes.c_void_p),
        ("q_items", ctypes.POINTER(QItem)),
        ("q_count", ctypes.c_int32),
    ]


class LmuSumItem(ctypes.Structure):
    """Mirror of `lmufft_lmu_sum_item` (include/lmufft.h)."""

    _fields_ = [
        ("shape", Shape),
        ("gw_u", ctypes.c_void_p),
        ("gb_u", ctypes.c_void_p),
        ("gW_h", ctypes.c_void_p),
        ("gb_h", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_size_t),
    ]


class WgradSumItem(ctypes.Structure):
    """Mirror of `lmufft_wgrad_sum_item` (include/lmufft.h)."""

    _fields_ = [
        ("rows", ctypes.c_int64),
        ("in_", ctypes.c_int32),
        ("out", ctypes.c_int32),
        ("dW", ctypes.c_void_p),
        ("db", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_size_t),
    ]


class WprepItem(ctypes.Structure):
    """Mirror of `lmufft_wprep_item` (include/lmufft.h)."""

    _fields_ = [
        ("W", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("N", ctypes.c_int32),
        ("K", ctypes.c_int32),
        ("N_pad", ctypes.c_int32),
        ("Wb", ctypes.c_void_p),
        ("WbT", ctypes.c_void_p),
        ("ldT", ctypes.c_int64),
        ("bias_rounded", ctypes.c_void_p),
    ]


class LrSchedule(ctypes.Structure):
    """Mirror of `lmufft_lr_schedule` (include/lmufft.h)."""

    _fields_ = [
        ("lr0", ctypes.c_double),
        ("max_lr", ctypes.c_double),
        ("decay_rate", ctypes.c_double),
        ("warmup_steps", ctypes.c_int32),
        ("decay_steps", ctypes.c_int32),
        ("scheduled", ctypes.c_int32),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C asr_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "asr_amd has no CPU or eager fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    sp = ctypes.POINTER(Shape)
    lib.lmufft_abi_version.restype = ci
    lib.lmufft_abi_version.argtypes = []
    lib.lmufft_last_error.restype = ctypes.c_char_p
    lib.lmufft_last_error.argtypes = []
    lib.lmufft_forward_workspace_bytes.restype = sz
    lib.lmufft_forward_workspace_bytes.argtypes = [sp, ci]
    lib.lmufft_backward_workspace_bytes.restype = sz
    lib.lmufft_backward_workspace_bytes.argtypes = [sp, ci]
    lib.lmufft_conv_workspace_bytes.restype = sz
    lib.lmufft_conv_workspace_bytes.argtypes = [sp, ci, ci]
    lib.lmufft_forward_f32.restype = ci
    lib.lmufft_forward_f32.argtypes = [sp] + [vp] * 11 + [ci, vp, sz, vp]
    lib.lmufft_backward_f32.restype = ci
    lib.lmufft_backward_f32.argtypes = [sp] + [vp] * 16 + [ci, vp, sz, vp]
    lib.lmufft_conv_forward_f32.restype = ci
    lib.lmufft_conv_forward_f32.argtypes = [sp, vp, vp, vp, vp, ci, vp, sz, vp]
    lib.lmufft_conv_backward_f32.restype = ci
    lib.lmufft_conv_backward_f32.argtypes = [sp, vp, vp, vp, vp, ci, vp, sz, vp]
    lib.lmufft_conv_backward_noalloc_f32.restype = ci
    lib.lmufft_conv_backward_noalloc_f32.argtypes = [sp, vp, vp, vp, vp, ci, vp]
    lib.lmufft_melprep_bytes.restype = sz
    lib.lmufft_melprep_bytes.argtypes = []
    lib.lmufft_melprep.restype = ci
    lib.lmufft_melprep.argtypes = [vp, vp]
    lib.lmufft_mel_frames.restype = ci
    lib.lmufft_mel_frames.argtypes = [ctypes.c_int32]
    lib.lmufft_mel_forward_f32.restype = ci
    lib.lmufft_mel_forward_f32.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, vp,
                                           ctypes.c_int32, ci, vp, vp]
    i32 = ctypes.c_int32
    lib.lmufft_ctc_workspace_bytes.restype = sz
    lib.lmufft_ctc_workspace_bytes.argtypes = [i32, i32]
    lib.lmufft_ctc_greedy.restype = ci
    lib.lmufft_ctc_greedy.argtypes = [vp, ci, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.lmufft_attention_fwd.restype = ci
    lib.lmufft_attention_fwd.argtypes = [vp, vp, vp, ci, vp, i32, i32, i32, i32, vp, vp]
    lib.lmufft_attention_fwd_ld.restype = ci
    lib.lmufft_attention_fwd_ld.argtypes = [vp, vp, vp, ci, ctypes.c_int64, vp, i32, i32, i32, i32, vp, vp]
    lib.lmufft_linear_bf16.restype = ci
    lib.lmufft_linear_bf16.argtypes = [vp, vp, vp, ctypes.c_int64, i32, i32, vp, vp]
    lib.lmufft_linear_ln_bf16.restype = ci
    lib.lmufft_linear_ln_bf16.argtypes = [vp, vp, vp, ctypes.c_int64, i32, i32, vp, vp, ci, vp, vp, ctypes.c_float,
                                          vp, ci, vp]
    lib.lmufft_add_layernorm.restype = ci
    lib.lmufft_add_layernorm.argtypes = [vp, ci, vp, ci, vp, vp, ctypes.c_float, ctypes.c_int64, i32, vp, ci, vp]
    lib.lmufft_fftprep_bytes.restype = sz
    lib.lmufft_fftprep_bytes.argtypes = [sp]
    lib.lmufft_fftprep_f32.restype = ci
    lib.lmufft_fftprep_f32.argtypes = [sp, vp, vp, vp]
    lib.lmufft_hprep_bytes.restype = sz
    lib.lmufft_hprep_bytes.argtypes = [sp]
    lib.lmufft_hprep_bf16.restype = ci
    lib.lmufft_hprep_bf16.argtypes = [sp, vp, vp, vp]
    lib.lmufft_wprep_bytes.restype = sz
    lib.lmufft_wprep_bytes.argtypes = [sp]
    lib.lmufft_wprep_bf16.restype = ci
    lib.lmufft_wprep_bf16.argtypes = [sp, vp, vp, vp]
    lib.lmufft_forward_bf16.restype = ci
    lib.lmufft_forward_bf16.argtypes = [sp] + [vp] * 13 + [ci, vp, sz, vp]
    lib.lmufft_forward_ln_bf16.restype = ci
    lib.lmufft_forward_ln_bf16.argtypes = [sp] + [vp] * 14 + [ctypes.c_float, vp, ci, ci, vp, sz, vp]
    lib.lmufft_conv_forward_bf16.restype = ci
    lib.lmufft_conv_forward_bf16.argtypes = [sp, vp, vp, vp, vp]
    lib.lmufft_conv_backward_bf16.restype = ci
    lib.lmufft_conv_backward_bf16.argtypes = [sp, vp, vp, vp, vp, sz, vp]
    lib.lmufft_backward_bf16.restype = ci
    lib.lmufft_backward_bf16.argtypes = [sp] + [vp] * 17 + [ci, vp, sz, vp]
    lib.lmufft_backward_data_bf16.restype = ci
    lib.lmufft_backward_data_bf16.argtypes = [sp] + [vp] * 13 + [ci, vp, sz, vp]
    lib.lmufft_backward_flags_bf16.restype = ci
    lib.lmufft_backward_flags_bf16.argtypes = [sp] + [vp] * 17 + [ci, vp, sz, vp, ctypes.c_uint32,
                                               ctypes.POINTER(BwdExtras)]
    lib.lmufft_qt_bytes.restype = sz
    lib.lmufft_qt_bytes.argtypes = [sp]
    lib.lmufft_qprep_many_bf16.restype = ci
    lib.lmufft_qprep_many_bf16.argtypes = [ctypes.POINTER(QItem), ctypes.c_int32, vp]
    lib.lmufft_backward_partial_bf16.restype = ci
    lib.lmufft_backward_partial_bf16.argtypes = [sp] + [vp] * 13 + [ci, vp, sz, vp]
    lib.lmufft_backward_weights_sum_many_bf16.restype = ci
    lib.lmufft_backward_weights_sum_many_bf16.argtypes = [ctypes.POINTER(LmuSumItem), ctypes.c_int32, vp]
    lib.lmufft_backward_weights_bf16.restype = ci
    lib.lmufft_backward_weights_bf16.argtypes = [sp] + [vp] * 6 + [vp, sz, vp]
    lib.lmufft_scan_chunk.restype = ctypes.c_int32
    lib.lmufft_scan_chunk.argtypes = []
    lib.lmufft_scanprep_bytes.restype = sz
    lib.lmufft_scanprep_bytes.argtypes = [ctypes.c_int32]
    lib.lmufft_scanprep_build_host.restype = ci
    lib.lmufft_scanprep_build_host.argtypes = [vp, vp, ctypes.c_int32, vp]
    lib.lmufft_conv_scan_workspace_bytes.restype = sz
    lib.lmufft_conv_scan_workspace_bytes.argtypes = [sp]
    lib.lmufft_conv_forward_scan_bf16.restype = ci
    lib.lmufft_conv_forward_scan_bf16.argtypes = [sp, vp, vp, vp, vp, sz, vp]
    lib.lmufft_conv_backward_scan_bf16.restype = ci
    lib.lmufft_conv_backward_scan_bf16.argtypes = [sp, vp, vp, vp, vp, sz, vp]
    u32, i32, i64, f32 = ctypes.c_uint32, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.lmufft_attention_fwd_train.restype = ci
    lib.lmufft_attention_fwd_train.argtypes = [vp, vp, vp, ci, i64, vp, i32, i32, i32, i32, f32, u32, u32, vp, vp, vp]
    lib.lmufft_attention_bwd.restype = ci
    lib.lmufft_attention_bwd.argtypes = [vp, vp, vp, vp, vp, vp, ci, i64, vp, i32, i32, i32, i32, f32, u32, u32,
                                         vp, vp, vp, i64, vp, vp]
    lib.lmufft_layernorm_bwd_workspace_bytes.restype = sz
    lib.lmufft_layernorm_bwd_workspace_bytes.argtypes = [i32]
    lib.lmufft_layernorm_bwd.restype = ci
    lib.lmufft_layernorm_bwd.argtypes = [vp, ci, vp, ci, vp, ci, vp, f32, i64, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.lmufft_add_layernorm_dropout.restype = ci
    lib.lmufft_add_layernorm_dropout.argtypes = [vp, ci, vp, ci, vp, vp, f32, i64, i32, vp, ci, f32, u32, u32, vp, vp]
    lib.lmufft_layernorm_dropout_bwd.restype = ci
    lib.lmufft_layernorm_dropout_bwd.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, f32, i64, i32, f32, u32, u32, vp, vp,
                                                 vp, vp, vp, sz, vp]
    lib.lmufft_layernorm_dropout_bwd_relu.restype = ci
    lib.lmufft_layernorm_dropout_bwd_relu.argtypes = lib.lmufft_layernorm_dropout_bwd.argtypes + [ci]
    lib.lmufft_layernorm_param_sum_many.restype = ci
    lib.lmufft_layernorm_param_sum_many.argtypes = [ctypes.POINTER(LnSumItem), i32, vp]
    lib.lmufft_backward_needs_gm_bf16.restype = ci
    lib.lmufft_backward_needs_gm_bf16.argtypes = [sp, ci, ci]
    lib.lmufft_dropout_epoch.restype = ci
    lib.lmufft_dropout_epoch.argtypes = [vp]
    lib.lmufft_linear_wprep_bf16.restype = ci
    lib.lmufft_linear_wprep_bf16.argtypes = [vp, vp, i32, i32, i32, vp, vp, i64, vp, vp]
    lib.lmufft_linear_wprep_many_bf16.restype = ci
    lib.lmufft_linear_wprep_many_bf16.argtypes = [ctypes.POINTER(WprepItem), i32, vp]
    lib.lmufft_linear_wgrad_workspace_bytes.restype = sz
    lib.lmufft_linear_wgrad_workspace_bytes.argtypes = [i64, i32, i32]
    lib.lmufft_linear_wgrad_bf16.restype = ci
    lib.lmufft_linear_wgrad_bf16.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, sz, vp]
    lib.lmufft_linear_wgrad_slabs_bf16.restype = ci
    lib.lmufft_linear_wgrad_slabs_bf16.argtypes = [vp, vp, i64, i32, i32, vp, sz, vp]
    lib.lmufft_linear_wgrad_sum_bf16.restype = ci
    lib.lmufft_linear_wgrad_sum_bf16.argtypes = [i64, i32, i32, vp, vp, vp, sz, vp]
    lib.lmufft_linear_wgrad_sum_many_bf16.restype = ci
    lib.lmufft_linear_wgrad_sum_many_bf16.argtypes = [ctypes.POINTER(WgradSumItem), i32, vp]
    lib.lmufft_ctc_loss_workspace_bytes.restype = sz
    lib.lmufft_ctc_loss_workspace_bytes.argtypes = [i32, i32, i32]
    lib.lmufft_ctc_loss.restype = ci
    lib.lmufft_ctc_loss.argtypes = [vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, ci, ci, vp, vp, vp, sz, vp]
    lib.lmufft_ctc_mean_loss.restype = ci
    lib.lmufft_ctc_mean_loss.argtypes = [vp, vp, i32, ci, vp, vp]
    lib.lmufft_log_softmax_rows.restype = ci
    lib.lmufft_log_softmax_rows.argtypes = [vp, ci, i64, i64, i32, vp, vp]
    lib.lmufft_log_softmax_rows_bwd.restype = ci
    lib.lmufft_log_softmax_rows_bwd.argtypes = [vp, vp, i64, i32, vp, ci, i64, vp]
    lib.lmufft_ctc_pack_targets.restype = ci
    lib.lmufft_ctc_pack_targets.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, vp, vp, vp]
    lib.lmufft_grad_norm_workspace_bytes.restype = sz
    lib.lmufft_grad_norm_workspace_bytes.argtypes = []
    lib.lmufft_grad_norm2_f32.restype = ci
    lib.lmufft_grad_norm2_f32.argtypes = [vp, ctypes.c_int64, vp, vp, sz, vp]
    lib.lmufft_adamw_step_f32.restype = ci
    lib.lmufft_adamw_step_f32.argtypes = [vp, vp, vp, vp, ctypes.c_int64] + [ctypes.c_float] * 5 + [
        ctypes.c_int32, vp, ctypes.c_float, vp]
    lib.lmufft_adamw_step_clock_f32.restype = ci
    lib.lmufft_adamw_step_clock_f32.argtypes = [vp, vp, vp, vp, ctypes.c_int64, vp, ctypes.POINTER(LrSchedule)] + [
        ctypes.c_float] * 4 + [vp, ctypes.c_float, vp]
    lib.lmufft_profile_enable.restype = ci
    lib.lmufft_profile_enable.argtypes = [ci]
    lib.lmufft_profile_reset.restype = None
    lib.lmufft_profile_reset.argtypes = []
    lib.lmufft_profile_count.restype = ci
    lib.lmufft_profile_count.argtypes = []
    lib.lmufft_profile_get.restype = ci
    lib.lmufft_profile_get.argtypes = [ci, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_int64)]
    got = lib.lmufft_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, expected {ABI_VERSION}; rebuild it")
    return lib


lib = _load()

# every symbol include/lmufft.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "lmufft_abi_version",
    "lmufft_last_error",
    "lmufft_forward_workspace_bytes",
    "lmufft_forward_f32",
    "lmufft_backward_workspace_bytes",
    "lmufft_backward_f32",
    "lmufft_conv_workspace_bytes",
    "lmufft_conv_forward_f32",
    "lmufft_conv_backward_f32",
    "lmufft_conv_backward_noalloc_f32",
    "lmufft_fftprep_bytes",
    "lmufft_fftprep_f32",
    "lmufft_melprep_bytes",
    "lmufft_melprep",
    "lmufft_mel_frames",
    "lmufft_mel_forward_f32",
    "lmufft_attention_fwd",
    "lmufft_attention_fwd_ld",
    "lmufft_attention_fwd_train",
    "lmufft_attention_bwd",
    "lmufft_add_layernorm",
    "lmufft_layernorm_bwd_workspace_bytes",
    "lmufft_layernorm_bwd",
    "lmufft_add_layernorm_dropout",
    "lmufft_layernorm_dropout_bwd",
    "lmufft_layernorm_dropout_bwd_relu",
    "lmufft_layernorm_param_sum_many",
    "lmufft_dropout_epoch",
    "lmufft_linear_bf16",
    "lmufft_linear_ln_bf16",
    "lmufft_linear_wprep_bf16",
    "lmufft_linear_wprep_many_bf16",
    "lmufft_linear_wgrad_workspace_bytes",
    "lmufft_linear_wgrad_bf16",
    "lmufft_linear_wgrad_slabs_bf16",
    "lmufft_linear_wgrad_sum_bf16",
    "lmufft_linear_wgrad_sum_many_bf16",
    "lmufft_ctc_workspace_bytes",
    "lmufft_ctc_greedy",
    "lmufft_ctc_pack_targets",
    "lmufft_ctc_loss_workspace_bytes",
    "lmufft_ctc_loss",
    "lmufft_ctc_mean_loss",
    "lmufft_log_softmax_rows",
    "lmufft_log_softmax_rows_bwd",
    "lmufft_grad_norm_workspace_bytes",
    "lmufft_grad_norm2_f32",
    "lmufft_adamw_step_f32",
    "lmufft_adamw_step_clock_f32",
    "lmufft_hprep_bytes",
    "lmufft_hprep_bf16",
    "lmufft_wprep_bytes",
    "lmufft_wprep_bf16",
    "lmufft_forward_bf16",
    "lmufft_forward_ln_bf16",
    "lmufft_backward_needs_gm_bf16",
    "lmufft_backward_bf16",
    "lmufft_backward_data_bf16",
    "lmufft_backward_weights_bf16",
    "lmufft_backward_partial_bf16",
    "lmufft_backward_flags_bf16",
    "lmufft_qt_bytes",
    "lmufft_qprep_many_bf16",
    "lmufft_backward_weights_sum_many_bf16",
    "lmufft_conv_forward_bf16",
    "lmufft_conv_backward_bf16",
    "lmufft_scan_chunk",
    "lmufft_scanprep_bytes",
    "lmufft_scanprep_build_host",
    "lmufft_conv_scan_workspace_bytes",
    "lmufft_conv_forward_scan_bf16",
    "lmufft_conv_backward_scan_bf16",
    "lmufft_profile_enable",
    "lmufft_profile_reset",
    "lmufft_profile_count",
    "lmufft_profile_get",
)


def profile_enable(on: bool) -> None:
    lib.lmufft_profile_enable(1 if on else 0)


def profile_reset() -> None:
    lib.lmufft_profile_reset()


def profile_read() -> dict:
    """{kernel name: (total_ms, launches)} accumulated since the last reset."""
    out = {}
    for i in range(lib.lmufft_profile_count()):
        name = ctypes.create_string_buffer(128)
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        check(lib.lmufft_profile_get(i, name, 128, ctypes.byref(ms), ctypes.byref(n)), "lmufft_profile_get")
        out[name.value.decode()] = (ms.value, n.value)
    return out


def check(rc: int, what: str) -> None:
    """Turn a non-zero return code into the RuntimeError the reference's users expect."""
    if rc != 0:
        msg = lib.lmufft_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg}")
