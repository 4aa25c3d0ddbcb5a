"""ctypes binding of libmulan_hip.so (the C ABI declared in include/mulan_hip.h).

There is no CPU fallback: if the shared object is missing or a symbol cannot be resolved this
module raises, and every call checks the returned hipError_t.  torch is used only for device
memory and the current HIP stream.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmulan_hip.so")
if os.environ.get("MULAN_HIP_LIB"):        # dev A/B builds (python -m mulan_amd.build --variant b -D...): another build of THIS library
    LIB_PATH = os.environ["MULAN_HIP_LIB"]

P, I, F, Z, U, LL, D = c_void_p, c_int, c_float, c_size_t, c_ulonglong, c_longlong, c_double


class SlabReduction(ctypes.Structure):
    """mulan_slab_reduction of include/mulan_hip.h: one pending slab reduction {slab, out, S, E, accumulate}"""
    _fields_ = [("slab", c_void_p), ("out", c_void_p), ("S", c_int), ("E", c_int), ("accumulate", c_int), ("reserved", c_int)]


# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/mulan_hip.h
SIGNATURES = {
    "mulan_conv3x3_fwd": [P, P, P, P, I, P, P, I, I, I, I, I, P],
    "mulan_conv3x3_wflip": [P, P, I, I, P],
    "mulan_conv3x3_wgrad_workspace": [I, I, I, I, I],
    "mulan_conv3x3_wgrad": [P, P, P, P, I, I, I, I, I, I, P],
    "mulan_conv3x3_pack_bf16x6_bytes": [I, I],
    "mulan_conv3x3_pack_bf16x6": [P, P, I, I, I, P],
    "mulan_conv3x3_fwd_bf16x6": [P, P, P, P, I, P, P, I, I, I, I, I, P],
    "mulan_conv3x3_wgrad_bf16x6_workspace": [I, I, I, I, I],
    "mulan_conv3x3_wgrad_bf16x6": [P, P, P, P, I, I, I, I, I, I, P],
    "mulan_absmax_rows": [P, P, I, Z, P],
    "mulan_add_absmax_rows": [P, P, P, P, I, Z, P],
    "mulan_add_absmax_rows_colsum": [P, P, P, P, P, I, Z, I, P],
    "mulan_conv3x3_pack_f16x3_bytes": [I, I],
    "mulan_conv3x3_pack_f16x3": [P, P, P, I, I, I, P],
    "mulan_conv3x3_fwd_f16x3": [P, P, P, P, P, P, I, P, P, P, P, I, I, I, I, I, P],
    "mulan_conv3x3_fwd_f16x3_alone": [P, P, P, P, P, P, I, P, P, P, P, I, I, I, I, I, I, P],
    "mulan_conv3x3_planes_bytes": [I, I, I, I],
    "mulan_conv3x3_wgrad_f16x3_planes_workspace": [I, I, I, I, I, I],
    "mulan_conv3x3_wgrad_f16x3_planes": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "mulan_conv3x3_wgrad_f16x3_planes_splits": [I, I, I, I, I, I],
    "mulan_conv3x3_wgrad_f16x3_planes_fold": [P, P, P, P, P, I, I, I, I, I, I, P, I, P],
    "mulan_slab_reduce": [P, P, I, I, I, P],
    "mulan_conv3x3_wgrad_f16x3_workspace": [I, I, I, I, I],
    "mulan_conv3x3_wgrad_f16x3": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "mulan_param_maxima": [P, P, I, P, P],
    "mulan_param_pack_f16x3": [P, P, I, P, P, P],
    "mulan_linear_pack_f16x3_bytes": [I, I],
    "mulan_linear_pack_f16x3": [P, P, P, I, I, I, P],
    "mulan_linear_f16x3": [P, P, P, P, I, I, P, P, P, P, P, P, P, I, I, I, I, P],
    "mulan_linear_wgrad_f16x3_planes_workspace": [I, I, I, I, I, I],
    "mulan_linear_wgrad_f16x3_planes": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "mulan_linear_wgrad_f16x3_x32_workspace": [I, I, I, I, I, I],
    "mulan_linear_wgrad_f16x3_x32": [P, P, I, I, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "mulan_gemm": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, LL, LL, LL, LL, F, F, P, P],
    "mulan_gemm_workspace": [I, I, I, I],
    "mulan_groupnorm_fwd": [P, P, I, I, P, P, P, P, P, I, I, I, F, I, F, U, U, P, P],
    "mulan_groupnorm_bwd": [P, P, P, I, I, P, P, P, P, P, P, P, P, I, I, I, I, F, U, U, I, P, P, P, P, P, P],
    "mulan_groupnorm_fwd_dyn": [P, P, I, I, P, P, P, P, P, I, I, I, F, I, F, U, U, P, P, P],
    "mulan_groupnorm_fwd_planes": [P, P, I, I, P, P, P, P, P, I, I, I, F, I, F, U, U, P, P, P],
    "mulan_conv3x3_fwd_f16x3_planes_in": [P, P, P, P, P, P, I, P, P, P, I, I, I, I, I, P],
    "mulan_conv3x3_fwd_f16x3_planes_in_stats": [P, P, P, P, P, P, I, P, P, P, P, I, I, I, I, I, I, P],
    "mulan_groupnorm_stats": [P, P, I, I, P, P, P, P, P, I, I, I, F, P],
    "mulan_conv3x3_fwd_f16x3_gn_in": [P, P, I, I, P, P, P, P, I, I, F, P, P, P, I, P, P, P, P, I, P, P, P, P, P, I, I, I, I, P],
    "mulan_conv3x3_f16x3_tile_rows": [I, I, I, I],
    "mulan_groupnorm_bwd_dyn": [P, P, P, I, I, P, P, P, P, P, P, P, P, I, I, I, I, F, U, U, P, I, P, P, P, P, P, P],
    "mulan_groupnorm_bwd_fused": [P, P, P, I, I, P, P, P, P, P, P, P, P, I, I, I, I, F, U, U, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "mulan_groupnorm_bwd_fused_planes": [P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, F, U, U, P, P, P, P, P, P, P, P, P, P],
    "mulan_groupnorm_fwd_planes_keepbits": [P, P, I, I, P, P, P, P, P, I, I, I, F, I, F, U, U, P, P, P, P],
    "mulan_groupnorm_fwd_stream": [P, P, I, I, P, P, P, P, P, P, P, P, I, I, I, I, F, I, F, U, U, P, P, P, P],
    "mulan_groupnorm_bwd_stream": [P, P, P, P, I, I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, U, U, P, P, P, P, P, P, P,
                                   P, P, P, P, P, P, P],
    "mulan_act_fwd": [P, P, Z, I, F, P],
    "mulan_act_bwd": [P, P, P, Z, I, P],
    "mulan_colsum": [P, P, I, I, I, I, I, P],
    "mulan_colsum_pair": [P, P, P, I, I, P],
    "mulan_softmax_fwd": [P, P, Z, I, P],
    "mulan_softmax_bwd": [P, P, P, Z, I, P],
    "mulan_softmax_scaled_fwd": [P, P, Z, I, F, P],
    "mulan_softmax_scaled_bwd": [P, P, P, Z, I, F, P, P],
    "mulan_linear_pack_f16x3_batched": [P, P, P, I, I, I, I, P],
    "mulan_linear_f16x3_batched": [P, P, I, P, P, P, P, P, I, I, I, P],
    "mulan_bmm_tn_f16x3_planes": [P, P, P, P, P, I, I, I, I, I, P],
    "mulan_attention_fwd_f16x3": [P, P, P, P, P, P, P, P, I, I, I, F, P],
    "mulan_attention_delta": [P, P, P, I, I, I, P],
    "mulan_attention_pack_f16x3": [P, P, P, P, I, I, I, P],
    "mulan_attention_bwd_f16x3": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P],
    "mulan_fourier_fwd": [P, P, Z, P],
    "mulan_fourier_bwd": [P, P, P, Z, I, P],
    "mulan_temb_fwd": [P, P, I, I, I, I, P],
    "mulan_temb_bwd": [P, P, P, I, I, I, I, P],
    "mulan_rowbcast": [P, P, Z, I, I, I, I, P],
    "mulan_axpby": [P, P, Z, F, F, P],
    "mulan_encode_u8": [P, P, Z, P],
    "mulan_poly_gamma_fwd": [P, P, P, P, P, P, P, P, I, I, F, F, P],
    "mulan_poly_gamma_bwd": [P, P, P, P, P, P, P, P, P, I, I, F, F, P],
    "mulan_expm1_weight_fwd": [P, P, P, Z, F, P],
    "mulan_expm1_weight_bwd": [P, P, P, P, P, Z, F, P],
    "mulan_qsample_fwd": [P, P, P, P, I, P, P, P, P, P, P, P, P, I, I, P],
    "mulan_qsample_bwd": [P, P, P, P, I, P, P, P, P, P, P, P, P, P, I, I, P],
    "mulan_diffloss_fwd": [I, P, P, P, I, P, P, P, P, I, I, P],
    "mulan_diffloss_bwd": [I, P, P, P, I, P, P, P, P, P, P, P, P, I, I, P],
    "mulan_topk_fwd": [P, P, P, P, P, P, I, I, I, F, P],
    "mulan_topk_bwd": [P, P, P, P, P, P, I, I, P],
    "mulan_ancestral_step": [P, P, P, P, P, P, Z, I, I, P],
    "mulan_decode_argmax": [P, P, P, Z, I, P],
    "mulan_decode_sample": [P, P, P, Z, I, U, U, P],
    "mulan_decode_logprobs": [P, P, P, Z, I, P],
    "mulan_rowmean": [P, P, I, I, P],
    "mulan_ode_drift": [P, P, P, P, P, P, P, Z, I, I, P],
    "mulan_ode_div": [P, P, P, P, P, I, I, I, I, P],
    "mulan_rk_combine": [P, P, Z, P, I, D, P, P, Z, P],
    "mulan_rk_workspace_bytes": [],
    "mulan_rk_error_norm": [P, P, P, Z, P, D, D, D, P, P, Z, P],
    "mulan_rk_init_norms": [P, P, P, D, D, P, P, Z, P],
    "mulan_normal_logp": [P, P, I, I, P],
    "mulan_noise": [P, Z, U, U, I, F, F, P],
    "mulan_dequantize": [P, P, P, P, Z, I, F, P],
    "mulan_adamw_ema_step": [P, P, P, P, P, Z, Z, F, F, F, F, F, I, F, F, P],
    "mulan_adamw_ema_step_scaled": [P, P, P, P, P, Z, Z, F, F, F, F, F, I, F, F, P, P],
    "mulan_adamw_ema_step_dyn": [P, P, P, P, P, Z, Z, F, F, F, F, F, F, P, P, P],
    "mulan_global_norm_clip_workspace": [],
    "mulan_global_norm_clip": [P, Z, F, F, P, P, P],
    "mulan_randn": [P, Z, U, U, P],
    "mulan_version": [],
    "mulan_set_tuning": [I, I],
    "mulan_set_debug_buffer": [P],
    "mulan_event_create": [P],
    "mulan_event_destroy": [P],
    "mulan_event_record_external": [P, P, P],
    "mulan_stream_wait_event": [P, P],
    "mulan_signal_create": [P],
    "mulan_signal_destroy": [P],
    "mulan_signal_set": [P, P, P],
    "mulan_stream_wait_signal": [P, P, ctypes.c_uint],
    "mulan_signal_read": [P, P],
}
_RESTYPES = {"mulan_rk_workspace_bytes": c_size_t, "mulan_global_norm_clip_workspace": c_size_t, "mulan_conv3x3_wgrad_workspace": c_size_t, "mulan_conv3x3_pack_bf16x6_bytes": c_size_t, "mulan_conv3x3_wgrad_bf16x6_workspace": c_size_t,
             "mulan_conv3x3_pack_f16x3_bytes": c_size_t, "mulan_conv3x3_planes_bytes": c_size_t, "mulan_linear_pack_f16x3_bytes": c_size_t,
             "mulan_linear_wgrad_f16x3_planes_workspace": c_size_t, "mulan_linear_wgrad_f16x3_x32_workspace": c_size_t,
             "mulan_conv3x3_wgrad_f16x3_planes_workspace": c_size_t, "mulan_conv3x3_wgrad_f16x3_workspace": c_size_t, "mulan_gemm_workspace": c_size_t, "mulan_version": c_char_p}


class MulanHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the library (after torch, so the HIP runtime already in the process is reused)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (libamdhip64.so.7 must be resident first)
    if not os.path.exists(LIB_PATH):
        raise MulanHipError(
            f"{LIB_PATH} not found: build it with `python -m mulan_amd.build` "
            "(there is no CPU fallback for the MuLAN hot path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    for kv in filter(None, os.environ.get("MULAN_TUNE", "").split(",")):   # developer knobs, e.g. MULAN_TUNE=8=1
        k, v = kv.split("=")
        lib.mulan_set_tuning(int(k), int(v))
    return lib


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise MulanHipError("non-contiguous tensor passed to the HIP boundary")
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def check(rc, name):
    if rc != 0:
        raise MulanHipError(f"{name} failed with hipError_t {rc}")


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MulanHipError(f"{name} failed with hipError_t {rc}")


def version():
    return load().mulan_version().decode()
