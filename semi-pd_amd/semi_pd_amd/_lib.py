"""ctypes binding of libsemipd_hip.so (the C-ABI declared in include/semipd.h).

This is the only place that loads native code.  There is no CPU or PyTorch fallback:
if the shared library is missing, or a call fails, a RuntimeError is raised
(the reference raises through TORCH_CHECK the same way,
sgl-kernel/csrc/elementwise/fused_add_rms_norm_kernel.cu:51-52).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "SEMIPD_HIP_LIB", os.path.join(os.path.dirname(_PKG_DIR), "lib", "libsemipd_hip.so")
)

F32, F16, BF16, F8E5M2, F8E4M3 = 0, 1, 2, 3, 4
_DTYPE_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
_KV_DTYPE_CODE = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32,
                  torch.float8_e5m2: F8E5M2, torch.float8_e4m3fn: F8E4M3}

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None

_i64, _i32, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "semipd_version": [],
    "semipd_last_error": [],
    "semipd_rmsnorm": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _i32, _vp],
    "semipd_fused_add_rmsnorm": [_vp, _vp, _vp, _i64, _i64, _f32, _i32, _vp],
    "semipd_silu_and_mul": [_vp, _vp, _i64, _i64, _i32, _vp],
    "semipd_rope_inplace": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _vp],
    "semipd_rope_inplace_strided": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_rope_kv_store": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32,
                             _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp],
    "semipd_rope_kv_store_planes": [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _i64,
                                    _i32, _i32, _vp],
    "semipd_kv_store": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "semipd_kv_store_cvt": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_build_kv_indices": [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp],
    "semipd_compute_positions": [_vp, _vp, _vp, _vp, _i64, _vp],
    "semipd_decode_attention": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i64,
                                _i64, _i64, _i64, _i32, _f32, _f32, _i32, _i32, _vp],
    "semipd_decode_rope_attention_planes_supported": [_i32, _i32, _i32, _i32, _i32],
    "semipd_decode_rope_attention_planes": [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32,
                                            _i64, _i64, _i64, _i32, _i32, _f32, _f32, _i32, _i32, _vp],
    "semipd_extend_attention": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                                _i64, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _f32, _i32, _i32, _vp],
    "semipd_extend_attention_masked": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32,
                                       _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _f32, _i32, _i32, _vp],
    "semipd_gather_rows": [_vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "semipd_argmax": [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_lm_head_argmax": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_lm_head_argmax_workspace": [_i64, _i64],
    "semipd_linear_workspace": [_i64, _i64],
    "semipd_linear": [_vp, _vp, _vp, _vp, _sz, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_input_to_float8": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_bmm_fp8": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32,
                       _i32, _vp],
    "semipd_runtime_version": [_vp, _vp],
    "semipd_stream_create": [_i32, _vp],
    "semipd_stream_abort_capture": [_vp],
    "semipd_clear_last_error": [],
    "semipd_stream_linear_workspace": [_i64],
    "semipd_stream_linear_set_cus": [_i32],
    "semipd_stream_linear_f32": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "semipd_gemm_tall_set_cus": [_i32],
    "semipd_gemm_tall_set_form": [_i32],
    "semipd_gemm_tall": [_vp, _vp, _vp, _vp, _sz, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_gemm_tall_planes": [_vp, _vp, _sz, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _vp],
    "semipd_moe_stream_gemm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    "semipd_moe_sum_scale_add": [_vp, _vp, _vp, _i64, _i32, _i64, _f32, _i32, _i32, _vp],
    "semipd_moe_gemm_tall": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    "semipd_dense_gemm_init": [_sz],
    "semipd_dense_gemm_set_cus": [_i32],
    "semipd_dense_gemm_import": [C.c_char_p, _vp],
    "semipd_dense_gemm_library_version": [_vp],
    "semipd_dense_gemm_tune": [_i64, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "semipd_dense_gemm": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "semipd_dense_gemm_report": [_vp, _sz],
    "semipd_stream_linear_planes": [_vp, _sz, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp],
    "semipd_fused_add_rmsnorm_planes": [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _f32, _i32, _vp],
    "semipd_stream_linear": [_vp, _vp, _vp, _vp, _sz, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_softmax_temperature": [_vp, _vp, _i64, _i64, _vp],
    "semipd_top_k_top_p_sampling_from_probs": [_vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp, _i64, _i64, _i32, _vp],
    "semipd_min_p_sampling_from_probs": [_vp, _vp, _vp, _f32, _vp, _i64, _i64, _vp],
    "semipd_top_k_renorm_prob": [_vp, _vp, _vp, _i32, _i64, _i64, _vp],
    "semipd_top_p_renorm_prob": [_vp, _vp, _vp, _f32, _i64, _i64, _vp],
    "semipd_token_logprobs": [_vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "semipd_topk_softmax": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "semipd_grouped_topk": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "semipd_moe_align_block_size": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp],
    "semipd_moe_grouped_gemm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32,
                                _i32, _i32, _vp],
    "semipd_moe_grouped_gemm_silu_supported": [_i64, _i64, _i64, _i32, _i32, _i32],
    "semipd_moe_grouped_gemm_silu": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp],
    "semipd_moe_sum": [_vp, _vp, _i64, _i32, _i64, _i32, _vp],
    "semipd_per_token_group_quant_fp8": [_vp, _vp, _vp, _i64, _i64, _i32, C.c_float, _i32, _vp],
    "semipd_fused_add_rmsnorm_quant_fp8": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, C.c_float, _i32, C.c_float, _i32, _vp],
    "semipd_silu_and_mul_quant_fp8": [_vp, _vp, _vp, _i64, _i64, _i32, C.c_float, _i32, _vp],
    "semipd_w8a8_block_fp8_matmul": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _sz, _vp],
    "semipd_moe_grouped_gemm_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32,
                                    _i32, _i32, _i32, _i32, _vp],
    "semipd_ipc_get_handle": [_vp, _vp, _vp],
    "semipd_ipc_open": [_vp, _i32, _vp],
    "semipd_ipc_close": [_vp],
    "semipd_ipc_num_open": [],
    "semipd_device_cu_count": [_i32, _vp],
    "semipd_cu_mask_fill": [_i32, _i32, _i32, _vp, _i32],
    "semipd_stream_create_cu_mask": [_i32, _vp, _i32, _vp],
    "semipd_stream_create_with_priority": [_i32, _i32, _vp, _vp],
    "semipd_mla_decode_prep_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _i32, _i32, _i64, _i64,
                                    _i64, _i64, _i64, _i64, _i32, _i32, _vp],
    "semipd_grouped_topk_planes": [_vp, _i32, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "semipd_moe_sum_scale_add_planes": [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _i64, _f32, _i32, _i32, _vp],
    "semipd_rmsnorm_quant_fp8": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _i32, _f32, _i32, _vp],
    "semipd_bmm_nk": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "semipd_mla_decode_prep": [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _i32, _i32, _i64,
                               _i64, _i64, _i32, _i32, _vp],
    "semipd_stream_destroy": [_vp],
    "semipd_stream_get_cu_mask": [_vp, _vp, _i32],
    "semipd_share_board_open": [C.c_char_p, _i32, _vp],
    "semipd_share_board_close": [_vp],
    "semipd_share_board_store": [_vp, _i32, _i64],
    "semipd_share_board_add": [_vp, _i32, _i64, _vp],
    "semipd_share_board_load": [_vp, _i32, _vp],
    "semipd_probe_cu_placement": [_vp, _i32, _i64, _vp],
    "semipd_launch_noop": [_i32, _vp],
    "semipd_ar_meta_size": [],
    "semipd_ar_region_size": [_sz],
    "semipd_ar_alloc_shared": [_sz, _vp],
    "semipd_ar_set_cu_trace": [_vp, _vp],
    "semipd_ar_free_shared": [_vp],
    "semipd_ar_init": [_vp, _sz, _i32, _i32, _vp],
    "semipd_ar_set_timeout_ms": [_vp, C.c_uint32],
    "semipd_ar_timed_out": [_vp, _vp],
    "semipd_ar_max_bytes": [_vp, _vp],
    "semipd_ar_all_reduce": [_vp, _vp, _vp, _sz, _i32, _vp],
    "semipd_ar_all_gather": [_vp, _vp, _vp, _sz, _vp],
    "semipd_ep_dispatch": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "semipd_ep_combine": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp],
    "semipd_ar_dispose": [_vp],
}
_RESTYPES = {"semipd_last_error": C.c_char_p, "semipd_lm_head_argmax_workspace": _sz,
             "semipd_linear_workspace": _sz, "semipd_stream_linear_workspace": _sz, "semipd_dense_gemm_report": _sz, "semipd_ar_meta_size": _sz, "semipd_ar_region_size": _sz}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load() -> C.CDLL:
    """Load (once) and return the shared library; fail loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libsemipd_hip.so not found at {LIB_PATH}: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C semi-pd_amd/csrc`). "
                "There is no CPU fallback for the Semi-PD hot path."
            )
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = header / library mismatch
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def last_error() -> str:
    msg = load().semipd_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {dtype} (float32 / float16 / bfloat16 only)") from None


def kv_dtype_code(dtype: torch.dtype) -> int:
    """Storage type of KV-pool rows: the activation type or one of the two OCP fp8 types."""
    try:
        return _KV_DTYPE_CODE[dtype]
    except KeyError:
        raise RuntimeError(f"unsupported KV-cache dtype {dtype}") from None


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: no host path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("semi_pd_amd ops need a tensor on a HIP device (got a CPU tensor)")
    return t.data_ptr()


def current_stream(device=None) -> int:
    """hipStream_t of torch's current stream (what get_cuda_stream() returns in
    sgl-kernel/python/sgl_kernel/utils.py)."""
    return torch.cuda.current_stream(device).cuda_stream
