"""`torch.ops.sgl_kernel.*` for the hot path, with the reference's op schemas.

The reference's kernels reach Python as torch ops registered by sgl-kernel/csrc/torch_extension.cc:49-175 (CUDA) and
torch_extension_rocm.cc:25-55 (the ROCm all-reduce set); its wrappers in sgl-kernel/python/sgl_kernel/*.py and a few
call sites in python/sglang call `torch.ops.sgl_kernel.<op>(...)` directly.  `register()` defines the same names with the
same schema strings in a torch.library and implements them on the HIP ("CUDA") dispatch key through the C-ABI of
include/semipd.h -- a process that imports this module instead of the sgl_kernel extension finds every op of the path
SURVEY 8(a) lists under its reference name and argument order.  The `cuda_stream` / `cublas_handle` integers of the
schemas are honoured (the launch goes to that stream) / ignored (no cuBLAS here).

Registered: rmsnorm, fused_add_rmsnorm, silu_and_mul, apply_rope_pos_ids_cos_sin_cache, moe_align_block_size, bmm_fp8, min_p_sampling_from_probs, top_k_renorm_probs_wrapper, top_p_renorm_probs,
top_k_top_p_sampling_from_probs, top_p_sampling_from_probs, and the ten ROCm custom all-reduce ops
(semi_pd_amd/sgl_kernel_allreduce.py).  Everything else in torch_extension.cc (speculative decoding trees, int8 / AWQ,
cutlass MoE, lightning attention ...) is outside the path.
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch

from semi_pd_amd import _lib, ops
from semi_pd_amd import sgl_kernel_allreduce as _ar

_LIB: Optional[torch.library.Library] = None

_SCHEMAS = [
    # torch_extension.cc:49-73
    "rmsnorm(Tensor! output, Tensor input, Tensor weight, float eps, int cuda_stream) -> ()",
    "fused_add_rmsnorm(Tensor! input, Tensor! residual, Tensor weight, float eps) -> ()",
    "silu_and_mul(Tensor! out, Tensor input, int cuda_stream) -> ()",
    "apply_rope_pos_ids_cos_sin_cache(Tensor q, Tensor k, Tensor! q_rope, Tensor! k_rope, Tensor cos_sin_cache, "
    "Tensor pos_ids, bool interleave, int cuda_stream) -> ()",
    # torch_extension.cc:115-118
    "moe_align_block_size(Tensor topk_ids, int num_experts, int block_size, Tensor! sorted_token_ids, Tensor! "
    "experts_ids, Tensor! num_tokens_post_pad, Tensor! token_cnts_buffer, Tensor! cumsum_buffer) -> ()",
    # torch_extension.cc:146-170
    "bmm_fp8(Tensor A, Tensor B, Tensor! D, Tensor A_scale, Tensor B_scale, Tensor workspace_buffer, int "
    "cublas_handle, int cuda_stream) -> ()",
    "min_p_sampling_from_probs(Tensor probs, Tensor uniform_samples, Tensor! samples, Tensor? maybe_min_p_arr, float "
    "min_p_val, bool deterministic, int cuda_stream) -> ()",
    "top_k_renorm_probs_wrapper(Tensor probs, Tensor! renorm_probs, Tensor? maybe_top_k_arr, int top_k_val, int "
    "cuda_stream) -> ()",
    "top_p_renorm_probs(Tensor probs, Tensor! renorm_probs, Tensor? maybe_top_p_arr, float top_p_val, int "
    "cuda_stream) -> ()",
    "top_k_top_p_sampling_from_probs(Tensor probs, Tensor uniform_samples, Tensor! samples, Tensor! success, Tensor? "
    "maybe_top_k_arr, float top_k_val, Tensor? maybe_top_p_arr, float top_p_val, bool deterministic, int "
    "cuda_stream) -> ()",
    "top_p_sampling_from_probs(Tensor probs, Tensor uniform_samples, Tensor! samples, Tensor! success, Tensor? "
    "maybe_top_p_arr, float top_p_val, bool deterministic, int cuda_stream) -> ()",
    # torch_extension_rocm.cc:25-55
    "init_custom_ar(Tensor meta, Tensor rank_data, str[] handles, int[] offsets, int rank, bool full_nvlink) -> int",
    "all_reduce_reg(int fa, Tensor inp, Tensor! out) -> ()",
    "all_reduce_unreg(int fa, Tensor inp, Tensor reg_buffer, Tensor! out) -> ()",
    "dispose(int fa) -> ()",
    "meta_size() -> int",
    "register_buffer(int fa, Tensor t, str[] handles, int[] offsets) -> ()",
    "get_graph_buffer_ipc_meta(int fa) -> (Tensor, int[])",
    "register_graph_buffers(int fa, str[] handles, int[][] offsets) -> ()",
    "allocate_meta_buffer(int size) -> Tensor",
    "get_meta_buffer_ipc_handle(Tensor inp) -> Tensor",
]


@contextlib.contextmanager
def _on(cuda_stream: int, device):
    """Launch on the hipStream_t the caller passed (0 = torch's current stream, as get_cuda_stream() returns it anyway)."""
    if not cuda_stream or cuda_stream == torch.cuda.current_stream(device).cuda_stream:
        yield
    else:
        with torch.cuda.stream(torch.cuda.ExternalStream(cuda_stream, device=device)):
            yield


# --------------------------------------------------------------------------- elementwise
def _rmsnorm(output, input, weight, eps, cuda_stream):
    with _on(cuda_stream, input.device):
        ops.rmsnorm(input, weight, eps, out=output)


def _fused_add_rmsnorm(input, residual, weight, eps):
    ops.fused_add_rmsnorm(input, residual, weight, eps)


def _silu_and_mul(out, input, cuda_stream):
    with _on(cuda_stream, input.device):
        ops.silu_and_mul(input, out=out)


def _apply_rope(q, k, q_rope, k_rope, cos_sin_cache, pos_ids, interleave, cuda_stream):
    """q / k [nnz, heads, head_size] views (elementwise.py:142-151 passes q and q_rope as the same view: in place)."""
    with _on(cuda_stream, q.device):
        if q_rope.data_ptr() != q.data_ptr():
            q_rope.copy_(q)
        if k_rope.data_ptr() != k.data_ptr():
            k_rope.copy_(k)
        nnz, head = q_rope.shape[0], q_rope.shape[-1]
        ops.apply_rope_with_cos_sin_cache_inplace(pos_ids, q_rope.view(nnz, -1), k_rope.view(nnz, -1), head, cos_sin_cache,
                                                  is_neox=not interleave)


# --------------------------------------------------------------------------- MoE
def _moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad,
                          token_cnts_buffer, cumsum_buffer):
    ops.moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad,
                             token_cnts_buffer, cumsum_buffer)


# --------------------------------------------------------------------------- fp8
def _bmm_fp8(A, B, D, A_scale, B_scale, workspace_buffer, cublas_handle, cuda_stream):
    with _on(cuda_stream, A.device):
        ops.bmm_fp8(A, B, A_scale, B_scale, D.dtype, out=D)


# --------------------------------------------------------------------------- sampling
def _opt(t):
    return _lib.ptr(t) if t is not None else None


def _stream(cuda_stream, device):
    return cuda_stream or _lib.current_stream(device)


def _min_p(probs, uniform_samples, samples, maybe_min_p_arr, min_p_val, deterministic, cuda_stream):
    B, V = probs.shape
    u = uniform_samples[0] if uniform_samples.dim() == 2 else uniform_samples
    _lib.check(_lib.load().semipd_min_p_sampling_from_probs(_lib.ptr(probs), _lib.ptr(u), _opt(maybe_min_p_arr),
                                                            float(min_p_val), _lib.ptr(samples), B, V,
                                                            _stream(cuda_stream, probs.device)), "min_p_sampling_from_probs")


def _top_k_renorm(probs, renorm_probs, maybe_top_k_arr, top_k_val, cuda_stream):
    B, V = probs.shape
    _lib.check(_lib.load().semipd_top_k_renorm_prob(_lib.ptr(probs), _lib.ptr(renorm_probs), _opt(maybe_top_k_arr),
                                                    int(top_k_val), B, V, _stream(cuda_stream, probs.device)),
               "top_k_renorm_probs_wrapper")


def _top_p_renorm(probs, renorm_probs, maybe_top_p_arr, top_p_val, cuda_stream):
    B, V = probs.shape
    _lib.check(_lib.load().semipd_top_p_renorm_prob(_lib.ptr(probs), _lib.ptr(renorm_probs), _opt(maybe_top_p_arr),
                                                    float(top_p_val), B, V, _stream(cuda_stream, probs.device)),
               "top_p_renorm_probs")


def _top_k_top_p(probs, uniform_samples, samples, success, maybe_top_k_arr, top_k_val, maybe_top_p_arr, top_p_val,
                 deterministic, cuda_stream):
    B, V = probs.shape
    _lib.check(_lib.load().semipd_top_k_top_p_sampling_from_probs(
        _lib.ptr(probs), _lib.ptr(uniform_samples), _opt(maybe_top_k_arr), int(top_k_val), _opt(maybe_top_p_arr),
        float(top_p_val), _lib.ptr(samples), _lib.ptr(success), B, V, uniform_samples.shape[0],
        _stream(cuda_stream, probs.device)), "top_k_top_p_sampling_from_probs")


def _top_p(probs, uniform_samples, samples, success, maybe_top_p_arr, top_p_val, deterministic, cuda_stream):
    """top-p alone = the joint sampler with top_k = vocabulary size (sampling.py:100-136)."""
    _top_k_top_p(probs, uniform_samples, samples, success, None, probs.shape[1], maybe_top_p_arr, top_p_val, deterministic,
                 cuda_stream)


_IMPLS = {
    "rmsnorm": _rmsnorm, "fused_add_rmsnorm": _fused_add_rmsnorm, "silu_and_mul": _silu_and_mul,
    "apply_rope_pos_ids_cos_sin_cache": _apply_rope, "moe_align_block_size": _moe_align_block_size,
    "bmm_fp8": _bmm_fp8, "min_p_sampling_from_probs": _min_p,
    "top_k_renorm_probs_wrapper": _top_k_renorm, "top_p_renorm_probs": _top_p_renorm,
    "top_k_top_p_sampling_from_probs": _top_k_top_p, "top_p_sampling_from_probs": _top_p,
    "all_reduce_reg": _ar.all_reduce_reg, "all_reduce_unreg": _ar.all_reduce_unreg,
    "register_buffer": lambda fa, t, handles, offsets: _ar.register_buffer(fa, t, [h.encode("latin-1") for h in handles], offsets),
    "get_meta_buffer_ipc_handle": _ar.get_meta_buffer_ipc_handle,
}
# ops without a tensor argument (or with CPU-only ones) have no dispatch key to hang on: registered for every backend
_IMPLS_ANY = {
    "init_custom_ar": lambda meta, rank_data, handles, offsets, rank, full_nvlink: _ar.init_custom_ar(
        meta, rank_data, [h.encode("latin-1") for h in handles], offsets, rank, full_nvlink),
    "dispose": _ar.dispose, "meta_size": _ar.meta_size,
    "get_graph_buffer_ipc_meta": _ar.get_graph_buffer_ipc_meta,
    "register_graph_buffers": lambda fa, handles, offsets: _ar.register_graph_buffers(
        fa, [h.encode("latin-1") for h in handles], offsets),
    "allocate_meta_buffer": _ar.allocate_meta_buffer,
}


def register() -> List[str]:
    """Defines and implements the ops once per process; returns their names.  Raises if another library has already
    defined `sgl_kernel::<op>` (the real extension is loaded: nothing to stand in for)."""
    global _LIB
    names = [s.split("(", 1)[0] for s in _SCHEMAS]
    if _LIB is not None:
        return names
    lib = torch.library.Library("sgl_kernel", "FRAGMENT")
    for schema in _SCHEMAS:
        lib.define(schema)
    for name, fn in _IMPLS.items():
        lib.impl(name, fn, "CUDA")
    # the 64-byte IPC handle tensor lives on the CPU in the reference too (torch_extension_rocm.cc:54)
    lib.impl("get_meta_buffer_ipc_handle", _ar.get_meta_buffer_ipc_handle, "CPU")
    for name, fn in _IMPLS_ANY.items():
        lib.impl(name, fn, "CompositeExplicitAutograd")
    _LIB = lib
    return names
