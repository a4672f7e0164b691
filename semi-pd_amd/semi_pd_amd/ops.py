"""Host-side operator layer: the reference's op seam, backed by libsemipd_hip.so.

Function names, argument meaning and error behaviour mirror
  * sgl-kernel/python/sgl_kernel/elementwise.py:9-151 (rmsnorm, fused_add_rmsnorm,
    silu_and_mul, apply_rope_with_cos_sin_cache_inplace),
  * sgl-kernel/python/sgl_kernel/moe.py:4-23 (moe_align_block_size),
  * layers/attention/triton_ops/decode_attention.py:625-636 (decode_attention_fwd),
  * layers/attention/triton_ops/extend_attention.py:291-307 (extend_attention_fwd),
  * layers/attention/utils.py:5-39 (create_flashinfer_kv_indices),
so a parity test written against the reference reads the same here.  Every op
launches on torch's *current* stream and never synchronises.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, current_stream, dtype_code, ptr


def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1] if t.numel() else 0


def _need_contig_last(t: torch.Tensor, name: str) -> None:
    if t.stride(-1) != 1:
        raise RuntimeError(f"{name}: last dimension must be contiguous")


# --------------------------------------------------------------------------- norm
def rmsnorm(input: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if input.dim() != 2:
        raise RuntimeError("rmsnorm: input must be 2-D [tokens, hidden]")
    if weight.dim() != 1 or weight.shape[0] != input.shape[1]:
        raise RuntimeError("rmsnorm: weight shape mismatch")
    if weight.dtype != input.dtype:
        raise RuntimeError("rmsnorm: weight dtype must match input")
    _need_contig_last(input, "rmsnorm")
    if out is None:
        out = torch.empty_like(input)
    _need_contig_last(out, "rmsnorm")
    lib = _lib.load()
    check(lib.semipd_rmsnorm(ptr(out), ptr(input), ptr(weight), input.shape[0], input.shape[1],
                             input.stride(0), out.stride(0), eps, dtype_code(input.dtype),
                             current_stream(input.device)), "rmsnorm")
    return out


def fused_add_rmsnorm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                      eps: float = 1e-6) -> None:
    if input.dim() != 2 or residual.shape != input.shape:
        raise RuntimeError("fused_add_rmsnorm: input/residual must be 2-D and the same shape")
    if weight.dim() != 1 or weight.shape[0] != input.shape[1]:
        raise RuntimeError("fused_add_rmsnorm: weight shape mismatch")
    if not (input.is_contiguous() and residual.is_contiguous()):
        raise RuntimeError("fused_add_rmsnorm: tensors must be contiguous")
    if not (weight.dtype == input.dtype == residual.dtype):
        raise RuntimeError("fused_add_rmsnorm: dtype mismatch")
    lib = _lib.load()
    check(lib.semipd_fused_add_rmsnorm(ptr(input), ptr(residual), ptr(weight), input.shape[0],
                                       input.shape[1], eps, dtype_code(input.dtype),
                                       current_stream(input.device)), "fused_add_rmsnorm")


# --------------------------------------------------------------------------- activation
def silu_and_mul(input: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if input.shape[-1] * input.dtype.itemsize % 16 != 0:
        raise ValueError("The pointers must be multiple of 16 bytes.")
    d = input.shape[-1] // 2
    if out is not None:
        if out.shape[:-1] != input.shape[:-1] or out.shape[-1] * 2 != input.shape[-1]:
            raise AssertionError(f"{input.shape} vs {out.shape}")
    else:
        out = torch.empty(input.shape[:-1] + (d,), device=input.device, dtype=input.dtype)
    if not (input.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("silu_and_mul: tensors must be contiguous")
    lib = _lib.load()
    check(lib.semipd_silu_and_mul(ptr(out), ptr(input), _rows(input), d, dtype_code(input.dtype),
                                  current_stream(input.device)), "silu_and_mul")
    return out


# --------------------------------------------------------------------------- rope
def apply_rope_with_cos_sin_cache_inplace(positions: torch.Tensor, query: torch.Tensor,
                                          key: torch.Tensor, head_size: int,
                                          cos_sin_cache: torch.Tensor, is_neox: bool = True) -> None:
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    if positions.dtype != torch.int64:
        positions = positions.long()
    nnz = query.shape[0]
    q = query.view(nnz, -1, head_size)
    k = key.view(nnz, -1, head_size)
    if q.stride(-1) != 1 or k.stride(-1) != 1 or q.stride(1) != head_size or k.stride(1) != head_size:
        raise RuntimeError("rope: heads must be densely packed")
    lib = _lib.load()
    check(lib.semipd_rope_inplace(ptr(q), ptr(k), ptr(cos_sin_cache), ptr(positions), nnz, q.shape[1],
                                  k.shape[1], head_size, cos_sin_cache.shape[1], q.stride(0),
                                  k.stride(0), 0 if is_neox else 1, dtype_code(query.dtype),
                                  current_stream(query.device)), "apply_rope")


def apply_rope_strided_inplace(positions: torch.Tensor, q_rot: torch.Tensor, k_rot: torch.Tensor,
                               cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """In-place RoPE on views q_rot [T, Hq, rot] / k_rot [T, Hk, rot] whose heads are strided slices
    of larger rows (MLA: DeepseekScalingRotaryEmbedding.forward, rotary_embedding.py:710-748)."""
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    if positions.dtype != torch.int64:
        positions = positions.long()
    T, Hq, rot = q_rot.shape
    Hk = k_rot.shape[1]
    if q_rot.stride(2) != 1 or k_rot.stride(2) != 1 or rot != cos_sin_cache.shape[1] or k_rot.shape[2] != rot:
        raise RuntimeError("apply_rope_strided_inplace: bad layout")
    lib = _lib.load()
    check(lib.semipd_rope_inplace_strided(ptr(q_rot), ptr(k_rot), ptr(cos_sin_cache), ptr(positions), T, Hq, Hk,
                                          rot, q_rot.stride(0), q_rot.stride(1), k_rot.stride(0),
                                          k_rot.stride(1), 0 if is_neox else 1, dtype_code(q_rot.dtype),
                                          current_stream(q_rot.device)), "rope_strided")


def rope_and_store_kv(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                      value: torch.Tensor, head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool,
                      k_buffer: torch.Tensor, v_buffer: torch.Tensor, loc: torch.Tensor) -> None:
    """RoPE on q/k in place fused with set_kv_buffer(loc, k, v)
    (layers/rotary_embedding.py:143-169 + mem_cache/memory_pool.py:316-346)."""
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    nnz = query.shape[0]
    q = query.view(nnz, -1, head_size)
    k = key.view(nnz, -1, head_size)
    v = value.view(nnz, k.shape[1], -1)
    v_head = v.shape[2]
    if loc.dtype != torch.int64 or positions.dtype != torch.int64:
        raise RuntimeError("rope_and_store_kv: loc / positions must be int64")
    if k_buffer.stride(-1) != 1 or v_buffer.stride(-1) != 1:
        raise RuntimeError("rope_and_store_kv: pool rows must be contiguous")
    lib = _lib.load()
    check(lib.semipd_rope_kv_store(ptr(q), ptr(k), ptr(v), ptr(k_buffer), ptr(v_buffer), ptr(loc),
                                   ptr(cos_sin_cache), ptr(positions), nnz, q.shape[1], k.shape[1],
                                   head_size, v_head, cos_sin_cache.shape[1], q.stride(0), k.stride(0),
                                   v.stride(0), k_buffer.stride(0), v_buffer.stride(0),
                                   0 if is_neox else 1, dtype_code(query.dtype),
                                   _lib.kv_dtype_code(k_buffer.dtype), current_stream(query.device)),
          "rope_kv_store")


def store_kv_rows(buffer: torch.Tensor, loc: torch.Tensor, src: torch.Tensor) -> None:
    """buffer[loc] = src  (memory_pool.py:345-346, 451-452); rows are the trailing dims."""
    if loc.dtype != torch.int64:
        loc = loc.long()
    n = src.shape[0]
    row_elems = src[0].numel() if n else 0
    if n and row_elems != buffer[0].numel():
        raise RuntimeError(f"store_kv_rows: rows of {row_elems} elements into a pool with rows of "
                           f"{buffer[0].numel()}")
    if n and not (src[0].is_contiguous() and buffer[0].is_contiguous()):
        raise RuntimeError("store_kv_rows: rows must be contiguous")
    lib = _lib.load()
    if buffer.dtype in (torch.float8_e5m2, torch.float8_e4m3fn):
        # fp8 pool: `cache_k.to(self.dtype)` of set_kv_buffer happens inside the scatter
        check(lib.semipd_kv_store_cvt(ptr(buffer), ptr(src), ptr(loc), n, row_elems, buffer.stride(0),
                                      src.stride(0) if n else 0, dtype_code(src.dtype),
                                      _lib.kv_dtype_code(buffer.dtype), current_stream(src.device)), "kv_store_cvt")
        return
    if src.dtype != buffer.dtype:
        raise RuntimeError("store_kv_rows: dtype mismatch")
    es = src.element_size()
    check(lib.semipd_kv_store(ptr(buffer), ptr(src), ptr(loc), n, row_elems * es, buffer.stride(0) * es,
                              src.stride(0) * es if n else 0, current_stream(src.device)), "kv_store")


def gather_rows(src: torch.Tensor, index: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = src[index[i]] for 2-D src (last-token gather, logits_processor.py:232-260)."""
    if src.dim() != 2 or src.stride(1) != 1:
        raise RuntimeError("gather_rows: src must be 2-D with contiguous rows")
    if index.dtype != torch.int64:
        index = index.long()
    n = index.shape[0]
    if out is None:
        out = torch.empty((n, src.shape[1]), dtype=src.dtype, device=src.device)
    es = src.element_size()
    lib = _lib.load()
    check(lib.semipd_gather_rows(ptr(out), ptr(src), ptr(index), n, src.shape[1] * es,
                                 src.stride(0) * es, current_stream(src.device)), "gather_rows")
    return out


# --------------------------------------------------------------------------- kv indices
def create_flashinfer_kv_indices(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                                 page_kernel_lens: torch.Tensor, kv_indptr: torch.Tensor,
                                 kv_start_idx: Optional[torch.Tensor], kv_indices: torch.Tensor) -> None:
    """Fills kv_indptr[:B+1] (cumsum) and kv_indices (layers/attention/utils.py:5-39 plus the
    torch.cumsum at triton_backend.py:96-97)."""
    if req_to_token.dtype != torch.int32 or kv_indptr.dtype != torch.int32 or kv_indices.dtype != torch.int32:
        raise RuntimeError("kv indices must be int32")
    if req_pool_indices.dtype != torch.int64:
        req_pool_indices = req_pool_indices.long()
    if page_kernel_lens.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("page_kernel_lens must be int32/int64")
    if kv_start_idx is not None and kv_start_idx.dtype != torch.int32:
        kv_start_idx = kv_start_idx.int()
    bs = req_pool_indices.shape[0]
    lib = _lib.load()
    check(lib.semipd_build_kv_indices(ptr(req_to_token), req_to_token.stride(0), ptr(req_pool_indices),
                                      ptr(page_kernel_lens), 1 if page_kernel_lens.dtype == torch.int64 else 0,
                                      ptr(kv_start_idx), ptr(kv_indptr), ptr(kv_indices), bs,
                                      current_stream(req_to_token.device)), "build_kv_indices")


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor,
                     extend_seq_lens_sum: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """positions, extend_start_loc (model_executor/forward_batch_info.py:393-466)."""
    bs = extend_seq_lens.shape[0]
    dev = extend_seq_lens.device
    positions = torch.empty(extend_seq_lens_sum, dtype=torch.int64, device=dev)
    start_loc = torch.empty(bs, dtype=torch.int32, device=dev)
    lib = _lib.load()
    check(lib.semipd_compute_positions(ptr(extend_prefix_lens.int()), ptr(extend_seq_lens.int()),
                                       ptr(positions), ptr(start_loc), bs, current_stream(dev)),
          "compute_positions")
    return positions, start_loc


# --------------------------------------------------------------------------- attention
def decode_attention_fwd(q: torch.Tensor, k_buffer: torch.Tensor, v_buffer: torch.Tensor,
                         o: torch.Tensor, kv_indptr: torch.Tensor, kv_indices: torch.Tensor,
                         attn_logits: Optional[torch.Tensor], num_kv_splits: int, sm_scale: float,
                         logit_cap: float = 0.0) -> None:
    """q [B,Hq,Dk]; k_buffer [N,Hkv,Dk]; v_buffer [N,Hkv,Dv]; o [B,Hq,Dv];
    attn_logits fp32 [B,Hq,num_kv_splits,Dv+1]."""
    B, Hq, Dk = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    if attn_logits is not None:
        assert num_kv_splits == attn_logits.shape[2]
        assert B <= attn_logits.shape[0] and attn_logits.dtype == torch.float32
        assert attn_logits.shape[3] == Dv + 1 and attn_logits.is_contiguous()
    assert B <= kv_indptr.shape[0] - 1
    if q.stride(2) != 1 or q.stride(1) != Dk or o.stride(2) != 1 or o.stride(1) != Dv:
        raise RuntimeError("decode_attention_fwd: q / o heads must be densely packed")
    if k_buffer.stride(2) != 1 or k_buffer.stride(1) != Dk or v_buffer.stride(2) != 1 or v_buffer.stride(1) != Dv:
        # MLA: v_buffer is a [..., :512] view of the latent rows -> stride(1) is the row stride, Hkv == 1
        if not (Hkv == 1 and k_buffer.stride(2) == 1 and v_buffer.stride(2) == 1):
            raise RuntimeError("decode_attention_fwd: pool heads must be densely packed")
    lib = _lib.load()
    check(lib.semipd_decode_attention(ptr(o), ptr(q), ptr(k_buffer), ptr(v_buffer), ptr(kv_indptr),
                                      ptr(kv_indices), ptr(attn_logits), B, Hq, Hkv, Dk, Dv,
                                      q.stride(0), o.stride(0), k_buffer.stride(0), v_buffer.stride(0),
                                      num_kv_splits, sm_scale, logit_cap, dtype_code(q.dtype),
                                      _lib.kv_dtype_code(k_buffer.dtype), current_stream(q.device)),
          "decode_attention")


def extend_attention_fwd(q_extend: torch.Tensor, k_extend: torch.Tensor, v_extend: torch.Tensor,
                         o_extend: torch.Tensor, k_buffer: Optional[torch.Tensor],
                         v_buffer: Optional[torch.Tensor], qo_indptr: torch.Tensor,
                         kv_indptr: torch.Tensor, kv_indices: Optional[torch.Tensor],
                         custom_mask: Optional[torch.Tensor], mask_indptr: Optional[torch.Tensor],
                         max_len_extend: int, sm_scale: Optional[float] = None, logit_cap: float = 0.0,
                         skip_prefix_custom_mask: bool = True) -> None:
    """q_extend [T,Hq,Dk], k_extend [T,Hkv,Dk], v_extend [T,Hkv,Dv], o_extend [T,Hq,Dv].
    custom_mask (bool / uint8, flat) + mask_indptr (int64 [B+1]) + skip_prefix_custom_mask: the arguments of the
    reference call (triton_ops/extend_attention.py:291-307), see include/semipd.h: semipd_extend_attention_masked."""
    T, Hq, Dk = q_extend.shape
    Hkv, Dv = v_extend.shape[1], v_extend.shape[2]
    sm_scale = sm_scale or 1.0 / (Dk ** 0.5)
    for name, t, d in (("q", q_extend, Dk), ("k", k_extend, Dk), ("v", v_extend, Dv), ("o", o_extend, Dv)):
        # a single-head tensor may be a column slice of wider rows (MLA: v = latent[..., :512])
        if t.stride(2) != 1 or (t.shape[1] > 1 and t.stride(1) != d):
            raise RuntimeError(f"extend_attention_fwd: {name}_extend heads must be densely packed")
    kb_stride = k_buffer.stride(0) if k_buffer is not None else 0
    vb_stride = v_buffer.stride(0) if v_buffer is not None else 0
    B = qo_indptr.shape[0] - 1
    lib = _lib.load()
    tail = (B, Hq, Hkv, Dk, Dv, q_extend.stride(0), k_extend.stride(0), v_extend.stride(0), o_extend.stride(0),
            kb_stride, vb_stride, int(max_len_extend), sm_scale, logit_cap, dtype_code(q_extend.dtype),
            _lib.kv_dtype_code(k_buffer.dtype if k_buffer is not None else q_extend.dtype),
            current_stream(q_extend.device))
    if custom_mask is not None:
        if mask_indptr is None:
            raise RuntimeError("extend_attention_fwd: custom_mask needs mask_indptr")
        if custom_mask.dtype not in (torch.bool, torch.uint8) or custom_mask.dim() != 1 or not custom_mask.is_contiguous():
            raise RuntimeError("extend_attention_fwd: custom_mask must be a flat, contiguous bool / uint8 tensor")
        if mask_indptr.dtype != torch.int64 or mask_indptr.dim() != 1 or mask_indptr.shape[0] < B + 1 \
                or not mask_indptr.is_contiguous():
            raise RuntimeError("extend_attention_fwd: mask_indptr must be a contiguous int64 tensor of batch + 1 entries")
        if custom_mask.device != q_extend.device or mask_indptr.device != q_extend.device:
            raise RuntimeError("extend_attention_fwd: custom_mask / mask_indptr on another device")
        check(lib.semipd_extend_attention_masked(ptr(o_extend), ptr(q_extend), ptr(k_extend), ptr(v_extend),
                                                 ptr(k_buffer), ptr(v_buffer), ptr(qo_indptr), ptr(kv_indptr),
                                                 ptr(kv_indices), ptr(custom_mask), ptr(mask_indptr),
                                                 1 if skip_prefix_custom_mask else 0, *tail),
              "extend_attention_masked")
        return
    check(lib.semipd_extend_attention(ptr(o_extend), ptr(q_extend), ptr(k_extend), ptr(v_extend),
                                      ptr(k_buffer), ptr(v_buffer), ptr(qo_indptr), ptr(kv_indptr),
                                      ptr(kv_indices), *tail),
          "extend_attention")


# --------------------------------------------------------------------------- sampling
def greedy_argmax(logits: torch.Tensor, out_dtype: torch.dtype = torch.int32) -> torch.Tensor:
    """Sampler greedy branch: torch.argmax(logits, -1) (layers/sampler.py:72-74)."""
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("greedy_argmax: logits must be 2-D with contiguous rows")
    out = torch.empty(logits.shape[0], dtype=out_dtype, device=logits.device)
    lib = _lib.load()
    check(lib.semipd_argmax(ptr(logits), ptr(out), logits.shape[0], logits.shape[1], logits.stride(0),
                            dtype_code(logits.dtype), 1 if out_dtype == torch.int64 else 0,
                            current_stream(logits.device)), "argmax")
    return out


def lm_head_argmax(hidden: torch.Tensor, weight: torch.Tensor, return_logits: bool = True,
                   out_dtype: torch.dtype = torch.int32):
    """fp32 logits = hidden @ weight.T (bf16 MFMA, fp32 accumulate) and their argmax
    (layers/logits_processor.py:394-445 + layers/sampler.py:72-74)."""
    if hidden.dim() != 2 or weight.dim() != 2 or hidden.shape[1] != weight.shape[1]:
        raise RuntimeError("lm_head_argmax: shape mismatch")
    if not (hidden.is_contiguous() and weight.is_contiguous()) or hidden.dtype != weight.dtype:
        raise RuntimeError("lm_head_argmax: contiguous tensors of one dtype required")
    B, V = hidden.shape[0], weight.shape[0]
    logits = torch.empty((B, V), dtype=torch.float32, device=hidden.device)
    out = torch.empty(B, dtype=out_dtype, device=hidden.device)
    lib = _lib.load()
    check(lib.semipd_lm_head_argmax(ptr(hidden), ptr(weight), ptr(logits), ptr(out), None, B,
                                    hidden.shape[1], V, dtype_code(hidden.dtype),
                                    1 if out_dtype == torch.int64 else 0,
                                    current_stream(hidden.device)), "lm_head_argmax")
    return (logits if return_logits else None), out


# --------------------------------------------------------------------------- decode-sized dense layers
_LINEAR_WS = {}     # (device index, hipStream_t) -> [workspace, generation]
_LINEAR_WS_BYTES = 64 << 20  # split-K partial planes + tile counters, shared by every layer on ONE stream of a process


def _linear_workspace_entry(device: torch.device):
    """Workspace of the decode-sized GEMMs, one per (device, stream): launches on one stream are ordered, so one
    buffer serves every layer there; a second compute stream in the process (a CU-masked stream, a co-located
    instance) gets its own and cannot overwrite planes that a consumer on the first stream has yet to read.
    Every call bumps the entry's generation: a SplitKPlanes remembers the one it was written under."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, current_stream(device))
    ent = _LINEAR_WS.get(key)
    if ent is None:
        ent = [torch.zeros(_LINEAR_WS_BYTES, dtype=torch.uint8, device=device), 0]  # counters must start at zero
        _LINEAR_WS[key] = ent
    ent[1] += 1
    return ent


def _linear_workspace(device: torch.device) -> torch.Tensor:
    return _linear_workspace_entry(device)[0]


def linear_is_supported(x: torch.Tensor, weight: torch.Tensor, max_rows: int = 256) -> bool:
    return (x.is_cuda and x.dim() == 2 and weight.dim() == 2 and 0 < x.shape[0] <= max_rows
            and x.dtype == weight.dtype and x.dtype in (torch.bfloat16, torch.float16)
            and x.stride(1) == 1 and weight.is_contiguous() and x.shape[1] == weight.shape[1]
            and weight.shape[1] % 32 == 0 and weight.shape[0] % 4 == 0 and x.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0)


def linear(x: torch.Tensor, weight: torch.Tensor, num_cus: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ weight.T for decode batches (UnquantizedLinearMethod.apply, layers/linear.py:165-172) with the
    weight-streaming split-K kernel.  num_cus = compute units of this process's CU-mask share."""
    if not linear_is_supported(x, weight):
        raise RuntimeError("linear: unsupported shapes / dtypes / strides for the weight-streaming kernel")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    elif out.shape != (M, N) or out.dtype != x.dtype or out.stride(1) != 1:
        raise RuntimeError("linear: bad out tensor")
    ws = _linear_workspace(x.device)
    check(_lib.load().semipd_linear(ptr(out), ptr(x), ptr(weight), ptr(ws), ws.numel(), M, N, K, x.stride(0),
                                    out.stride(0), int(num_cus), dtype_code(x.dtype), current_stream(x.device)),
          "linear")
    return out


# --------------------------------------------------------------------------- tall decode batches: tiled ping-pong GEMM
def gemm_tall_is_supported(x: torch.Tensor, weight: torch.Tensor, fuse_silu_mul: bool = False) -> bool:
    if not (x.is_cuda and x.dim() == 2 and weight.dim() == 2 and x.shape[0] > 0 and x.dtype == weight.dtype
            and x.dtype in (torch.bfloat16, torch.float16) and x.stride(1) == 1 and weight.is_contiguous()
            and x.shape[1] == weight.shape[1]):
        return False
    N, K = weight.shape
    n_out = N // 2 if fuse_silu_mul else N
    return (K % 64 == 0 and n_out % 16 == 0 and (not fuse_silu_mul or N % 2 == 0) and x.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0)


def gemm_tall_set_form(waves: int) -> None:
    """0 | 4 | 8: which kernel runs gemm_tall's 256 x 256 tiles (csrc/gemm8p.hip: gemm4w_kernel | gemm8p_kernel; 0 = by epilogue)."""
    check(_lib.load().semipd_gemm_tall_set_form(int(waves)), "gemm_tall_set_form")


def gemm_tall(x: torch.Tensor, weight: torch.Tensor, fuse_silu_mul: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ weight.T (optionally SiluAndMul of it) with the 256 x 256 ping-pong tile kernel (csrc/gemm8p.hip): decode
    batches above the streaming kernel's 64 rows, vocabulary-sized heads (UnquantizedLinearMethod.apply
    layers/linear.py:165-172; _get_logits layers/logits_processor.py:394-445)."""
    if not gemm_tall_is_supported(x, weight, fuse_silu_mul):
        raise RuntimeError("gemm_tall: unsupported shapes / dtypes / strides")
    M, K = x.shape
    N = weight.shape[0]
    n_out = N // 2 if fuse_silu_mul else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    elif out.shape != (M, n_out) or out.dtype != x.dtype or out.stride(1) != 1:
        raise RuntimeError("gemm_tall: bad out tensor")
    ws = _linear_workspace(x.device)
    check(_lib.load().semipd_gemm_tall(ptr(out), ptr(x), ptr(weight), ptr(ws), ws.numel(), M, N, K, x.stride(0),
                                       out.stride(0), 1 if fuse_silu_mul else 0, dtype_code(x.dtype),
                                       current_stream(x.device)), "gemm_tall")
    return out


def gemm_tall_planes(x: torch.Tensor, weight: torch.Tensor):
    """gemm_tall stopped before the reduction over its K slices (semipd_gemm_tall_planes): a SplitKPlanes for
    fused_add_rmsnorm_planes when the launch sliced K, the [rows, n] tensor when it did not."""
    if not gemm_tall_is_supported(x, weight):
        raise RuntimeError("gemm_tall_planes: unsupported shapes / dtypes / strides")
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    ent = _linear_workspace_entry(x.device)
    ws = ent[0]
    import ctypes as _C
    ks = _C.c_int(0)
    stream = current_stream(x.device)
    check(_lib.load().semipd_gemm_tall_planes(ptr(out), ptr(ws), ws.numel(), ptr(x), ptr(weight), M, N, K, x.stride(0),
                                              out.stride(0), dtype_code(x.dtype), _C.addressof(ks), stream),
          "gemm_tall_planes")
    if int(ks.value) <= 1:
        return out
    return SplitKPlanes(ws, int(ks.value), M, N, x.dtype, ent, stream)


# --------------------------------------------------------------------------- prefill-sized dense layers on a CU share
_DENSE_GEMM = {"ready": False, "tuned": set(), "cus": 0}


def set_declared_cus(cus: int) -> None:
    """The CU count of the stream the following GEMM choices are made for (ModelRunner.set_owned_cus): what was timed on the
    prefill share says nothing about the whole chip, so the tuned flag and the tiled-GEMM preference are filed per count
    like the C table (semipd_dense_gemm_set_cus)."""
    _DENSE_GEMM["cus"] = int(cus)


def dense_gemm_library_version() -> int:
    import ctypes as _C
    v = _C.c_int(0)
    check(_lib.load().semipd_dense_gemm_library_version(_C.addressof(v)), "dense_gemm_library_version")
    return int(v.value)


def dense_gemm_tune(n: int, k: int, rows, dtype: torch.dtype, num_full_search: int = 0, num_heuristics: int = 64,
                    max_solutions: int = 0) -> None:
    """Time hipBLASLt's solutions for a [n, k] weight at the row counts `rows` on the CUs THIS process owns and remember
    the winners (semipd_dense_gemm_tune; start-up only: allocates scratch operands and synchronises).  Candidates: the
    library's first `num_heuristics` heuristic results, plus all of its solutions at the first `num_full_search` row counts."""
    import ctypes as _C
    lib = _lib.load()
    check(lib.semipd_dense_gemm_init(0), "dense_gemm_init")
    arr = (_C.c_int64 * len(rows))(*[int(r) for r in rows])
    check(lib.semipd_dense_gemm_tune(int(n), int(k), _C.addressof(arr), len(rows), int(num_full_search), dtype_code(dtype),
                                     int(num_heuristics), int(max_solutions), current_stream(None)), "dense_gemm_tune")
    torch.cuda.synchronize()
    _DENSE_GEMM["ready"] = True
    _DENSE_GEMM["tuned"].add((_DENSE_GEMM["cus"], int(n), int(k), dtype))


def dense_gemm_import(text: str, shapes) -> int:
    """Take a tuning table written by dense_gemm_report (a start-up cache) instead of timing; `shapes` = the (n, k, dtype)
    the table was made for (they are marked tuned when at least one of their lines was taken).  Returns entries taken."""
    import ctypes as _C
    lib = _lib.load()
    check(lib.semipd_dense_gemm_init(0), "dense_gemm_init")
    got = _C.c_int(0)
    check(lib.semipd_dense_gemm_import(text.encode(), _C.addressof(got)), "dense_gemm_import")
    if got.value:
        _DENSE_GEMM["ready"] = True
        # what counts as tuned is what the C side HOLDS after the import (it skips a line whose solution does not validate or
        # whose kernel name does not match this library build), read back through its own report -- not what the text says
        want = {(int(n), int(k), dt) for n, k, dt in shapes}
        codes = {dtype_code(dt): dt for dt in (torch.bfloat16, torch.float16)}
        for line in dense_gemm_report().splitlines():
            f = dict(kv.split("=", 1) for kv in line.split() if "=" in kv)
            try:
                key = (int(f["n"]), int(f["k"]), codes[int(f["dtype"])])
                if key in want:
                    _DENSE_GEMM["tuned"].add((int(f["cus"]),) + key)
            except (KeyError, ValueError):
                continue
    return int(got.value)


def dense_gemm_is_tuned(weight: torch.Tensor) -> bool:
    return _DENSE_GEMM["ready"] and (_DENSE_GEMM["cus"], weight.shape[0], weight.shape[1], weight.dtype) in _DENSE_GEMM["tuned"]


# Prefill-sized batches of a layer whose weight shape was timed at start-up: where the tiled ping-pong GEMM (csrc/gemm8p.hip)
# beat the library's measured winner on this process's CU share (ModelRunner.tune_dense_gemms: e.g. down_proj at 1024 rows on
# 208 CUs, 117 us against 155; gate_up + SiLU at 2048 rows, 443 against 457 + 50).  (n, k, dtype, fuse_silu) -> [(rows, bool)]
_TALL_PREF = {}


def set_tall_preference(n: int, k: int, dtype: torch.dtype, fuse_silu_mul: bool, rows_and_wins) -> None:
    _TALL_PREF[(_DENSE_GEMM["cus"], int(n), int(k), dtype, bool(fuse_silu_mul))] = sorted((int(r), bool(w)) for r, w in rows_and_wins)


def tall_preferred(weight: torch.Tensor, rows: int, fuse_silu_mul: bool = False) -> bool:
    """Did gemm_tall win at the timed row count nearest (in log distance) to `rows` for this weight's shape?"""
    ent = _TALL_PREF.get((_DENSE_GEMM["cus"], weight.shape[0], weight.shape[1], weight.dtype, bool(fuse_silu_mul)))
    if not ent:
        return False
    import math
    best = min(ent, key=lambda rw: abs(math.log2(rw[0]) - math.log2(max(1, rows))))
    return best[1]


def dense_gemm_report() -> str:
    import ctypes as _C
    lib = _lib.load()
    need = lib.semipd_dense_gemm_report(None, 0)
    buf = _C.create_string_buffer(int(need))
    lib.semipd_dense_gemm_report(_C.addressof(buf), need)
    return buf.value.decode()


def dense_gemm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ weight.T (+ bias) for prefill-sized batches with the library solution measured fastest on this process's CU
    share (UnquantizedLinearMethod.apply -> F.linear, layers/linear.py:165-172)."""
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1] or x.dtype != weight.dtype:
        raise RuntimeError("dense_gemm: x [rows, k] and weight [n, k] of one dtype expected")
    if x.dtype not in (torch.bfloat16, torch.float16) or x.stride(1) != 1 or not weight.is_contiguous():
        raise RuntimeError("dense_gemm: bf16 / f16, unit inner stride, contiguous weight required")
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != weight.shape[0] or not bias.is_contiguous()):
        raise RuntimeError("dense_gemm: bias must be a contiguous [n] vector of the activation dtype")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    elif out.shape != (M, N) or out.dtype != x.dtype or out.stride(1) != 1:
        raise RuntimeError("dense_gemm: bad out tensor")
    check(_lib.load().semipd_dense_gemm(ptr(out), ptr(x), ptr(weight), ptr(bias), M, N, K, x.stride(0), out.stride(0),
                                        dtype_code(x.dtype), current_stream(x.device)), "dense_gemm")
    return out


STREAM_LINEAR_MAX_ROWS = 64      # the fp32, fused lm_head and grouped (MoE) forms of the streaming kernel
# dense layers (semipd_stream_linear / _planes): up to 128 rows -- four waves of 2 x 16 weight rows, an activation ring of two
# blocks under weight rings of three (csrc/stream_linear.hip).  SEMIPD_SL_WIDE=0: 65+ rows go to the tiled GEMM as before.
STREAM_LINEAR_DENSE_MAX_ROWS = 128 if os.environ.get("SEMIPD_SL_WIDE", "1") != "0" else 64


def stream_linear_is_supported(x: torch.Tensor, weight: torch.Tensor, fuse_silu_mul: bool = False) -> bool:
    if not (x.is_cuda and x.dim() == 2 and weight.dim() == 2 and 0 < x.shape[0] <= STREAM_LINEAR_DENSE_MAX_ROWS
            and x.dtype == weight.dtype and x.dtype in (torch.bfloat16, torch.float16)
            and x.stride(1) == 1 and weight.is_contiguous() and x.shape[1] == weight.shape[1]):
        return False
    N, K = weight.shape
    n_out = N // 2 if fuse_silu_mul else N
    return (K % 128 == 0 and n_out % 16 == 0 and (not fuse_silu_mul or N % 2 == 0)
            and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def stream_linear(x: torch.Tensor, weight: torch.Tensor, fuse_silu_mul: bool = False,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ weight.T for decode batches (at most 128 rows) with the LDS-DMA weight-streaming kernel; fuse_silu_mul:
    weight = merged [gate; up] and the result is SiluAndMul(x @ weight.T)
    (UnquantizedLinearMethod.apply layers/linear.py:165-172; LlamaMLP models/llama.py:88-92)."""
    if not stream_linear_is_supported(x, weight, fuse_silu_mul):
        raise RuntimeError("stream_linear: unsupported shapes / dtypes / strides for the weight-streaming kernel")
    M, K = x.shape
    N = weight.shape[0]
    n_out = N // 2 if fuse_silu_mul else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    elif out.shape != (M, n_out) or out.dtype != x.dtype or out.stride(1) != 1:
        raise RuntimeError("stream_linear: bad out tensor")
    ws = _linear_workspace(x.device)   # fp32 K-slice planes; calls on one stream are ordered, so one buffer serves all
    check(_lib.load().semipd_stream_linear(ptr(out), ptr(x), ptr(weight), ptr(ws), ws.numel(), M, N, K, x.stride(0),
                                           out.stride(0), 1 if fuse_silu_mul else 0, dtype_code(x.dtype),
                                           current_stream(x.device)), "stream_linear")
    return out


class SplitKPlanes:
    """The output of a decode-batch GEMM before its K-slice reduction: fp32 planes [ksplit, rows, n] in the
    workspace of the (device, stream) it was launched on, valid until the next GEMM that uses that workspace.  It is
    NOT a tensor: the consumer (`fused_add_rmsnorm_planes`, `rope_and_store_kv_planes`) must be the next workspace user
    on the same stream, and `check_live` enforces that with the generation stamp taken when the planes were written."""
    __slots__ = ("planes", "ksplit", "rows", "n", "dtype", "_entry", "_generation", "_stream")

    def __init__(self, planes, ksplit, rows, n, dtype, entry=None, stream=None):
        self.planes, self.ksplit, self.rows, self.n, self.dtype = planes, ksplit, rows, n, dtype
        self._entry, self._generation, self._stream = entry, (entry[1] if entry is not None else None), stream

    @property
    def shape(self):
        return (self.rows, self.n)

    def check_live(self, consumer: str) -> None:
        if self._entry is None:
            return
        if self._entry[1] != self._generation:
            raise RuntimeError(f"{consumer}: the K-slice planes were overwritten by a later GEMM on their stream "
                               f"(written at workspace generation {self._generation}, now {self._entry[1]}); the "
                               "consumer of a deferred reduction must directly follow its producer")
        if current_stream(self.planes.device) != self._stream:
            raise RuntimeError(f"{consumer}: K-slice planes must be consumed on the stream that produced them")


def stream_linear_planes(x: torch.Tensor, weight: torch.Tensor) -> SplitKPlanes:
    """x @ weight.T for decode batches, stopped before the reduction over the K slices (semipd_stream_linear_planes)."""
    if not stream_linear_is_supported(x, weight):
        raise RuntimeError("stream_linear_planes: unsupported shapes / dtypes / strides")
    M, K = x.shape
    N = weight.shape[0]
    ent = _linear_workspace_entry(x.device)
    ws = ent[0]
    import ctypes as _C
    ks = _C.c_int(0)
    stream = current_stream(x.device)
    check(_lib.load().semipd_stream_linear_planes(ptr(ws), ws.numel(), ptr(x), ptr(weight), M, N, K, x.stride(0),
                                                  dtype_code(x.dtype), _C.addressof(ks), stream),
          "stream_linear_planes")
    return SplitKPlanes(ws, int(ks.value), M, N, x.dtype, ent, stream)


def rope_and_store_kv_planes(positions: torch.Tensor, qkv: SplitKPlanes, num_q_heads: int, num_kv_heads: int,
                             head_size: int, cos_sin_cache: torch.Tensor, k_buffer: torch.Tensor, v_buffer: torch.Tensor,
                             loc: torch.Tensor) -> torch.Tensor:
    """rope_and_store_kv on a qkv row that is still the K-slice planes of the decode GEMM: returns the rotated q
    [tokens, Hq * head]; rotated k and v go to the pool rows `loc` (semipd_rope_kv_store_planes)."""
    if cos_sin_cache.dtype != torch.float32 or cos_sin_cache.shape[1] != head_size:
        raise RuntimeError("rope_and_store_kv_planes: fp32 cos / sin cache over the whole head expected")
    if qkv.n != (num_q_heads + 2 * num_kv_heads) * head_size:
        raise RuntimeError("rope_and_store_kv_planes: planes do not hold a [q | k | v] row")
    if positions.dtype != torch.int64:
        positions = positions.long()
    if loc.dtype != torch.int64:
        raise RuntimeError("rope_and_store_kv_planes: loc must be int64 pool rows (the kernel reads 8-byte indices)")
    if loc.numel() != qkv.rows or not loc.is_contiguous():
        raise RuntimeError("rope_and_store_kv_planes: one contiguous pool row index per token expected")
    for name, buf in (("k_buffer", k_buffer), ("v_buffer", v_buffer)):
        if buf.dim() != 3 or buf.shape[1:] != (num_kv_heads, head_size) or buf.stride(2) != 1 \
                or buf.stride(1) != head_size:
            raise RuntimeError(f"rope_and_store_kv_planes: {name} must be [slots, kv heads, head] with dense rows")
    if k_buffer.dtype != v_buffer.dtype:
        raise RuntimeError("rope_and_store_kv_planes: k_buffer / v_buffer dtype mismatch")
    qkv.check_live("rope_and_store_kv_planes")
    q = torch.empty((qkv.rows, num_q_heads * head_size), dtype=qkv.dtype, device=qkv.planes.device)
    check(_lib.load().semipd_rope_kv_store_planes(ptr(q), ptr(qkv.planes), qkv.ksplit, qkv.rows * qkv.n, ptr(k_buffer),
                                                  ptr(v_buffer), ptr(loc), ptr(cos_sin_cache), ptr(positions), qkv.rows,
                                                  num_q_heads, num_kv_heads, head_size, q.stride(0), k_buffer.stride(0),
                                                  v_buffer.stride(0), dtype_code(qkv.dtype), _lib.kv_dtype_code(k_buffer.dtype),
                                                  current_stream(q.device)), "rope_and_store_kv_planes")
    return q


def decode_rope_attention_planes_supported(num_q_heads: int, num_kv_heads: int, head_size: int, dtype: torch.dtype,
                                           kv_dtype: torch.dtype) -> bool:
    """Whether semipd_decode_rope_attention_planes is instantiated for this head geometry (csrc/decode_attention_fused.hip)."""
    return bool(_lib.load().semipd_decode_rope_attention_planes_supported(
        num_q_heads, num_kv_heads, head_size, dtype_code(dtype), _lib.kv_dtype_code(kv_dtype)))


def decode_rope_attention_planes(positions: torch.Tensor, qkv: SplitKPlanes, num_q_heads: int, num_kv_heads: int,
                                 head_size: int, cos_sin_cache: torch.Tensor, k_buffer: torch.Tensor, v_buffer: torch.Tensor,
                                 loc: torch.Tensor, kv_indptr: torch.Tensor, kv_indices: torch.Tensor, waves: int,
                                 sm_scale: float, logit_cap: float = 0.0, zsplits: int = 1,
                                 attn_logits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rope_and_store_kv_planes + decode_attention_fwd (`waves` kv splits, both stages) in ONE launch: the qkv row is
    still the K-slice planes of the decode GEMM; rotated k and v go to the pool rows `loc` (the last row of each request
    in kv_indices); returns the attention output [tokens, Hq * head] (semipd_decode_rope_attention_planes).  Same bits
    as the separate calls.  zsplits > 1 (small batches): that many workgroups per (request, kv head), each writing one
    merged partial to attn_logits fp32 [tokens, Hq, zsplits, head + 1]; stage 2 follows as a second launch."""
    if cos_sin_cache.dtype != torch.float32 or cos_sin_cache.shape[1] != head_size or not cos_sin_cache.is_contiguous():
        raise RuntimeError("decode_rope_attention_planes: contiguous fp32 cos / sin cache over the whole head expected")
    if qkv.n != (num_q_heads + 2 * num_kv_heads) * head_size:
        raise RuntimeError("decode_rope_attention_planes: planes do not hold a [q | k | v] row")
    if positions.dtype != torch.int64:
        positions = positions.long()
    if loc.dtype != torch.int64 or loc.numel() != qkv.rows or not loc.is_contiguous():
        raise RuntimeError("decode_rope_attention_planes: one contiguous int64 pool row index per token expected")
    if positions.numel() != qkv.rows or not positions.is_contiguous():
        raise RuntimeError("decode_rope_attention_planes: one contiguous position per token expected")
    for name, buf in (("k_buffer", k_buffer), ("v_buffer", v_buffer)):
        if buf.dim() != 3 or buf.shape[1:] != (num_kv_heads, head_size) or buf.stride(2) != 1 \
                or buf.stride(1) != head_size:
            raise RuntimeError(f"decode_rope_attention_planes: {name} must be [slots, kv heads, head] with dense rows")
    if k_buffer.dtype != v_buffer.dtype:
        raise RuntimeError("decode_rope_attention_planes: k_buffer / v_buffer dtype mismatch")
    if kv_indptr.dtype != torch.int32 or kv_indices.dtype != torch.int32 or qkv.rows > kv_indptr.shape[0] - 1:
        raise RuntimeError("decode_rope_attention_planes: int32 kv_indptr [tokens + 1] / kv_indices expected")
    if zsplits > 1:
        if (attn_logits is None or attn_logits.dtype != torch.float32 or not attn_logits.is_contiguous()
                or attn_logits.numel() < qkv.rows * num_q_heads * zsplits * (head_size + 1)):
            raise RuntimeError("decode_rope_attention_planes: contiguous fp32 attn_logits of at least "
                               "[tokens, Hq, zsplits, head + 1] required when zsplits > 1")
    qkv.check_live("decode_rope_attention_planes")
    o = torch.empty((qkv.rows, num_q_heads * head_size), dtype=qkv.dtype, device=qkv.planes.device)
    check(_lib.load().semipd_decode_rope_attention_planes(
        ptr(o), ptr(attn_logits) if zsplits > 1 else None, ptr(qkv.planes), qkv.ksplit, qkv.rows * qkv.n, ptr(k_buffer),
        ptr(v_buffer), ptr(loc), ptr(cos_sin_cache), ptr(positions), ptr(kv_indptr), ptr(kv_indices), qkv.rows,
        num_q_heads, num_kv_heads, head_size, o.stride(0), k_buffer.stride(0), v_buffer.stride(0), waves, int(zsplits),
        sm_scale, logit_cap, dtype_code(qkv.dtype),
        _lib.kv_dtype_code(k_buffer.dtype), current_stream(o.device)), "decode_rope_attention_planes")
    return o


def mla_decode_prep(qkv_a: SplitKPlanes, positions: torch.Tensor, cos_sin_cache: torch.Tensor, norm_weight: torch.Tensor,
                    eps: float, num_heads: int, nope_dim: int, rope_dim: int, lora_rank: int, kv_buffer: torch.Tensor,
                    loc: torch.Tensor, q_input: torch.Tensor) -> torch.Tensor:
    """MLA decode step between the merged [q | kv_a] GEMM (still K-slice planes) and the attention: returns q_nope
    [tokens, heads, nope]; the rotated q_pe goes to q_input[..., lora:], the normalised latent + rotated k_pe to the pool
    rows `loc` (semipd_mla_decode_prep).  One launch for two reductions, RMSNorm, RoPE, the q_pe copy and the KV store."""
    if qkv_a.n != num_heads * (nope_dim + rope_dim) + lora_rank + rope_dim:
        raise RuntimeError("mla_decode_prep: planes do not hold a [q | latent | k_pe] row")
    if cos_sin_cache.dtype != torch.float32 or cos_sin_cache.shape[1] != rope_dim or not cos_sin_cache.is_contiguous():
        raise RuntimeError("mla_decode_prep: contiguous fp32 cos / sin cache [max_pos, rope] expected")
    if positions.dtype != torch.int64:
        positions = positions.long()
    if loc.dtype != torch.int64 or loc.numel() != qkv_a.rows or not loc.is_contiguous():
        raise RuntimeError("mla_decode_prep: one contiguous int64 pool row per token expected")
    if q_input.shape != (qkv_a.rows, num_heads, lora_rank + rope_dim) or q_input.stride(2) != 1 or q_input.dtype != qkv_a.dtype:
        raise RuntimeError("mla_decode_prep: q_input must be [tokens, heads, lora + rope] of the GEMM's dtype")
    if kv_buffer.dim() != 3 or kv_buffer.shape[1:] != (1, lora_rank + rope_dim) or kv_buffer.stride(2) != 1:
        raise RuntimeError("mla_decode_prep: kv_buffer must be [slots, 1, lora + rope] with dense rows")
    qkv_a.check_live("mla_decode_prep")
    q_nope = torch.empty((qkv_a.rows, num_heads, nope_dim), dtype=qkv_a.dtype, device=qkv_a.planes.device)
    check(_lib.load().semipd_mla_decode_prep(ptr(q_nope), ptr(q_input), ptr(kv_buffer), ptr(qkv_a.planes), qkv_a.ksplit,
                                             qkv_a.rows * qkv_a.n, ptr(loc), ptr(cos_sin_cache), ptr(positions),
                                             ptr(norm_weight), float(eps), qkv_a.rows, num_heads, nope_dim, rope_dim, lora_rank,
                                             q_input.stride(0), q_input.stride(1), kv_buffer.stride(0),
                                             dtype_code(qkv_a.dtype), _lib.kv_dtype_code(kv_buffer.dtype), current_stream(q_input.device)),
          "mla_decode_prep")
    return q_nope


def mla_decode_prep_rows(q: torch.Tensor, latent: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor,
                         norm_weight: torch.Tensor, eps: float, nope_dim: int, lora_rank: int, kv_buffer: torch.Tensor,
                         loc: torch.Tensor, q_input: torch.Tensor) -> None:
    """mla_decode_prep for rows that are already tensors: q [T, H, nope + rope], latent [T, lora + rope] (any row strides);
    fills q_input[..., lora:] with the rotated q_pe and the pool rows `loc` with the normalised latent + rotated k_pe
    (semipd_mla_decode_prep_rows).  q and latent are left as they are."""
    T, H, qk = q.shape
    rope_dim = qk - nope_dim
    if latent.shape != (T, lora_rank + rope_dim) or q.stride(2) != 1 or latent.stride(1) != 1 or q.dtype != latent.dtype:
        raise RuntimeError("mla_decode_prep_rows: q [T, H, nope + rope] and latent [T, lora + rope] of one dtype expected")
    if cos_sin_cache.dtype != torch.float32 or cos_sin_cache.shape[1] != rope_dim or not cos_sin_cache.is_contiguous():
        raise RuntimeError("mla_decode_prep_rows: contiguous fp32 cos / sin cache [max_pos, rope] expected")
    if positions.dtype != torch.int64:
        positions = positions.long()
    if loc.dtype != torch.int64 or loc.numel() != T or not loc.is_contiguous():
        raise RuntimeError("mla_decode_prep_rows: one contiguous int64 pool row per token expected")
    if q_input.shape != (T, H, lora_rank + rope_dim) or q_input.stride(2) != 1 or q_input.dtype != q.dtype:
        raise RuntimeError("mla_decode_prep_rows: q_input must be [tokens, heads, lora + rope] of q's dtype")
    if kv_buffer.dim() != 3 or kv_buffer.shape[1:] != (1, lora_rank + rope_dim) or kv_buffer.stride(2) != 1:
        raise RuntimeError("mla_decode_prep_rows: kv_buffer must be [slots, 1, lora + rope] with dense rows")
    check(_lib.load().semipd_mla_decode_prep_rows(ptr(q_input), ptr(kv_buffer), ptr(q), ptr(latent), ptr(loc), ptr(cos_sin_cache),
                                                  ptr(positions), ptr(norm_weight), float(eps), T, H, nope_dim, rope_dim,
                                                  lora_rank, q.stride(0), q.stride(1), latent.stride(0), q_input.stride(0),
                                                  q_input.stride(1), kv_buffer.stride(0), dtype_code(q.dtype),
                                                  _lib.kv_dtype_code(kv_buffer.dtype), current_stream(q.device)),
          "mla_decode_prep_rows")


def fused_add_rmsnorm_planes(p: SplitKPlanes, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """residual += T(sum of the planes); returns RMSNorm(residual) * weight -- fused_add_rmsnorm on the GEMM output
    that was never written (layers/layernorm.py:47-76)."""
    if residual is None:
        raise RuntimeError("fused_add_rmsnorm_planes: a residual is required (a layer whose output is deferred planes "
                           "must be followed by the fused add + norm; reduce the planes with splitk_planes_reduce "
                           "otherwise)")
    if residual.shape != (p.rows, p.n) or residual.dtype != p.dtype or not residual.is_contiguous():
        raise RuntimeError("fused_add_rmsnorm_planes: residual must be a contiguous [rows, n] tensor of the GEMM's dtype")
    p.check_live("fused_add_rmsnorm_planes")
    out = torch.empty_like(residual)
    check(_lib.load().semipd_fused_add_rmsnorm_planes(ptr(out), ptr(residual), ptr(weight), ptr(p.planes), p.ksplit,
                                                      p.rows * p.n, p.rows, p.n, float(eps), dtype_code(p.dtype),
                                                      current_stream(residual.device)), "fused_add_rmsnorm_planes")
    return out


# --------------------------------------------------------------------------- stochastic sampling
def _probs_2d(probs: torch.Tensor, name: str) -> Tuple[int, int]:
    if probs.dim() != 2 or probs.dtype != torch.float32 or not probs.is_contiguous():
        raise RuntimeError(f"{name}: probs must be a contiguous fp32 [batch, vocab] tensor")
    return probs.shape[0], probs.shape[1]


def _per_row(value, batch: int, dtype: torch.dtype, device, name: str):
    """(tensor-or-None, scalar): the reference ops accept a python scalar or a [batch] tensor."""
    if isinstance(value, torch.Tensor):
        if value.numel() != batch:
            raise RuntimeError(f"{name}: expected {batch} per-row values, got {value.numel()}")
        return value.to(device=device, dtype=dtype).contiguous(), 0
    return None, value


def softmax_temperature_(logits: torch.Tensor, temperatures: Optional[torch.Tensor]) -> torch.Tensor:
    """logits.div_(temperatures); logits[:] = softmax(logits, -1) in one pass set (sampler.py:78-81)."""
    B, V = _probs_2d(logits, "softmax_temperature_")
    t = None
    if temperatures is not None:
        t = temperatures.reshape(-1).to(torch.float32).contiguous()
        if t.numel() != B:
            raise RuntimeError("softmax_temperature_: one temperature per row required")
    check(_lib.load().semipd_softmax_temperature(ptr(logits), ptr(t) if t is not None else None, B, V,
                                                 current_stream(logits.device)), "softmax_temperature")
    return logits


def top_k_top_p_sampling_from_probs(probs: torch.Tensor, uniform_samples: torch.Tensor, top_k, top_p,
                                    filter_apply_order: str = "joint"):
    """sgl_kernel.top_k_top_p_sampling_from_probs (python/sgl_kernel/sampling.py:139-165):
    returns (samples int32 [batch], success bool [batch])."""
    if filter_apply_order != "joint":
        raise RuntimeError("top_k_top_p_sampling_from_probs: only filter_apply_order='joint' is implemented")
    B, V = _probs_2d(probs, "top_k_top_p_sampling_from_probs")
    if uniform_samples.dim() != 2 or uniform_samples.shape[1] != B or uniform_samples.dtype != torch.float32 \
            or not uniform_samples.is_contiguous():
        raise RuntimeError("top_k_top_p_sampling_from_probs: uniform_samples must be fp32 [rounds, batch]")
    ks, k_val = _per_row(top_k, B, torch.int32, probs.device, "top_k")
    ps, p_val = _per_row(top_p, B, torch.float32, probs.device, "top_p")
    out = torch.empty(B, dtype=torch.int32, device=probs.device)
    success = torch.empty(B, dtype=torch.uint8, device=probs.device)
    check(_lib.load().semipd_top_k_top_p_sampling_from_probs(
        ptr(probs), ptr(uniform_samples), ptr(ks) if ks is not None else None, int(k_val),
        ptr(ps) if ps is not None else None, float(p_val), ptr(out), ptr(success), B, V,
        uniform_samples.shape[0], current_stream(probs.device)), "top_k_top_p_sampling_from_probs")
    return out, success.bool()


def min_p_sampling_from_probs(probs: torch.Tensor, uniform_samples: torch.Tensor, min_p) -> torch.Tensor:
    """sgl_kernel.min_p_sampling_from_probs (python/sgl_kernel/sampling.py:194-210)."""
    B, V = _probs_2d(probs, "min_p_sampling_from_probs")
    u = uniform_samples
    if u.dim() == 2:  # the sampler hands over [rounds, batch]; only the first round is consumed
        u = u[0]
    if u.numel() != B or u.dtype != torch.float32 or not u.is_contiguous():
        raise RuntimeError("min_p_sampling_from_probs: uniform_samples must be fp32 [batch]")
    ms, m_val = _per_row(min_p, B, torch.float32, probs.device, "min_p")
    out = torch.empty(B, dtype=torch.int32, device=probs.device)
    check(_lib.load().semipd_min_p_sampling_from_probs(
        ptr(probs), ptr(u), ptr(ms) if ms is not None else None, float(m_val), ptr(out), B, V,
        current_stream(probs.device)), "min_p_sampling_from_probs")
    return out


def top_k_renorm_prob(probs: torch.Tensor, top_k) -> torch.Tensor:
    """sgl_kernel.top_k_renorm_prob (python/sgl_kernel/sampling.py:25-32)."""
    B, V = _probs_2d(probs, "top_k_renorm_prob")
    ks, k_val = _per_row(top_k, B, torch.int32, probs.device, "top_k")
    out = torch.empty_like(probs)
    check(_lib.load().semipd_top_k_renorm_prob(ptr(probs), ptr(out), ptr(ks) if ks is not None else None,
                                               int(k_val), B, V, current_stream(probs.device)),
          "top_k_renorm_prob")
    return out


def top_p_renorm_prob(probs: torch.Tensor, top_p) -> torch.Tensor:
    """sgl_kernel.top_p_renorm_prob (python/sgl_kernel/sampling.py:53-60)."""
    B, V = _probs_2d(probs, "top_p_renorm_prob")
    ps, p_val = _per_row(top_p, B, torch.float32, probs.device, "top_p")
    out = torch.empty_like(probs)
    check(_lib.load().semipd_top_p_renorm_prob(ptr(probs), ptr(out), ptr(ps) if ps is not None else None,
                                               float(p_val), B, V, current_stream(probs.device)),
          "top_p_renorm_prob")
    return out


def token_logprobs(logits: torch.Tensor, token_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(log_softmax(logits)[b, token_ids[b]], logsumexp(logits[b])) for fp32 logits [batch, vocab]
    (Sampler.forward with return_logprob, layers/sampler.py:74-75, 150-155)."""
    B, V = _probs_2d(logits, "token_logprobs")
    ids = token_ids.reshape(-1).to(torch.int32).contiguous()
    if ids.numel() != B:
        raise RuntimeError("token_logprobs: one token id per row required")
    out = torch.empty(B, dtype=torch.float32, device=logits.device)
    lse = torch.empty(B, dtype=torch.float32, device=logits.device)
    check(_lib.load().semipd_token_logprobs(ptr(logits), ptr(ids), ptr(out), ptr(lse), B, V,
                                            current_stream(logits.device)), "token_logprobs")
    return out, lse


# --------------------------------------------------------------------------- MoE
def topk_softmax(gating_output: torch.Tensor, topk: int, renormalize: bool):
    """fused_topk (layers/moe/topk.py:44-75): fp32 softmax + top-k."""
    T, E = gating_output.shape
    w = torch.empty((T, topk), dtype=torch.float32, device=gating_output.device)
    ids = torch.empty((T, topk), dtype=torch.int32, device=gating_output.device)
    g = gating_output.contiguous()
    lib = _lib.load()
    check(lib.semipd_topk_softmax(ptr(g), ptr(w), ptr(ids), T, E, topk, int(renormalize),
                                  dtype_code(g.dtype), current_stream(g.device)), "topk_softmax")
    return w, ids


def grouped_topk(gating_output: torch.Tensor, topk: int, renormalize: bool, num_expert_group: int,
                 topk_group: int, correction_bias: Optional[torch.Tensor] = None,
                 scoring_func: str = "softmax"):
    """grouped_topk / biased_grouped_topk (layers/moe/topk.py:79-160)."""
    if scoring_func not in ("softmax", "sigmoid"):
        raise ValueError(f"Scoring function '{scoring_func}' is not supported.")
    if isinstance(gating_output, SplitKPlanes):
        # the router GEMM stopped before its K-slice reduction (MoEGate on the weight-streaming kernel): the routing kernel
        # sums the planes itself
        p = gating_output
        p.check_live("grouped_topk")
        T, E = p.rows, p.n
        dev = p.planes.device
        w = torch.empty((T, topk), dtype=torch.float32, device=dev)
        ids = torch.empty((T, topk), dtype=torch.int32, device=dev)
        bias = correction_bias.float().contiguous() if correction_bias is not None else None
        check(_lib.load().semipd_grouped_topk_planes(ptr(p.planes), p.ksplit, p.rows * p.n, ptr(bias), ptr(w), ptr(ids), T, E,
                                                     topk, num_expert_group, topk_group, int(renormalize),
                                                     1 if scoring_func == "sigmoid" else 0, dtype_code(p.dtype),
                                                     current_stream(dev)), "grouped_topk_planes")
        return w, ids
    T, E = gating_output.shape
    w = torch.empty((T, topk), dtype=torch.float32, device=gating_output.device)
    ids = torch.empty((T, topk), dtype=torch.int32, device=gating_output.device)
    g = gating_output.contiguous()
    bias = correction_bias.float().contiguous() if correction_bias is not None else None
    lib = _lib.load()
    check(lib.semipd_grouped_topk(ptr(g), ptr(bias), ptr(w), ptr(ids), T, E, topk, num_expert_group,
                                  topk_group, int(renormalize), 1 if scoring_func == "sigmoid" else 0,
                                  dtype_code(g.dtype), current_stream(g.device)), "grouped_topk")
    return w, ids


def moe_align_block_size(topk_ids: torch.Tensor, num_experts: int, block_size: int,
                         sorted_token_ids: torch.Tensor, experts_ids: torch.Tensor,
                         num_tokens_post_pad: torch.Tensor, token_cnts_buffer: Optional[torch.Tensor],
                         cumsum_buffer: torch.Tensor) -> None:
    """Same argument list as sgl_kernel.moe_align_block_size (python/sgl_kernel/moe.py:4-23);
    token_cnts_buffer is accepted and unused (the histogram lives in LDS)."""
    if topk_ids.dtype != torch.int32:
        raise RuntimeError("moe_align_block_size: topk_ids must be int32")
    if cumsum_buffer.numel() < num_experts + 1:
        raise RuntimeError("moe_align_block_size: cumsum_buffer needs num_experts+1 entries")
    lib = _lib.load()
    check(lib.semipd_moe_align_block_size(ptr(topk_ids), topk_ids.numel(), num_experts, block_size,
                                          ptr(sorted_token_ids), ptr(experts_ids),
                                          ptr(num_tokens_post_pad), ptr(cumsum_buffer),
                                          sorted_token_ids.numel(), current_stream(topk_ids.device)),
          "moe_align_block_size")


def moe_grouped_gemm(a: torch.Tensor, w: torch.Tensor, c: torch.Tensor,
                     topk_weights: Optional[torch.Tensor], sorted_token_ids: torch.Tensor,
                     expert_ids: torch.Tensor, num_tokens_post_pad: torch.Tensor, num_valid: int,
                     top_k_div: int, mul_routed_weight: bool, block_m: int = 64) -> None:
    """invoke_fused_moe_kernel (layers/moe/fused_moe_triton/fused_moe.py:501-612), bf16/f16."""
    E, N, K = w.shape
    if a.shape[-1] != K or c.shape[-1] != N or not (a.is_contiguous() and w.is_contiguous() and c.is_contiguous()):
        raise RuntimeError("moe_grouped_gemm: shape / contiguity mismatch")
    lib = _lib.load()
    check(lib.semipd_moe_grouped_gemm(ptr(c), ptr(a), ptr(w), ptr(topk_weights), ptr(sorted_token_ids),
                                      ptr(expert_ids), ptr(num_tokens_post_pad), num_valid, N, K,
                                      sorted_token_ids.numel(), top_k_div, int(mul_routed_weight),
                                      block_m, dtype_code(a.dtype), current_stream(a.device)),
          "moe_grouped_gemm")


def moe_grouped_gemm_silu(a: torch.Tensor, w: torch.Tensor, sorted_token_ids: torch.Tensor, expert_ids: torch.Tensor,
                          num_tokens_post_pad: torch.Tensor, num_valid: int, top_k_div: int,
                          block_m: int = 128) -> Optional[torch.Tensor]:
    """GEMM1 of fused_experts with SiluAndMul in its epilogue (semipd_moe_grouped_gemm_silu): [num_valid, N / 2], or None
    when the call does not qualify (the caller then runs moe_grouped_gemm + silu_and_mul)."""
    E, N, K = w.shape
    lib = _lib.load()
    if a.shape[-1] != K or not (a.is_contiguous() and w.is_contiguous()) or not lib.semipd_moe_grouped_gemm_silu_supported(
            num_valid, N, K, top_k_div, block_m, dtype_code(a.dtype)):
        return None
    c = torch.empty((num_valid, N // 2), dtype=a.dtype, device=a.device)
    check(lib.semipd_moe_grouped_gemm_silu(ptr(c), ptr(a), ptr(w), ptr(sorted_token_ids), ptr(expert_ids),
                                           ptr(num_tokens_post_pad), num_valid, N, K, sorted_token_ids.numel(), top_k_div,
                                           block_m, dtype_code(a.dtype), current_stream(a.device)), "moe_grouped_gemm_silu")
    return c


def moe_sum_scale_add(x: torch.Tensor, scale: float = 1.0, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """moe_sum(x) * scale + addend in one launch with the roundings of the three (DeepseekV2MoE.forward,
    models/deepseek_v2.py:139-160): x [T, topk, H]; the multiply is skipped for scale == 1 like the model skips it."""
    T, k, H = x.shape
    out = torch.empty((T, H), dtype=x.dtype, device=x.device)
    if isinstance(addend, SplitKPlanes):
        # the addend's GEMM stopped before its K-slice reduction (the shared experts' down_proj): summed in this launch
        if addend.shape != (T, H) or addend.dtype != x.dtype:
            raise RuntimeError("moe_sum_scale_add: the addend's planes must hold a [T, H] result of the same dtype")
        addend.check_live("moe_sum_scale_add")
        check(_lib.load().semipd_moe_sum_scale_add_planes(ptr(out), ptr(x.contiguous()), ptr(addend.planes), addend.ksplit,
                                                          addend.rows * addend.n, T, k, H, float(scale), int(scale != 1.0),
                                                          dtype_code(x.dtype), current_stream(x.device)),
              "moe_sum_scale_add_planes")
        return out
    if addend is not None and (addend.shape != (T, H) or addend.dtype != x.dtype or not addend.is_contiguous()):
        raise RuntimeError("moe_sum_scale_add: addend must be a contiguous [T, H] tensor of the same dtype")
    check(_lib.load().semipd_moe_sum_scale_add(ptr(out), ptr(x.contiguous()), ptr(addend), T, k, H, float(scale),
                                               int(scale != 1.0), dtype_code(x.dtype), current_stream(x.device)),
          "moe_sum_scale_add")
    return out


def moe_stream_gemm_is_supported(a: torch.Tensor, w: torch.Tensor, fuse_silu_mul: bool) -> bool:
    E, N, K = w.shape
    n_out = N // 2 if fuse_silu_mul else N
    return (a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float16) and a.shape[-1] == K and a.is_contiguous()
            and w.is_contiguous() and K % 128 == 0 and n_out % 16 == 0 and (not fuse_silu_mul or N % 2 == 0))


def moe_stream_gemm(a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, topk_weights: Optional[torch.Tensor],
                    sorted_token_ids: torch.Tensor, expert_ids: torch.Tensor, num_tokens_post_pad: torch.Tensor,
                    num_valid: int, top_k_div: int, mul_routed_weight: bool, block_m: int,
                    fuse_silu_mul: bool = False) -> None:
    """invoke_fused_moe_kernel (fused_moe.py:501-612) for decode-sized calls with the LDS-DMA streaming kernel (grouped
    form of csrc/stream_linear.hip): sorted_token_ids / expert_ids from moe_align_block_size with block size block_m
    (16, 32, 48 or 64).  c[id] = a[id // top_k_div] @ w[expert].T (SiLU(gate) * up of it with fuse_silu_mul)."""
    E, N, K = w.shape
    n_out = N // 2 if fuse_silu_mul else N
    if not moe_stream_gemm_is_supported(a, w, fuse_silu_mul) or c.shape[-1] != n_out or not c.is_contiguous() \
            or block_m not in (16, 32, 48, 64) or sorted_token_ids.numel() % block_m:
        raise RuntimeError("moe_stream_gemm: shape / contiguity / block-size mismatch")
    if sorted_token_ids.dtype != torch.int32 or expert_ids.dtype != torch.int32 or num_tokens_post_pad.dtype != torch.int32:
        raise RuntimeError("moe_stream_gemm: int32 routing tensors expected")
    if mul_routed_weight and (topk_weights is None or topk_weights.dtype != torch.float32):
        raise RuntimeError("moe_stream_gemm: fp32 topk_weights required")
    check(_lib.load().semipd_moe_stream_gemm(ptr(c), ptr(a), ptr(w), ptr(topk_weights), ptr(sorted_token_ids),
                                             ptr(expert_ids), ptr(num_tokens_post_pad), num_valid, N, K,
                                             sorted_token_ids.numel(), top_k_div, int(mul_routed_weight),
                                             int(fuse_silu_mul), int(block_m), dtype_code(a.dtype),
                                             current_stream(a.device)), "moe_stream_gemm")


import os as _os

MOE_TALL_BLOCK_M = int(_os.environ.get("SEMIPD_MOE_TALL_BLOCK_M", "256"))   # 256: 256 x 256 tiles; 128: 128 x 512 tiles


def moe_gemm_tall_is_supported(a: torch.Tensor, w: torch.Tensor, fuse_silu_mul: bool) -> bool:
    E, N, K = w.shape
    n_out = N // 2 if fuse_silu_mul else N
    return (a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float16) and a.shape[-1] == K and a.is_contiguous()
            and w.is_contiguous() and K % 64 == 0 and n_out % 16 == 0 and (not fuse_silu_mul or N % 2 == 0))


def moe_gemm_tall(a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, topk_weights: Optional[torch.Tensor],
                  sorted_token_ids: torch.Tensor, expert_ids: torch.Tensor, num_tokens_post_pad: torch.Tensor,
                  num_valid: int, top_k_div: int, mul_routed_weight: bool, fuse_silu_mul: bool = False,
                  block_m: Optional[int] = None) -> None:
    """invoke_fused_moe_kernel (fused_moe.py:501-612) for prefill-sized calls with the ping-pong tile kernel
    (csrc/gemm8p.hip, grouped form; 256 x 256 tiles for block_m = 256, 128 rows x 512 columns for block_m = 128):
    sorted_token_ids / expert_ids must come from moe_align_block_size with block size block_m (default
    MOE_TALL_BLOCK_M).  c[id] = a[id // top_k_div] @ w[expert].T (SiLU(gate) * up of it with fuse_silu_mul)."""
    E, N, K = w.shape
    n_out = N // 2 if fuse_silu_mul else N
    block_m = MOE_TALL_BLOCK_M if block_m is None else int(block_m)
    if not moe_gemm_tall_is_supported(a, w, fuse_silu_mul) or c.shape[-1] != n_out or not c.is_contiguous() \
            or block_m not in (128, 256) or sorted_token_ids.numel() % block_m:
        raise RuntimeError("moe_gemm_tall: shape / contiguity / block-size mismatch")
    if sorted_token_ids.dtype != torch.int32 or expert_ids.dtype != torch.int32 or num_tokens_post_pad.dtype != torch.int32:
        raise RuntimeError("moe_gemm_tall: int32 routing tensors expected")
    if mul_routed_weight and (topk_weights is None or topk_weights.dtype != torch.float32):
        raise RuntimeError("moe_gemm_tall: fp32 topk_weights required")
    check(_lib.load().semipd_moe_gemm_tall(ptr(c), ptr(a), ptr(w), ptr(topk_weights), ptr(sorted_token_ids), ptr(expert_ids),
                                           ptr(num_tokens_post_pad), num_valid, N, K, sorted_token_ids.numel(), top_k_div,
                                           int(mul_routed_weight), int(fuse_silu_mul), block_m,
                                           dtype_code(a.dtype), current_stream(a.device)), "moe_gemm_tall")


def moe_sum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t] = x[t].sum(0) for x [T, topk, H] (fused_moe.py:1144-1148)."""
    T, k, H = x.shape
    if out is None:
        out = torch.empty((T, H), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    check(lib.semipd_moe_sum(ptr(out), ptr(x.contiguous()), T, k, H, dtype_code(x.dtype),
                             current_stream(x.device)), "moe_sum")
    return out


# --------------------------------------------------------------------------- block-scaled fp8
FP8_DTYPE = torch.float8_e4m3fn  # OCP e4m3fn: what gfx950's matrix cores take (the reference's HIP branch: e4m3fnuz)


def per_token_group_quant_fp8(x: torch.Tensor, group_size: int, eps: float = 1e-10,
                              dtype: torch.dtype = FP8_DTYPE, column_major_scales: bool = False):
    """layers/quantization/fp8_kernel.py:165-250: (x_q, x_s) with one fp32 scale per group of `group_size`
    consecutive elements of the last dimension; row-major scales only."""
    if dtype != FP8_DTYPE:
        raise RuntimeError(f"per_token_group_quant_fp8: only {FP8_DTYPE} is supported on gfx950")
    if column_major_scales:
        raise RuntimeError("per_token_group_quant_fp8: column-major scales are a DeepGEMM layout, not used here")
    if x.shape[-1] % group_size != 0:
        raise RuntimeError("the last dimension of `x` cannot be divisible by `group_size`")
    if not x.is_contiguous():
        raise RuntimeError("`x` is not contiguous")
    x_q = torch.empty(x.shape, dtype=dtype, device=x.device)
    x_s = torch.empty(x.shape[:-1] + (x.shape[-1] // group_size,), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    check(lib.semipd_per_token_group_quant_fp8(ptr(x_q), ptr(x_s), ptr(x), x.numel() // x.shape[-1], x.shape[-1],
                                               group_size, float(eps), dtype_code(x.dtype), current_stream(x.device)),
          "per_token_group_quant_fp8")
    return x_q, x_s


_F8_CODE = {torch.float8_e5m2: 3, torch.float8_e4m3fn: 4}   # SEMIPD_F8E5M2 / SEMIPD_F8E4M3
_AMAX_WS = {}


INPUT_TO_FLOAT8_WORKSPACE_BYTES = 4096   # include/semipd.h


def input_to_float8(x: torch.Tensor, dtype: torch.dtype = FP8_DTYPE) -> Tuple[torch.Tensor, torch.Tensor]:
    """Tensor-wise dynamic quantisation (layers/quantization/fp8_utils.py:137-149): (x_fp8 contiguous, 1 / scale)
    with scale = fp8_max / amax(|x|).  x: [batch, m, k] (or [m, k]) with a contiguous last dimension; a transposed
    view like q_nope.transpose(0, 1) is read through its strides."""
    if dtype not in _F8_CODE:
        raise RuntimeError(f"input_to_float8: {dtype} is not an OCP fp8 type")
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    if x3.dim() != 3 or x3.stride(2) != 1 or not x3.is_cuda:
        raise RuntimeError("input_to_float8: [batch, m, k] (or [m, k]) device tensor with a contiguous last dimension")
    B, M, K = x3.shape
    q = torch.empty((B, M, K), dtype=dtype, device=x.device)
    scale_inv = torch.empty((), dtype=torch.float32, device=x.device)
    st = current_stream(x.device)
    key = (x.device.index if x.device.index is not None else torch.cuda.current_device(), st)   # one per stream
    ws = _AMAX_WS.get(key)
    if ws is None:
        ws = _AMAX_WS[key] = torch.zeros(INPUT_TO_FLOAT8_WORKSPACE_BYTES // 4, dtype=torch.float32, device=x.device)
    check(_lib.load().semipd_input_to_float8(ptr(q), ptr(scale_inv), ptr(ws), ptr(x3), B, M, K, x3.stride(0), x3.stride(1),
                                             dtype_code(x.dtype), _F8_CODE[dtype], st), "input_to_float8")
    return (q if x.dim() == 3 else q[0]), scale_inv


def bmm_fp8(A: torch.Tensor, B: torch.Tensor, A_scale: torch.Tensor, B_scale: torch.Tensor, dtype: torch.dtype,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sgl_kernel.bmm_fp8 (python/sgl_kernel/gemm.py:66-82): A [b, m, k] fp8 row-major, B [b, k, n] fp8 COLUMN-major
    (a transposed view of a contiguous [b, n, k] tensor), per-tensor fp32 scales; returns (or fills) [b, m, n] in
    `dtype`.  `out` may be any view whose last dimension is contiguous (e.g. the [h, T, 512] transpose of
    q_input[T, h, :512])."""
    if A.dim() != 3 or B.dim() != 3 or A.shape[0] != B.shape[0] or A.shape[2] != B.shape[1]:
        raise RuntimeError("bmm_fp8: A [b, m, k] and B [b, k, n] expected")
    if A.dtype not in _F8_CODE or B.dtype not in _F8_CODE:
        raise RuntimeError("bmm_fp8: fp8 operands expected")
    if A.stride(2) != 1 or B.stride(1) != 1:
        raise RuntimeError("bmm_fp8: A must be row-major and B column-major (k contiguous in both)")
    b, m, k = A.shape
    n = B.shape[2]
    if out is None:
        out = torch.empty((b, m, n), dtype=dtype, device=A.device)
    elif out.shape != (b, m, n) or out.dtype != dtype or out.stride(2) != 1:
        raise RuntimeError("bmm_fp8: bad out tensor")
    a_s = A_scale.reshape(-1).to(torch.float32)
    b_s = B_scale.reshape(-1).to(torch.float32)
    check(_lib.load().semipd_bmm_fp8(ptr(out), ptr(A), ptr(B), ptr(a_s), ptr(b_s), b, m, n, k, A.stride(0), A.stride(1),
                                     B.stride(0), B.stride(2), out.stride(0), out.stride(1), _F8_CODE[A.dtype],
                                     _F8_CODE[B.dtype], dtype_code(dtype), current_stream(A.device)), "bmm_fp8")
    return out


def bmm_nk(x: torch.Tensor, w_nk: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = x[b] @ w_nk[b].T for decode batches of an unquantised MLA model (semipd_bmm_nk): x [b, m, k] and w_nk
    [b, n, k] with k contiguous (any batch / row strides), out [b, m, n] any view whose last dimension is contiguous.
    Stands where forward_absorb calls torch.bmm (models/deepseek_v2.py:655-667, 690-700)."""
    if x.dim() != 3 or w_nk.dim() != 3 or x.shape[0] != w_nk.shape[0] or x.shape[2] != w_nk.shape[2]:
        raise RuntimeError("bmm_nk: x [b, m, k] and w_nk [b, n, k] expected")
    if x.dtype != w_nk.dtype or x.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("bmm_nk: bf16 / f16 operands of one dtype expected")
    if x.stride(2) != 1 or w_nk.stride(2) != 1:
        raise RuntimeError("bmm_nk: k must be contiguous in both operands")
    b, m, k = x.shape
    n = w_nk.shape[1]
    if out is None:
        out = torch.empty((b, m, n), dtype=x.dtype, device=x.device)
    elif out.shape != (b, m, n) or out.dtype != x.dtype or out.stride(2) != 1:
        raise RuntimeError("bmm_nk: bad out tensor")
    check(_lib.load().semipd_bmm_nk(ptr(out), ptr(x), ptr(w_nk), b, m, n, k, x.stride(0), x.stride(1), w_nk.stride(0),
                                    w_nk.stride(1), out.stride(0), out.stride(1), dtype_code(x.dtype),
                                    current_stream(x.device)), "bmm_nk")
    return out


def fused_add_rmsnorm_quant_fp8(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                group_size: int, q_eps: float = 1e-10):
    """fused_add_rmsnorm (in place on input / residual) plus per_token_group_quant_fp8 of the normalised rows in
    the same kernel; returns (x_q, x_s).  Same bytes as the two calls."""
    if input.dim() != 2 or residual.shape != input.shape or not (input.is_contiguous() and residual.is_contiguous()):
        raise RuntimeError("fused_add_rmsnorm_quant_fp8: input / residual must be contiguous 2-D tensors of one shape")
    if weight.shape != (input.shape[1],) or not (weight.dtype == input.dtype == residual.dtype):
        raise RuntimeError("fused_add_rmsnorm_quant_fp8: weight shape / dtype mismatch")
    x_q = torch.empty(input.shape, dtype=FP8_DTYPE, device=input.device)
    x_s = torch.empty((input.shape[0], input.shape[1] // group_size), dtype=torch.float32, device=input.device)
    lib = _lib.load()
    check(lib.semipd_fused_add_rmsnorm_quant_fp8(ptr(input), ptr(residual), ptr(weight), ptr(x_q), ptr(x_s),
                                                 input.shape[0], input.shape[1], float(eps), int(group_size),
                                                 float(q_eps), dtype_code(input.dtype), current_stream(input.device)),
          "fused_add_rmsnorm_quant_fp8")
    return x_q, x_s


def rmsnorm_quant_fp8(input: torch.Tensor, weight: torch.Tensor, eps: float, group_size: int, q_eps: float = 1e-10):
    """rmsnorm(input) plus per_token_group_quant_fp8 of the result in one kernel: (out, (x_q, x_s)); input may be a view
    with a row stride (the q_a part of a merged GEMM output).  Same bytes as the two calls (semipd_rmsnorm_quant_fp8)."""
    if input.dim() != 2 or input.stride(1) != 1 or weight.shape != (input.shape[1],) or weight.dtype != input.dtype:
        raise RuntimeError("rmsnorm_quant_fp8: a 2-D input with contiguous rows and a matching weight expected")
    T, H = input.shape
    out = torch.empty((T, H), dtype=input.dtype, device=input.device)
    x_q = torch.empty((T, H), dtype=FP8_DTYPE, device=input.device)
    x_s = torch.empty((T, H // group_size), dtype=torch.float32, device=input.device)
    check(_lib.load().semipd_rmsnorm_quant_fp8(ptr(out), ptr(input), ptr(weight), ptr(x_q), ptr(x_s), T, H, input.stride(0),
                                               out.stride(0), float(eps), int(group_size), float(q_eps),
                                               dtype_code(input.dtype), current_stream(input.device)), "rmsnorm_quant_fp8")
    return out, (x_q, x_s)


def silu_and_mul_quant_fp8(x: torch.Tensor, group_size: int, eps: float = 1e-10):
    """silu_and_mul(x) followed by per_token_group_quant_fp8(., group_size) in one kernel: (x_q [..., d], x_s
    [..., d / group_size]) for x [..., 2 d]; the same bytes as the two calls (fused_moe.py:1104-1125)."""
    d = x.shape[-1] // 2
    if x.shape[-1] % 2 or d % group_size != 0:
        raise RuntimeError("the last dimension of `x` cannot be divisible by `group_size`")
    if not x.is_contiguous():
        raise RuntimeError("`x` is not contiguous")
    x_q = torch.empty(x.shape[:-1] + (d,), dtype=FP8_DTYPE, device=x.device)
    x_s = torch.empty(x.shape[:-1] + (d // group_size,), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    check(lib.semipd_silu_and_mul_quant_fp8(ptr(x_q), ptr(x_s), ptr(x), x.numel() // x.shape[-1], d, group_size,
                                            float(eps), dtype_code(x.dtype), current_stream(x.device)),
          "silu_and_mul_quant_fp8")
    return x_q, x_s


def w8a8_block_fp8_matmul(A: torch.Tensor, B: torch.Tensor, As: torch.Tensor, Bs: torch.Tensor,
                          block_size, output_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """fp8_kernel.py:694-800: A [..., K] fp8 with As [..., ceil(K/bk)], B [N, K] fp8 with Bs [ceil(N/bn), ceil(K/bk)]."""
    block_n, block_k = int(block_size[0]), int(block_size[1])
    N, K = B.shape
    if A.shape[-1] != K or A.shape[:-1] != As.shape[:-1] or not (A.is_contiguous() and B.is_contiguous()):
        raise RuntimeError("w8a8_block_fp8_matmul: shape / contiguity mismatch")
    if As.shape[-1] != -(-K // block_k) or tuple(Bs.shape) != (-(-N // block_n), -(-K // block_k)):
        raise RuntimeError("w8a8_block_fp8_matmul: scale shapes do not match the block size")
    if A.dtype != FP8_DTYPE or B.dtype != FP8_DTYPE:
        raise RuntimeError(f"w8a8_block_fp8_matmul: operands must be {FP8_DTYPE}")
    M = A.numel() // K
    C = torch.empty(A.shape[:-1] + (N,), dtype=output_dtype, device=A.device)
    lib = _lib.load()
    ws = _linear_workspace(A.device)
    check(lib.semipd_w8a8_block_fp8_matmul(ptr(C), ptr(A), ptr(As.contiguous().float()), ptr(B),
                                           ptr(Bs.contiguous().float()), M, N, K, block_n, block_k,
                                           dtype_code(output_dtype), ptr(ws), ws.numel() * ws.element_size(),
                                           current_stream(A.device)),
          "w8a8_block_fp8_matmul")
    return C


def moe_grouped_gemm_fp8(a_q: torch.Tensor, a_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, c: torch.Tensor,
                         topk_weights: Optional[torch.Tensor], sorted_token_ids: torch.Tensor,
                         expert_ids: torch.Tensor, num_tokens_post_pad: torch.Tensor, num_valid: int,
                         top_k_div: int, mul_routed_weight: bool, block_shape, block_m: int = 64) -> None:
    """invoke_fused_moe_kernel with use_fp8_w8a8 and block_shape (fused_moe.py:501-612, kernel :174-243)."""
    E, N, K = w_q.shape
    if a_q.shape[-1] != K or c.shape[-1] != N or not (a_q.is_contiguous() and w_q.is_contiguous() and c.is_contiguous()
                                                      and a_s.is_contiguous() and w_s.is_contiguous()):
        raise RuntimeError("moe_grouped_gemm_fp8: shape / contiguity mismatch")
    lib = _lib.load()
    check(lib.semipd_moe_grouped_gemm_fp8(ptr(c), ptr(a_q), ptr(a_s), ptr(w_q), ptr(w_s), ptr(topk_weights),
                                          ptr(sorted_token_ids), ptr(expert_ids), ptr(num_tokens_post_pad), num_valid,
                                          N, K, sorted_token_ids.numel(), top_k_div, int(mul_routed_weight), block_m,
                                          int(block_shape[0]), int(block_shape[1]), dtype_code(c.dtype),
                                          current_stream(c.device)), "moe_grouped_gemm_fp8")
