"""Norm / RoPE / activation / linear / logits / sampler layers of the hot path, each a thin
torch.nn.Module over the HIP operators (semi_pd_amd.ops).  Class names follow the reference:
  RMSNorm                     layers/layernorm.py:37-76
  RotaryEmbedding / get_rope  layers/rotary_embedding.py:61-169, 633-795, 993-1170
  SiluAndMul                  layers/activation.py:41-51
  Column/Row/QKV/MergedColumn parallel linear, VocabParallelEmbedding, ParallelLMHead
                              layers/linear.py:296-460, 725-1100, 1103-1280; vocab_parallel_embedding.py:174-500
  LogitsProcessor             layers/logits_processor.py:220-445
  Sampler                     layers/sampler.py:29-171 (greedy branch)
Dense GEMMs: hipBLASLt through F.linear (SURVEY §2.2 "TP linear") for prefill-sized calls, the persistent
weight-streaming kernel (ops.stream_linear) for batches of at most 128 rows, i.e. every decode step.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from semi_pd_amd import ops
from semi_pd_amd.layers.fp8 import (FP8_DTYPE, apply_w8a8_block_fp8_linear, check_quantisable_input, scale_shape,
                                    shard_rows_of_scale)
from semi_pd_amd.distributed import (all_reduce_overlap_chunks, get_tensor_model_parallel_rank,
                                     get_tensor_model_parallel_world_size, tensor_model_parallel_all_gather,
                                     tensor_model_parallel_all_reduce, tensor_model_parallel_all_reduce_async)


# --------------------------------------------------------------------------- norm / activation
class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)
        self.variance_epsilon = eps

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        if isinstance(x, ops.SplitKPlanes):
            # the row-parallel layer in front stopped before its K-slice reduction: sum, add, normalise in one launch
            return ops.fused_add_rmsnorm_planes(x, residual, self.weight.data, self.variance_epsilon), residual
        if residual is not None:
            ops.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return ops.rmsnorm(x, self.weight.data, self.variance_epsilon)

    def forward_quant(self, x: torch.Tensor, residual: torch.Tensor, group_size: int):
        """fused add + norm, and the per-token-group fp8 quantisation of the result for the block-fp8 layers that
        read it, in one kernel: (x, residual, (x_q, x_s))."""
        x_quant = ops.fused_add_rmsnorm_quant_fp8(x, residual, self.weight.data, self.variance_epsilon, group_size)
        return x, residual, x_quant


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_and_mul(x)


# --------------------------------------------------------------------------- rotary embedding
class RotaryEmbedding(nn.Module):
    """cos/sin cache [max_pos, rot] in fp32 (rotary_embedding.py:78-82 keeps fp32 on the kernel path),
    applied in place by the HIP kernel; optionally fused with the KV-pool store."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool, dtype: torch.dtype):
        super().__init__()
        self.head_size, self.rotary_dim = head_size, rotary_dim
        self.max_position_embeddings, self.base = max_position_embeddings, base
        self.is_neox_style, self.dtype = is_neox_style, dtype
        self.register_buffer("cos_sin_cache", self._compute_cos_sin_cache(), persistent=False)

    def _compute_inv_freq(self, base: float) -> torch.Tensor:
        return 1.0 / (base ** (torch.arange(0, self.rotary_dim, 2, dtype=torch.float) / self.rotary_dim))

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        inv_freq = self._compute_inv_freq(self.base)
        t = torch.arange(self.max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        ops.apply_rope_with_cos_sin_cache_inplace(positions, query, key, self.head_size, self.cos_sin_cache,
                                                  self.is_neox_style)
        return query, key

    def forward_and_store(self, positions, query, key, value, k_buffer, v_buffer, loc):
        ops.rope_and_store_kv(positions, query, key, value, self.head_size, self.cos_sin_cache,
                              self.is_neox_style, k_buffer, v_buffer, loc)
        return query, key

    def supports_planes(self) -> bool:
        return self.is_neox_style and self.rotary_dim == self.head_size and self.head_size % 16 == 0

    def forward_and_store_planes(self, positions, qkv_planes, num_q_heads, num_kv_heads, k_buffer, v_buffer, loc):
        """Decode batches: the qkv GEMM stopped before its K-slice reduction; one kernel sums the planes, rotates q and
        k and stores k / v (ops.rope_and_store_kv_planes).  Returns q."""
        return ops.rope_and_store_kv_planes(positions, qkv_planes, num_q_heads, num_kv_heads, self.head_size,
                                            self.cos_sin_cache, k_buffer, v_buffer, loc)


class Llama3RotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype,
                 scaling_factor: float, low_freq_factor: float, high_freq_factor: float, orig_max_position: int):
        self.scaling_factor, self.low_freq_factor = scaling_factor, low_freq_factor
        self.high_freq_factor, self.orig_max_position = high_freq_factor, orig_max_position
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype)

    def _compute_inv_freq(self, base: float) -> torch.Tensor:
        inv_freqs = super()._compute_inv_freq(base)
        low_freq_wavelen = self.orig_max_position / self.low_freq_factor
        high_freq_wavelen = self.orig_max_position / self.high_freq_factor
        wave_len = 2 * math.pi / inv_freqs
        if self.low_freq_factor != self.high_freq_factor:
            smooth = (self.orig_max_position / wave_len - self.low_freq_factor) / (
                self.high_freq_factor - self.low_freq_factor)
        else:
            smooth = 0
        return torch.where(
            wave_len < high_freq_wavelen, inv_freqs,
            torch.where(wave_len > low_freq_wavelen, inv_freqs / self.scaling_factor,
                        (1 - smooth) * inv_freqs / self.scaling_factor + smooth * inv_freqs))


def yarn_get_mscale(scale: float = 1, mscale: float = 1) -> float:
    if scale <= 1:
        return 1.0
    return 0.1 * mscale * math.log(scale) + 1.0


class DeepseekScalingRotaryEmbedding(RotaryEmbedding):
    """YaRN cache with mscale baked in (rotary_embedding.py:633-708)."""

    def __init__(self, head_size, rotary_dim, max_position_embeddings, base, is_neox_style, scaling_factor,
                 dtype, *, extrapolation_factor: float = 1, attn_factor: float = 1, beta_fast: int = 32,
                 beta_slow: int = 1, mscale: float = 1, mscale_all_dim: float = 0):
        self.scaling_factor, self.extrapolation_factor = scaling_factor, extrapolation_factor
        self.attn_factor, self.beta_fast, self.beta_slow = attn_factor, beta_fast, beta_slow
        self.mscale = float(yarn_get_mscale(scaling_factor, float(mscale))
                            / yarn_get_mscale(scaling_factor, float(mscale_all_dim)) * attn_factor)
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype)

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        dim, base, mp = self.rotary_dim, self.base, self.max_position_embeddings

        def corr_dim(num_rot):
            return (dim * math.log(mp / (num_rot * 2 * math.pi))) / (2 * math.log(base))

        low = max(math.floor(corr_dim(self.beta_fast)), 0)
        high = min(math.ceil(corr_dim(self.beta_slow)), dim - 1)
        if low == high:
            high += 0.001
        ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float) - low) / (high - low), 0, 1)
        pos_freqs = base ** (torch.arange(0, dim, 2, dtype=torch.float) / dim)
        inv_extra, inv_inter = 1.0 / pos_freqs, 1.0 / (self.scaling_factor * pos_freqs)
        mask = (1 - ramp) * self.extrapolation_factor
        inv_freq = inv_inter * (1 - mask) + inv_extra * mask
        t = torch.arange(mp * self.scaling_factor, dtype=torch.float32)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos() * self.mscale, freqs.sin() * self.mscale), dim=-1)


_ROPE_DICT: Dict[Tuple, RotaryEmbedding] = {}


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool = True,
             rope_scaling: Optional[Dict[str, Any]] = None, dtype: Optional[torch.dtype] = None) -> RotaryEmbedding:
    """Subset of get_rope (rotary_embedding.py:993-1170): default, llama3, deepseek_yarn."""
    dtype = dtype or torch.get_default_dtype()
    key = (head_size, rotary_dim, max_position, base, is_neox_style,
           tuple(sorted((k, str(v)) for k, v in rope_scaling.items())) if rope_scaling else None, dtype)
    if key in _ROPE_DICT:
        return _ROPE_DICT[key]
    if rope_scaling is None:
        rope = RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype)
    else:
        kind = rope_scaling.get("rope_type", rope_scaling.get("type"))
        if kind == "llama3":
            rope = Llama3RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype,
                                         rope_scaling["factor"], rope_scaling["low_freq_factor"],
                                         rope_scaling["high_freq_factor"],
                                         rope_scaling["original_max_position_embeddings"])
        elif kind == "deepseek_yarn":
            extra = {k: v for k, v in rope_scaling.items()
                     if k in ("extrapolation_factor", "attn_factor", "beta_fast", "beta_slow", "mscale",
                              "mscale_all_dim")}
            rope = DeepseekScalingRotaryEmbedding(head_size, rotary_dim,
                                                  rope_scaling["original_max_position_embeddings"], base,
                                                  is_neox_style, rope_scaling["factor"], dtype, **extra)
        elif kind in (None, "default"):
            rope = RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype)
        else:
            raise ValueError(f"Unknown RoPE scaling type {kind}")
    _ROPE_DICT[key] = rope
    return rope


# --------------------------------------------------------------------------- dense GEMM dispatch
# Batches of at most 128 rows (decode steps, a short last chunk of a prefill) go through the LDS-DMA
# weight-streaming kernel (csrc/stream_linear.hip); everything else stays on hipBLASLt (F.linear).
_STREAM_LINEAR = {"enabled": False, "timing": None}


def stream_linear_enabled() -> bool:
    return bool(_STREAM_LINEAR["enabled"])


def set_stream_linear(enabled: bool):
    """Called once per process by the model runner."""
    _STREAM_LINEAR["enabled"] = bool(enabled)


def set_stream_linear_timing(kernel_timing) -> None:
    """bench.py's roofline: a sampled (eager) decode step brackets the streaming GEMMs of its first layer with HIP
    events (model_executor/kernel_timing.py)."""
    _STREAM_LINEAR["timing"] = kernel_timing


def _timed_stream(fn, x: torch.Tensor, weight: torch.Tensor, n_out: int, n_kernels: int):
    kt = _STREAM_LINEAR["timing"]
    if kt is None or not kt.active or kt.linear_budget <= 0:
        return fn()
    kt.linear_budget -= 1
    t0 = kt.start()
    out = fn()
    es = weight.element_size()
    # SURVEY 8d "Logits + argmax"-style weight-stream formula: the weight once, activations in, result out
    kt.stop("stream_linear", t0, weight.numel() * es + x.numel() * es + x.shape[0] * n_out * es, 0.0, n_kernels)
    return out


def _timed_gemm(fn, x: torch.Tensor, weight: torch.Tensor, n_out: int):
    """The four dense layers of the first decoder layer of a sampled PREFILL batch (bench.py's roofline_extra.prefill_gemm):
    whatever serves them -- a library solution timed on the share, the tiled GEMM, with or without the SiLU epilogue --
    bracketed like the streaming GEMMs of a decode step; flops = 2 rows n k."""
    kt = _STREAM_LINEAR["timing"]
    if kt is None or not kt.active or kt.linear_budget <= 0 or x.dim() != 2 or x.shape[0] <= GEMM_TALL_MAX_ROWS:
        return fn()
    kt.linear_budget -= 1
    t0 = kt.start()
    out = fn()
    es = weight.element_size()
    kt.stop("prefill_gemm", t0, (weight.numel() + x.numel() + x.shape[0] * n_out) * es,
            2.0 * x.shape[0] * weight.shape[0] * weight.shape[1], 1)
    return out


GEMM_TALL_MAX_ROWS = 256   # above: the library GEMM (prefill-sized; its solution timed on the share where that was asked for)


def dense_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UnquantizedLinearMethod.apply (layers/linear.py:165-172).  `out`: a [rows, n] view with unit inner stride to write
    into (the token chunks of RowParallelLinear._forward_overlapped)."""
    if (_STREAM_LINEAR["enabled"] and bias is None and x.dim() == 2 and x.shape[0] <= ops.STREAM_LINEAR_DENSE_MAX_ROWS
            and ops.stream_linear_is_supported(x, weight)):
        return _timed_stream(lambda: ops.stream_linear(x, weight, out=out), x, weight, weight.shape[0], 2)
    if _STREAM_LINEAR["enabled"] and bias is None and x.dim() == 2 and _takes_tiled_gemm(x, weight):
        # the tiled ping-pong GEMM (csrc/gemm8p.hip): a tall decode batch (129 .. 256 rows) of an untuned layer, or a
        # prefill-sized batch of a shape for which it beat the library's measured winner on this share
        return _timed_gemm(lambda: ops.gemm_tall(x, weight, out=out), x, weight, weight.shape[0])
    if x.dim() == 2 and x.shape[0] > 0 and ops.dense_gemm_is_tuned(weight) and x.stride(1) == 1:
        # prefill-sized batch of a layer whose library solutions were timed on this process's CU share at start-up
        # (ModelRunner.tune_dense_gemms): the measured winner instead of the library's whole-device heuristic
        return _timed_gemm(lambda: ops.dense_gemm(x, weight, bias, out=out), x, weight, weight.shape[0])
    if out is None:
        return F.linear(x, weight, bias)
    # one rounding, like F.linear (mm + a separate add would round twice)
    if bias is not None:
        return torch.addmm(bias, x, weight.t(), out=out)
    return torch.mm(x, weight.t(), out=out)


# RowParallelLinear.forward(defer_reduce=True) above the streaming kernel's rows: SEMIPD_TALL_PLANES=1 hands the tiled GEMM's
# K-slice planes to the norm.  OFF by default: alone the pair is 4-12 us shorter per layer, next to a running decode instance
# the serving run is slower with it (six alternating runs on two boxes, profiles/r05_tall_planes_in_situ_ab.txt: prefill batch
# 27.2-27.6 -> 27.8-28.9 ms, TTFT p50 36.5-37.9 -> 38.1-39.1, TBT p50 4.9-5.1 -> 5.5)
_TALL_PLANES = os.environ.get("SEMIPD_TALL_PLANES", "0") == "1"


def _takes_tiled_gemm(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """dense_linear's two conditions for ops.gemm_tall, bias-free call."""
    rows = x.shape[0]
    if ops.STREAM_LINEAR_DENSE_MAX_ROWS < rows <= GEMM_TALL_MAX_ROWS:
        return not ops.dense_gemm_is_tuned(weight) and ops.gemm_tall_is_supported(x, weight)
    return rows > GEMM_TALL_MAX_ROWS and ops.tall_preferred(weight, rows) and ops.gemm_tall_is_supported(x, weight)


def gate_up_silu(x: torch.Tensor, gate_up_proj: "MergedColumnParallelLinear", act_fn) -> torch.Tensor:
    """act_fn(gate_up_proj(x)) (models/llama.py:88-92); one launch for decode batches of a bf16 / f16 layer."""
    if (_STREAM_LINEAR["enabled"] and gate_up_proj.quant_config is None and gate_up_proj.bias is None
            and isinstance(act_fn, SiluAndMul) and x.dim() == 2 and x.shape[0] <= ops.STREAM_LINEAR_DENSE_MAX_ROWS
            and ops.stream_linear_is_supported(x, gate_up_proj.weight, fuse_silu_mul=True)):
        w = gate_up_proj.weight
        return _timed_stream(lambda: ops.stream_linear(x, w, fuse_silu_mul=True), x, w, w.shape[0] // 2, 1)
    if (_STREAM_LINEAR["enabled"] and gate_up_proj.quant_config is None and gate_up_proj.bias is None
            and isinstance(act_fn, SiluAndMul) and x.dim() == 2
            and ops.STREAM_LINEAR_DENSE_MAX_ROWS < x.shape[0] <= GEMM_TALL_MAX_ROWS
            and not ops.dense_gemm_is_tuned(gate_up_proj.weight)
            and ops.gemm_tall_is_supported(x, gate_up_proj.weight, fuse_silu_mul=True)):
        return ops.gemm_tall(x, gate_up_proj.weight, fuse_silu_mul=True)   # SiLU * mul in the GEMM's epilogue
    if (_STREAM_LINEAR["enabled"] and gate_up_proj.quant_config is None and gate_up_proj.bias is None
            and isinstance(act_fn, SiluAndMul) and x.dim() == 2 and x.shape[0] > GEMM_TALL_MAX_ROWS
            and ops.tall_preferred(gate_up_proj.weight, x.shape[0], fuse_silu_mul=True)
            and ops.gemm_tall_is_supported(x, gate_up_proj.weight, fuse_silu_mul=True)):
        # prefill-sized batch: the tiled GEMM with the SiLU epilogue beat the library's winner + silu_and_mul on this share
        w = gate_up_proj.weight
        return _timed_gemm(lambda: ops.gemm_tall(x, w, fuse_silu_mul=True), x, w, w.shape[0] // 2)
    return act_fn(gate_up_proj(x))


# --------------------------------------------------------------------------- tensor-parallel linear
def _shard(n: int, tp: int) -> int:
    assert n % tp == 0, f"{n} is not divisible by tp={tp}"
    return n // tp


class ColumnParallelLinear(nn.Module):
    """Y = X W^T with W [out/tp, in] (layers/linear.py:296-460)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False, params_dtype=None, quant_config=None):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.output_size_per_partition = _shard(output_size, tp)
        self.quant_config = quant_config
        self.weight = nn.Parameter(torch.empty(self.output_size_per_partition, input_size,
                                               dtype=FP8_DTYPE if quant_config else params_dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(self.output_size_per_partition, dtype=params_dtype),
                                 requires_grad=False) if bias else None
        if quant_config:
            # Fp8LinearMethod.create_weights (quantization/fp8.py:205-330): one fp32 scale per weight block
            check_quantisable_input(input_size, quant_config.weight_block_size, type(self).__name__)
            self.weight_scale_inv = nn.Parameter(torch.empty(
                scale_shape(self.output_size_per_partition, input_size, quant_config.weight_block_size),
                dtype=torch.float32), requires_grad=False)
            self.weight.weight_block_size = quant_config.weight_block_size
        self._tag_shards(input_size, output_size)

    def _row_ranges(self, output_size: int):
        """Row ranges of the full [out, in] weight owned by this rank (weight_loader semantics,
        layers/linear.py:350-400)."""
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        n = output_size // tp
        return [(rank * n, (rank + 1) * n)]

    def _tag_shards(self, input_size: int, output_size: int):
        ranges = self._row_ranges(output_size)
        self.weight.tp_full_shape = (output_size, input_size)
        self.weight.tp_shard = lambda full: torch.cat([full[a:b] for a, b in ranges], 0)
        if self.quant_config:
            block = self.quant_config.weight_block_size
            self.weight_scale_inv.tp_full_shape = scale_shape(output_size, input_size, block)
            if get_tensor_model_parallel_world_size() > 1:
                s_ranges = shard_rows_of_scale(ranges, block[0])
                self.weight_scale_inv.tp_shard = lambda full: torch.cat([full[a:b] for a, b in s_ranges], 0)
            else:
                self.weight_scale_inv.tp_shard = lambda full: full
        if self.bias is not None:
            self.bias.tp_full_shape = (output_size,)
            self.bias.tp_shard = lambda full: torch.cat([full[a:b] for a, b in ranges], 0)

    def forward(self, x, x_quant=None):
        if self.quant_config:
            return apply_w8a8_block_fp8_linear(x, self.weight, self.quant_config.weight_block_size,
                                               self.weight_scale_inv, self.bias, x_quant=x_quant)
        return dense_linear(x, self.weight, self.bias)

    def forward_planes(self, x):
        """Decode batches of an unquantised, bias-free layer: the K-slice planes of the streaming GEMM (ops.SplitKPlanes)
        for a consumer that sums them itself, or None when this call is not eligible."""
        if (_STREAM_LINEAR["enabled"] and self.quant_config is None and self.bias is None and x.dim() == 2
                and x.shape[0] <= ops.STREAM_LINEAR_DENSE_MAX_ROWS and self.weight.shape[0] % 8 == 0
                and ops.stream_linear_is_supported(x, self.weight)):
            return _timed_stream(lambda: ops.stream_linear_planes(x, self.weight), x, self.weight, self.weight.shape[0], 1)
        return None


class ReplicatedLinear(ColumnParallelLinear):
    """Every rank holds the whole weight (layers/linear.py:170-293; q_a_proj, kv_a_proj_with_mqa)."""

    def _row_ranges(self, output_size: int):
        return [(0, output_size)]

    def __init__(self, input_size: int, output_size: int, bias: bool = False, params_dtype=None, quant_config=None):
        nn.Module.__init__(self)
        self.output_size_per_partition = output_size
        self.quant_config = quant_config
        self.weight = nn.Parameter(torch.empty(output_size, input_size, dtype=FP8_DTYPE if quant_config else params_dtype),
                                   requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(output_size, dtype=params_dtype), requires_grad=False) if bias else None
        if quant_config:
            check_quantisable_input(input_size, quant_config.weight_block_size, type(self).__name__)
            self.weight_scale_inv = nn.Parameter(torch.empty(
                scale_shape(output_size, input_size, quant_config.weight_block_size), dtype=torch.float32),
                requires_grad=False)
            self.weight.weight_block_size = quant_config.weight_block_size


class MergedColumnParallelLinear(ColumnParallelLinear):
    """gate_up_proj: the per-rank shard is [gate/tp ; up/tp] (linear.py:463-722)."""

    def __init__(self, input_size: int, output_sizes, bias: bool = False, params_dtype=None, quant_config=None):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias, params_dtype, quant_config)

    def _row_ranges(self, output_size: int):
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        out, base = [], 0
        for size in self.output_sizes:
            n = size // tp
            out.append((base + rank * n, base + (rank + 1) * n))
            base += size
        return out


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection; kv heads are replicated when Hkv < tp (linear.py:725-1100,
    models/llama.py:115-131)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: int,
                 bias: bool = False, params_dtype=None, quant_config=None):
        tp = get_tensor_model_parallel_world_size()
        self.head_size = head_size
        self.num_heads = _shard(total_num_heads, tp)
        if total_num_kv_heads >= tp:
            self.num_kv_heads = _shard(total_num_kv_heads, tp)
            self.num_kv_head_replicas = 1
        else:
            self.num_kv_heads = 1
            self.num_kv_head_replicas = _shard(tp, total_num_kv_heads)
        self.total_num_heads, self.total_num_kv_heads = total_num_heads, total_num_kv_heads
        out = (self.num_heads + 2 * self.num_kv_heads) * tp * head_size
        super().__init__(hidden_size, out, bias, params_dtype, quant_config)

    def _row_ranges(self, output_size: int):
        # full layout: [q heads | k heads | v heads] (linear.py:880-960 shard ids "q","k","v")
        rank = get_tensor_model_parallel_rank()
        d = self.head_size
        q0 = rank * self.num_heads * d
        kv_rank = rank // self.num_kv_head_replicas
        k_base = self.total_num_heads * d
        v_base = k_base + self.total_num_kv_heads * d
        kn = self.num_kv_heads * d
        return [(q0, q0 + self.num_heads * d), (k_base + kv_rank * kn, k_base + (kv_rank + 1) * kn),
                (v_base + kv_rank * kn, v_base + (kv_rank + 1) * kn)]

    def _tag_shards(self, input_size: int, output_size: int):
        full_out = (self.total_num_heads + 2 * self.total_num_kv_heads) * self.head_size
        super()._tag_shards(input_size, full_out)


class RowParallelLinear(nn.Module):
    """Y = all_reduce(X_shard W_shard^T) with W [out, in/tp] (layers/linear.py:1103-1280, reduce :1266)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False, reduce_results: bool = True,
                 params_dtype=None, quant_config=None):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.input_size_per_partition = _shard(input_size, tp)
        self.reduce_results = reduce_results
        self.quant_config = quant_config
        self.weight = nn.Parameter(torch.empty(output_size, self.input_size_per_partition,
                                               dtype=FP8_DTYPE if quant_config else params_dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(output_size, dtype=params_dtype), requires_grad=False) if bias else None
        rank, n = get_tensor_model_parallel_rank(), self.input_size_per_partition
        self.weight.tp_full_shape = (output_size, input_size)
        self.weight.tp_shard = lambda full: full[:, rank * n:(rank + 1) * n].contiguous()
        if quant_config:
            block = quant_config.weight_block_size
            check_quantisable_input(n, block, type(self).__name__)
            self.weight.weight_block_size = block
            self.weight_scale_inv = nn.Parameter(torch.empty(scale_shape(output_size, n, block), dtype=torch.float32),
                                                 requires_grad=False)
            self.weight_scale_inv.tp_full_shape = scale_shape(output_size, input_size, block)
            if tp > 1:
                if n % block[1]:
                    raise ValueError(f"input partition {n} is not a multiple of the weight block width {block[1]}")
                nb = n // block[1]
                self.weight_scale_inv.tp_shard = lambda full: full[:, rank * nb:(rank + 1) * nb].contiguous()
            else:
                self.weight_scale_inv.tp_shard = lambda full: full

    def forward(self, x, defer_reduce: bool = False):
        """defer_reduce: the caller hands the result straight to RMSNorm.forward(x, residual); for a decode batch
        on one GPU the layer then returns the K-slice planes of the streaming GEMM (ops.SplitKPlanes) and the norm
        kernel does the reduction."""
        if (defer_reduce and _STREAM_LINEAR["enabled"] and self.quant_config is None and self.bias is None
                and get_tensor_model_parallel_world_size() == 1 and x.dim() == 2
                and x.shape[0] <= ops.STREAM_LINEAR_DENSE_MAX_ROWS and self.weight.shape[0] % 8 == 0
                and ops.stream_linear_is_supported(x, self.weight)):
            return _timed_stream(lambda: ops.stream_linear_planes(x, self.weight), x, self.weight, self.weight.shape[0], 1)
        if (defer_reduce and _TALL_PLANES and _STREAM_LINEAR["enabled"] and self.quant_config is None and self.bias is None
                and get_tensor_model_parallel_world_size() == 1 and x.dim() == 2
                and _takes_tiled_gemm(x, self.weight)):
            # taller batches where dense_linear would take the tiled GEMM (the same choice, the same kernel): its K-slice
            # planes go to the norm as they are -- no reduction launch, no round trip of the [rows, n] result
            return ops.gemm_tall_planes(x, self.weight)
        # bias is added on rank 0 only so that the sum over ranks adds it once (linear.py:1258-1262)
        bias = self.bias if (self.bias is not None and get_tensor_model_parallel_rank() == 0) else None
        if self.quant_config:
            out = apply_w8a8_block_fp8_linear(x, self.weight, self.quant_config.weight_block_size,
                                              self.weight_scale_inv, bias)
        else:
            chunks = all_reduce_overlap_chunks(x.shape[0]) if (self.reduce_results and x.dim() == 2) else 1
            if chunks > 1:
                return self._forward_overlapped(x, bias, chunks)
            out = dense_linear(x, self.weight, bias)
        if self.reduce_results and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out

    def _forward_overlapped(self, x: torch.Tensor, bias: Optional[torch.Tensor], chunks: int) -> torch.Tensor:
        """Prefill-sized call under tensor parallelism: the all-reduce of token chunk i (communication stream) runs
        while the GEMM of chunk i + 1 runs (compute stream).  The reduce is element-wise, so chunking it changes no bit;
        every chunk goes through dense_linear like the blocking form (round 5: it used torch.mm, which bypassed the GEMMs
        chosen on the instance's CU share exactly where they matter, at >= 1024 tokens).  A GEMM whose algorithm depends
        on the row count may round a chunk differently from the whole batch -- as any two batch sizes may."""
        T = x.shape[0]
        out = torch.empty((T, self.weight.shape[0]), dtype=x.dtype, device=x.device)
        step = -(-T // chunks)
        step = -(-step // 16) * 16
        pending = []
        for a in range(0, T, step):
            o = out[a:a + step]
            # the same GEMM choice as the blocking form makes for rows of this count (share-tuned library solution,
            # tiled kernel, ...): dense_linear, writing into the chunk's rows of `out`
            dense_linear(x[a:a + step], self.weight, bias, out=o)
            pending.append(tensor_model_parallel_all_reduce_async(o))
        for p in pending:
            p.wait()
        return out

    def forward_prequantized(self, x_q: torch.Tensor, x_s: torch.Tensor, out_dtype: torch.dtype):
        """The block-fp8 layer on activations that are quantised already (by the kernel that produced them, e.g.
        ops.silu_and_mul_quant_fp8): skips apply_w8a8_block_fp8_linear's own quantisation pass."""
        out = ops.w8a8_block_fp8_matmul(x_q, self.weight, x_s, self.weight_scale_inv, self.quant_config.weight_block_size,
                                        output_dtype=out_dtype)
        if self.bias is not None and get_tensor_model_parallel_rank() == 0:
            out = out + self.bias
        if self.reduce_results and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class VocabParallelEmbedding(nn.Module):
    """Embedding rows sharded over ranks; out-of-shard ids contribute zeros, then all-reduce
    (layers/vocab_parallel_embedding.py:174-500, reduce :487)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, params_dtype=None, pad_to: int = 64):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        rank = get_tensor_model_parallel_rank()
        self.org_vocab_size = num_embeddings
        padded = -(-num_embeddings // (pad_to * tp)) * (pad_to * tp)
        self.num_embeddings_padded = padded
        self.num_embeddings_per_partition = padded // tp
        self.vocab_start_index = rank * self.num_embeddings_per_partition
        self.vocab_end_index = self.vocab_start_index + self.num_embeddings_per_partition
        self.tp_size = tp
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim,
                                               dtype=params_dtype), requires_grad=False)
        a, b, org = self.vocab_start_index, self.vocab_end_index, num_embeddings
        self.weight.tp_full_shape = (num_embeddings, embedding_dim)

        def shard(full):  # rows of the real vocabulary, zero rows for the padding
            out = torch.zeros(b - a, full.shape[1], dtype=full.dtype, device=full.device)
            hi = min(b, org)
            if hi > a:
                out[: hi - a] = full[a:hi]
            return out

        self.weight.tp_shard = shard

    def forward(self, input_ids: torch.Tensor):
        if self.tp_size == 1:
            return F.embedding(input_ids, self.weight)
        mask = (input_ids >= self.vocab_start_index) & (input_ids < self.vocab_end_index)
        local = torch.where(mask, input_ids - self.vocab_start_index, torch.zeros_like(input_ids))
        out = F.embedding(local, self.weight)
        out = out * mask.unsqueeze(-1).to(out.dtype)
        return tensor_model_parallel_all_reduce(out)


class ParallelLMHead(VocabParallelEmbedding):
    pass


# --------------------------------------------------------------------------- logits / sampler
@dataclass
class LogitsProcessorOutput:
    next_token_logits: Optional[torch.Tensor]
    next_token_ids: Optional[torch.Tensor] = None
    hidden_states: Optional[torch.Tensor] = None
    # filled by the sampler when a request asked for logprobs (logits_processor.py:40-70)
    next_token_logprobs: Optional[torch.Tensor] = None            # fp32 [B]
    next_token_top_logprobs_val: Optional[list] = None            # per row: k values (descending)
    next_token_top_logprobs_idx: Optional[list] = None            # per row: k token ids


LM_HEAD_FUSED_MAX_ROWS = int(os.environ.get("SEMIPD_LM_HEAD_FUSED_MAX_ROWS", "64"))


class LogitsProcessor(nn.Module):
    """Last-token gather + lm_head + all-gather + fp32 (logits_processor.py:220-445).  With TP=1 and a
    greedy batch the argmax is produced by the same HIP call (ops.lm_head_argmax); with TP>1 the
    local shard of the logits is computed, gathered and then reduced with ops.greedy_argmax."""

    def __init__(self, vocab_size: int, final_logit_softcapping: Optional[float] = None):
        super().__init__()
        self.vocab_size = vocab_size
        self.final_logit_softcapping = final_logit_softcapping

    def forward(self, input_ids, hidden_states: torch.Tensor, lm_head: VocabParallelEmbedding,
                forward_batch) -> LogitsProcessorOutput:
        if forward_batch.forward_mode.is_extend():
            last_index = torch.cumsum(forward_batch.extend_seq_lens, dim=0, dtype=torch.int64) - 1
            pruned = ops.gather_rows(hidden_states, last_index)
        else:
            pruned = hidden_states
        # The fused lm_head + argmax kernel walks the weights once per 64 rows: at 256 rows it takes 1.18 ms where the
        # library GEMM + argmax takes 0.36 (profiles/r02_kbench_all_final_tree.txt), at <= 64 rows it is at par or ahead
        # and returns fp32 logits.  Above that the reference's own form: bf16 GEMM, then fp32 (logits_processor.py:394-445).
        if get_tensor_model_parallel_world_size() == 1 and self.final_logit_softcapping is None \
                and pruned.shape[0] <= LM_HEAD_FUSED_MAX_ROWS:
            # rows >= vocab_size are padding (vocab_parallel_embedding.py pads to a multiple of 64)
            logits, ids = ops.lm_head_argmax(pruned.contiguous(), lm_head.weight[: self.vocab_size],
                                             return_logits=True)
            return LogitsProcessorOutput(logits, next_token_ids=ids)
        if (_STREAM_LINEAR["enabled"] and pruned.dim() == 2 and pruned.shape[0] > LM_HEAD_FUSED_MAX_ROWS
                and ops.gemm_tall_is_supported(pruned, lm_head.weight)):
            # more than 64 rows: the tiled ping-pong GEMM reads the vocabulary-sized weight once (128 256 x 4096 at 256
            # rows: 268 us against the library's 307 on the whole chip, 494 against 552 on a 96-CU share); logits in the
            # activation type like the reference's matmul (logits_processor.py:394-445)
            logits = ops.gemm_tall(pruned, lm_head.weight)
        elif (_STREAM_LINEAR["enabled"] and pruned.dim() == 2 and pruned.shape[0] <= ops.STREAM_LINEAR_MAX_ROWS
              and ops.stream_linear_is_supported(pruned.contiguous(), lm_head.weight)):
            # at most 64 rows that the fused lm_head + argmax call above did not take (tensor parallelism: every rank holds
            # a slice of the vocabulary; soft-capped logits): the weight-streaming kernel on the rank's rows, logits in the
            # activation type like the reference's matmul
            logits = ops.stream_linear(pruned.contiguous(), lm_head.weight)
        else:
            logits = torch.matmul(pruned, lm_head.weight.T)
        logits = tensor_model_parallel_all_gather(logits)
        logits = logits[:, : self.vocab_size].float()
        if self.final_logit_softcapping:
            logits = self.final_logit_softcapping * torch.tanh(logits / self.final_logit_softcapping)
        return LogitsProcessorOutput(logits)


class Sampler(nn.Module):
    """layers/sampler.py:38-171.  Greedy batches: argmax over fp32 logits -> int32 ids (:72-74; already
    produced by the fused lm_head call when it ran).  Otherwise logits / T -> softmax -> joint
    top-k / top-p rejection sampling, or top-k renorm -> top-p renorm -> min-p sampling when a request
    asks for min_p (:77-107), all in HIP (csrc/sampling.hip)."""

    MAX_TOP_K_ROUND = 32  # sampler.py:92

    def forward(self, logits_output: LogitsProcessorOutput, sampling_info=None, return_logprob: bool = False,
                top_logprobs_nums: Optional[list] = None) -> torch.Tensor:
        penalised = sampling_info is not None and getattr(sampling_info, "has_penalties", False)
        if penalised:
            # sampling_batch_info.py:188-191: penalties change the fp32 logits before anything else looks at them;
            # ids the fused lm_head call produced from the raw logits no longer hold
            logits = logits_output.next_token_logits
            if logits is None:
                raise RuntimeError("Sampler: penalties need the fp32 logits")
            if logits.dtype != torch.float32 or not logits.is_contiguous():
                logits = logits_output.next_token_logits = logits.float().contiguous()
            sampling_info.apply_penalties(logits)
        if sampling_info is None or getattr(sampling_info, "is_all_greedy", True):
            logits = logits_output.next_token_logits
            if logits_output.next_token_ids is not None and not penalised:
                ids = logits_output.next_token_ids
            else:
                if not logits.is_contiguous():
                    logits = logits.contiguous()
                ids = ops.greedy_argmax(logits)
            if return_logprob:
                # logprobs = log_softmax(logits) (sampler.py:74-75); only the gathered values and the
                # top-k rows are materialised
                if logits is None:
                    raise RuntimeError("Sampler: return_logprob needs the fp32 logits")
                logits = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else \
                    logits.float().contiguous()
                token_lp, lse = ops.token_logprobs(logits, ids)

                def top_logprobs(k):
                    vals, idx = torch.topk(logits, k, dim=-1)
                    return vals - lse[:, None], idx

                self._attach_logprobs(logits_output, token_lp, top_logprobs_nums, top_logprobs)
            return ids
        logits = logits_output.next_token_logits
        if logits is None:
            raise RuntimeError("Sampler: stochastic sampling needs the fp32 logits")
        if logits.dtype != torch.float32 or not logits.is_contiguous():
            logits = logits.float().contiguous()
        batch = logits.shape[0]
        if len(sampling_info) != batch:
            raise RuntimeError(f"Sampler: {len(sampling_info)} sampling params for {batch} logit rows")
        probs = ops.softmax_temperature_(logits, sampling_info.temperatures)  # in place, like the reference
        uniform_samples = torch.rand((self.MAX_TOP_K_ROUND, batch), device=probs.device)
        if sampling_info.need_min_p_sampling:
            filtered = ops.top_k_renorm_prob(probs, sampling_info.top_ks)
            filtered = ops.top_p_renorm_prob(filtered, sampling_info.top_ps)
            ids = ops.min_p_sampling_from_probs(filtered, uniform_samples, sampling_info.min_ps)
        else:
            ids, _ = ops.top_k_top_p_sampling_from_probs(probs, uniform_samples, sampling_info.top_ks,
                                                         sampling_info.top_ps, filter_apply_order="joint")
        if return_logprob:
            # logprobs = log(top-p-normalised probs), clamped away from -inf (sampler.py:84-89, 121-125)
            lp = torch.log(ops.top_p_renorm_prob(probs, sampling_info.top_ps)).clamp_(
                min=torch.finfo(torch.float32).min)
            token_lp = lp.gather(1, ids.long().view(-1, 1)).view(-1)
            self._attach_logprobs(logits_output, token_lp, top_logprobs_nums, lambda k: torch.topk(lp, k, dim=-1))
        return ids

    @staticmethod
    def _attach_logprobs(out: LogitsProcessorOutput, token_lp: torch.Tensor, top_nums, topk_fn):
        """sampler.py:139-155 + get_top_logprobs (:246-262): one top-k of the largest k requested."""
        out.next_token_logprobs = token_lp
        kmax = max(top_nums) if top_nums else 0
        if kmax > 0:
            vals, idx = topk_fn(kmax)
            vals, idx = vals.tolist(), idx.tolist()
            out.next_token_top_logprobs_val = [v[:k] for v, k in zip(vals, top_nums)]
            out.next_token_top_logprobs_idx = [i[:k] for i, k in zip(idx, top_nums)]
