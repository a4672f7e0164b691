"""MoE routing + fused experts on the HIP kernels.

Reference: layers/moe/topk.py:162-220 (`select_experts`), layers/moe/fused_moe_triton/fused_moe.py
:409-498 (`moe_align_block_size`), :961-1165 (`fused_experts_impl`: align -> GEMM1 -> SiLU*mul ->
GEMM2 x routed weight -> sum over top-k), layers/moe/fused_moe_triton/layer.py:239-641 (`FusedMoE`,
weights w13 [E, 2N, K] and w2 [E, K, N], TP all-reduce :637-638).
"""
from __future__ import annotations

import os

from typing import Optional

import torch
from torch import nn

from semi_pd_amd import ops
from semi_pd_amd.layers.fp8 import FP8_DTYPE, check_quantisable_input, scale_shape
from semi_pd_amd.distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                                     tensor_model_parallel_all_reduce)

MOE_BLOCK_M = 64  # rows per expert block = M tile of the grouped GEMM

# --enable-ep-moe (server_args.py; read through global_server_args_dict in models/deepseek_v2.py:190-200): the
# routed experts are partitioned over the TP ranks by EXPERT (E / tp whole experts per rank, layers/moe/ep_moe/
# layer.py:106-190) instead of by intermediate column.  Set by the model runner before the model is built.
_EXPERT_PARALLEL = False


def set_expert_parallel(enabled: bool) -> None:
    global _EXPERT_PARALLEL
    _EXPERT_PARALLEL = bool(enabled)


def expert_parallel_enabled() -> bool:
    return _EXPERT_PARALLEL


# --enable-ep-all-to-all (with --enable-ep-moe; no counterpart in the reference, whose expert parallelism keeps every token
# on every rank and all-reduces the partial outputs, ep_moe/layer.py:190; SURVEY 8f-4 / BASELINE config 5): every rank
# routes only its own slice of the tokens, the rows travel to the ranks that own their experts and back through the
# peer-memory all-to-all (csrc/all_reduce.hip: semipd_ep_dispatch / semipd_ep_combine), the slices are all-gathered.
_EP_ALL_TO_ALL = False
EP_REGION_CAPACITY = int(os.environ.get("SEMIPD_EP_REGION_MB", "128")) << 20   # payload bytes of a peer-memory slot


def set_expert_all_to_all(enabled: bool) -> None:
    global _EP_ALL_TO_ALL
    _EP_ALL_TO_ALL = bool(enabled)


def expert_all_to_all_enabled() -> bool:
    return _EP_ALL_TO_ALL


def select_experts(hidden_states: torch.Tensor, router_logits: torch.Tensor, top_k: int,
                   use_grouped_topk: bool, renormalize: bool, topk_group: Optional[int] = None,
                   num_expert_group: Optional[int] = None, correction_bias: Optional[torch.Tensor] = None):
    """topk.py:162-220: grouped_topk / biased_grouped_topk when use_grouped_topk, else fused_topk."""
    if use_grouped_topk:
        assert topk_group is not None and num_expert_group is not None
        return ops.grouped_topk(router_logits, top_k, renormalize, num_expert_group, topk_group, correction_bias)
    return ops.topk_softmax(router_logits, top_k, renormalize)


# rows (tokens x top-k) from which the expert GEMMs take the 256-row-block tiled kernel, and the least average rows per
# expert (below that the padding to 256 per expert costs more than the faster tile buys).  DeepSeek-V2-Lite experts,
# whole fused MoE: T = 2048 410 -> 362 us, 4096 702 -> 611, 8192 1158 -> 1032 (0.73 -> 0.82 PFLOP/s)
# (profiles/r03_kbench_moe_tall_blocks.txt)
MOE_TALL_MIN_ROWS = int(os.environ.get("SEMIPD_MOE_TALL_MIN_ROWS", "4096"))
MOE_TALL_MIN_ROWS_PER_EXPERT = int(os.environ.get("SEMIPD_MOE_TALL_MIN_ROWS_PER_EXPERT", "128"))
# below those bounds, from this many rows per expert up: the 128-row x 512-column geometry of the same kernel (most experts
# are one block; the 256-row geometry takes over where most would be two).  One prefill request of DeepSeek-V2-Lite on
# the 128-CU share: T = 512 338 -> 306 us, 1024 381 -> 331, 1280 ~470 -> 417; from 1408 tokens the 256-row geometry
# (632 -> 463 at 1536) (profiles/r03_kbench_moe_mid_geometry.txt)
MOE_MID_MIN_ROWS_PER_EXPERT = int(os.environ.get("SEMIPD_MOE_MID_MIN_ROWS_PER_EXPERT", "40"))
MOE_STREAM_DECODE = os.environ.get("SEMIPD_MOE_STREAM_DECODE", "1") != "0"   # A/B knob: 0 = the register-fragment kernel
MOE_STREAM_MAX_TOKENS = int(os.environ.get("SEMIPD_MOE_STREAM_MAX_TOKENS", "320"))


def _local_ids(topk_ids: torch.Tensor, expert_offset: int) -> torch.Tensor:
    """Expert-parallel ranks hold experts [offset, offset + E_local): ids are shifted so that the local ones land
    in [0, E_local); the others fall outside and moe_align_block_size drops them (their rows of the output stay
    zero, the all-reduce over the ranks adds the other ranks' experts; ep_moe/layer.py:211-246, 340-360)."""
    return topk_ids if expert_offset == 0 else (topk_ids - expert_offset).to(torch.int32)


def _finish(c3: torch.Tensor, out_scale: float, out_addend: Optional[torch.Tensor]) -> torch.Tensor:
    """The sum over the top-k rows; with a scale and / or an addend (DeepseekV2MoE's `* routed_scaling_factor` and
    `+ shared_output`) continued in the same launch with the roundings of the separate ones."""
    if out_scale == 1.0 and out_addend is None:
        return ops.moe_sum(c3)
    return ops.moe_sum_scale_add(c3, out_scale, out_addend)


def fused_experts(hidden_states: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, topk_weights: torch.Tensor,
                  topk_ids: torch.Tensor, expert_offset: int = 0, partial_experts: bool = False,
                  out_scale: float = 1.0, out_addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fused_experts_impl (fused_moe.py:961-1165), bf16/f16 path.  hidden [T, K]; w1 [E, 2N, K];
    w2 [E, K, N]; returns [T, K].  partial_experts: w1 / w2 hold only the experts [expert_offset, expert_offset + E)."""
    topk_ids = _local_ids(topk_ids, expert_offset)
    T, K = hidden_states.shape
    E, N2, _ = w1.shape
    if T == 0:   # an empty batch (an idle DP / EP rank): nothing to route, and no block size to divide by
        return out_addend if out_addend is not None else hidden_states.new_zeros((0, K))
    topk = topk_ids.shape[1]
    dev, dt = hidden_states.device, hidden_states.dtype
    numel = T * topk
    # rows per block: every block streams its expert's weights once, so prefill chunks use the taller
    # block (fused_moe.py:614-700 get_default_config picks BLOCK_SIZE_M by M the same way)
    # prefill-sized calls with enough rows per expert: 256-row blocks and the tiled ping-pong GEMM (csrc/gemm8p.hip, grouped
    # form), SiLU * mul in GEMM1's epilogue; every expert's rows are padded to a multiple of 256, which is why it only
    # pays from a few hundred rows per expert up
    tall = (numel >= MOE_TALL_MIN_ROWS and numel >= MOE_TALL_MIN_ROWS_PER_EXPERT * E
            and ops.moe_gemm_tall_is_supported(hidden_states, w1, True) and w2.shape[2] % 64 == 0 and K % 16 == 0)
    # decode batches: blocks of 16 ceil(T / 16) rows up to 64 tokens -- a token routes to an expert at most once, so no
    # expert needs more -- blocks of 64 above (an expert with more rows takes several), and the grouped LDS-DMA streaming
    # kernel (csrc/stream_linear.hip), SiLU * mul in GEMM1's epilogue
    small = (MOE_STREAM_DECODE and T <= MOE_STREAM_MAX_TOKENS and ops.moe_stream_gemm_is_supported(hidden_states, w1, True)
             and w2.shape[2] % 128 == 0 and K % 16 == 0)
    # in between (one prefill request of a many-expert model: ~100 rows per expert): the same kernel in its 128-row x
    # 512-column geometry -- half the padding, and every expert's weights are read by one row of tiles
    if not tall and not small and (numel >= MOE_MID_MIN_ROWS_PER_EXPERT * E and ops.moe_gemm_tall_is_supported(hidden_states, w1, True)
                                   and w2.shape[2] % 64 == 0 and K % 16 == 0):
        tall, tall_block = True, 128
    else:
        tall_block = ops.MOE_TALL_BLOCK_M
    block_m = tall_block if tall else (min(64, 16 * -(-T // 16)) if small else
                                       (MOE_BLOCK_M if numel <= 2048 else 2 * MOE_BLOCK_M))
    max_sorted = -(-(numel + E * (block_m - 1)) // block_m) * block_m
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=dev)
    expert_ids = torch.empty((max_sorted + block_m - 1) // block_m, dtype=torch.int32, device=dev)
    num_post_pad = torch.empty(1, dtype=torch.int32, device=dev)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=dev)
    ops.moe_align_block_size(topk_ids, E, block_m, sorted_ids, expert_ids, num_post_pad, None, cumsum)
    if small:
        c2 = torch.empty((numel, N2 // 2), dtype=dt, device=dev)
        ops.moe_stream_gemm(hidden_states, w1, c2, None, sorted_ids, expert_ids, num_post_pad, numel, topk, False, block_m, True)
        c3 = (torch.zeros if partial_experts else torch.empty)((numel, K), dtype=dt, device=dev)
        ops.moe_stream_gemm(c2, w2, c3, topk_weights.reshape(-1), sorted_ids, expert_ids, num_post_pad, numel, 1, True, block_m)
        return _finish(c3.view(T, topk, K), out_scale, out_addend)
    if tall:
        c2 = torch.empty((numel, N2 // 2), dtype=dt, device=dev)
        ops.moe_gemm_tall(hidden_states, w1, c2, None, sorted_ids, expert_ids, num_post_pad, numel, topk, False, True,
                          block_m=block_m)
        c3 = (torch.zeros if partial_experts else torch.empty)((numel, K), dtype=dt, device=dev)
        ops.moe_gemm_tall(c2, w2, c3, topk_weights.reshape(-1), sorted_ids, expert_ids, num_post_pad, numel, 1, True, False,
                          block_m=block_m)
        return _finish(c3.view(T, topk, K), out_scale, out_addend)
    # prefill-sized calls: SiLU * mul in GEMM1's epilogue (no [T * k, 2N] intermediate); same bits as the two calls
    c2 = ops.moe_grouped_gemm_silu(hidden_states, w1, sorted_ids, expert_ids, num_post_pad, numel, topk, block_m)
    if c2 is None:
        c1 = torch.empty((numel, N2), dtype=dt, device=dev)
        ops.moe_grouped_gemm(hidden_states, w1, c1, None, sorted_ids, expert_ids, num_post_pad, numel, topk, False,
                             block_m)
        c2 = ops.silu_and_mul(c1)
    c3 = (torch.zeros if partial_experts else torch.empty)((numel, K), dtype=dt, device=dev)
    ops.moe_grouped_gemm(c2, w2, c3, topk_weights.reshape(-1), sorted_ids, expert_ids, num_post_pad, numel, 1, True,
                         block_m)
    return _finish(c3.view(T, topk, K), out_scale, out_addend)


def fused_experts_fp8(hidden_states: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, w1_scale: torch.Tensor,
                      w2_scale: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor, block_shape,
                      block_m: Optional[int] = None, expert_offset: int = 0, partial_experts: bool = False,
                      x_quant=None, out_scale: float = 1.0, out_addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fused_experts_impl with use_fp8_w8a8 and block_shape = [block_n, block_k] (fused_moe.py:961-1165;
    activation quantisation inside invoke_fused_moe_kernel :526-545): the activations of both GEMMs are
    quantised per token and group of block_k, the weights are fp8 [E, 2N, K] / [E, K, N] with one scale per
    block_n x block_k tile.  Returns [T, K] in the dtype of hidden_states."""
    topk_ids = _local_ids(topk_ids, expert_offset)
    T, K = hidden_states.shape
    E, N2, _ = w1.shape
    topk = topk_ids.shape[1]
    dev, dt = hidden_states.device, hidden_states.dtype
    numel = T * topk
    block_k = int(block_shape[1])
    if block_m is None:
        block_m = MOE_BLOCK_M if numel <= 2048 else 2 * MOE_BLOCK_M
    max_sorted = numel + E * (block_m - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=dev)
    expert_ids = torch.empty((max_sorted + block_m - 1) // block_m, dtype=torch.int32, device=dev)
    num_post_pad = torch.empty(1, dtype=torch.int32, device=dev)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=dev)
    ops.moe_align_block_size(topk_ids, E, block_m, sorted_ids, expert_ids, num_post_pad, None, cumsum)
    a_q, a_s = x_quant if x_quant is not None else ops.per_token_group_quant_fp8(hidden_states, block_k)
    c1 = torch.empty((numel, N2), dtype=dt, device=dev)
    ops.moe_grouped_gemm_fp8(a_q, a_s, w1, w1_scale, c1, None, sorted_ids, expert_ids, num_post_pad, numel, topk,
                             False, block_shape, block_m)
    c2_q, c2_s = ops.silu_and_mul_quant_fp8(c1, block_k)  # SiLU * mul and the quantisation of its output in one pass
    c3 = (torch.zeros if partial_experts else torch.empty)((numel, K), dtype=dt, device=dev)
    ops.moe_grouped_gemm_fp8(c2_q, c2_s, w2, w2_scale, c3, topk_weights.reshape(-1).float(), sorted_ids, expert_ids,
                             num_post_pad, numel, 1, True, block_shape, block_m)
    return _finish(c3.view(T, topk, K), out_scale, out_addend)


class FusedMoE(nn.Module):
    def __init__(self, num_experts: int, top_k: int, hidden_size: int, intermediate_size: int,
                 renormalize: bool = True, use_grouped_topk: bool = False, num_expert_group: Optional[int] = None,
                 topk_group: Optional[int] = None, correction_bias: Optional[torch.Tensor] = None,
                 reduce_results: bool = False, params_dtype=None, quant_config=None):
        super().__init__()
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.expert_parallel = expert_parallel_enabled() and tp > 1
        if self.expert_parallel:
            # EPMoE (ep_moe/layer.py:113-160): whole experts per rank, full intermediate size
            assert num_experts % tp == 0, f"{num_experts} experts cannot be split over {tp} ranks"
            self.expert_offset = rank * (num_experts // tp)
            self._init_expert_parallel(num_experts, num_experts // tp, rank, hidden_size, intermediate_size,
                                       params_dtype, quant_config)
            self.top_k, self.renormalize = top_k, renormalize
            self.use_grouped_topk, self.num_expert_group, self.topk_group = use_grouped_topk, num_expert_group, topk_group
            self.correction_bias, self.reduce_results, self.quant_config = correction_bias, reduce_results, quant_config
            return
        self.expert_offset = 0
        assert intermediate_size % tp == 0
        n = intermediate_size // tp
        self.top_k, self.renormalize = top_k, renormalize
        self.use_grouped_topk, self.num_expert_group, self.topk_group = use_grouped_topk, num_expert_group, topk_group
        self.correction_bias = correction_bias
        self.reduce_results = reduce_results
        self.quant_config = quant_config
        wdt = FP8_DTYPE if quant_config else params_dtype
        self.w13_weight = nn.Parameter(torch.empty(num_experts, 2 * n, hidden_size, dtype=wdt), requires_grad=False)
        self.w2_weight = nn.Parameter(torch.empty(num_experts, hidden_size, n, dtype=wdt), requires_grad=False)
        if quant_config:
            # Fp8MoEMethod.create_weights (quantization/fp8.py:470-620), block-wise branch
            bn, bk = quant_config.weight_block_size
            check_quantisable_input(hidden_size, (bn, bk), "FusedMoE w13")
            check_quantisable_input(n, (bn, bk), "FusedMoE w2")
            self.w13_weight.weight_block_size = self.w2_weight.weight_block_size = (bn, bk)
            if tp > 1 and (n % bn or n % bk):
                raise ValueError(f"intermediate size per rank {n} is not a multiple of the weight block {bn} x {bk}")
            self.w13_weight_scale_inv = nn.Parameter(
                torch.empty((num_experts,) + scale_shape(2 * n, hidden_size, (bn, bk)), dtype=torch.float32),
                requires_grad=False)
            self.w2_weight_scale_inv = nn.Parameter(
                torch.empty((num_experts,) + scale_shape(hidden_size, n, (bn, bk)), dtype=torch.float32),
                requires_grad=False)
            full_i = intermediate_size
            nb, kb = -(-n // bn), -(-n // bk)
            self.w13_weight_scale_inv.tp_full_shape = (num_experts,) + scale_shape(2 * full_i, hidden_size, (bn, bk))
            self.w13_weight_scale_inv.tp_shard = (lambda full: torch.cat(
                [full[:, rank * nb:(rank + 1) * nb], full[:, full_i // bn + rank * nb: full_i // bn + (rank + 1) * nb]],
                1).contiguous()) if tp > 1 else (lambda full: full)
            self.w2_weight_scale_inv.tp_full_shape = (num_experts,) + scale_shape(hidden_size, full_i, (bn, bk))
            self.w2_weight_scale_inv.tp_shard = (lambda full: full[:, :, rank * kb:(rank + 1) * kb].contiguous()) \
                if tp > 1 else (lambda full: full)
        inter = intermediate_size
        self.w13_weight.tp_full_shape = (num_experts, 2 * inter, hidden_size)
        self.w13_weight.tp_shard = lambda full: torch.cat(
            [full[:, rank * n:(rank + 1) * n], full[:, inter + rank * n: inter + (rank + 1) * n]], 1).contiguous()
        self.w2_weight.tp_full_shape = (num_experts, hidden_size, inter)
        self.w2_weight.tp_shard = lambda full: full[:, :, rank * n:(rank + 1) * n].contiguous()

    def _init_expert_parallel(self, E, e_local, rank, hidden_size, n, params_dtype, quant_config):
        wdt = FP8_DTYPE if quant_config else params_dtype
        lo, hi = rank * e_local, (rank + 1) * e_local
        self.w13_weight = nn.Parameter(torch.empty(e_local, 2 * n, hidden_size, dtype=wdt), requires_grad=False)
        self.w2_weight = nn.Parameter(torch.empty(e_local, hidden_size, n, dtype=wdt), requires_grad=False)
        self.w13_weight.tp_full_shape = (E, 2 * n, hidden_size)
        self.w2_weight.tp_full_shape = (E, hidden_size, n)
        self.w13_weight.tp_shard = self.w2_weight.tp_shard = lambda full: full[lo:hi].contiguous()
        if quant_config:
            bn, bk = quant_config.weight_block_size
            check_quantisable_input(hidden_size, (bn, bk), "FusedMoE w13")
            check_quantisable_input(n, (bn, bk), "FusedMoE w2")
            self.w13_weight.weight_block_size = self.w2_weight.weight_block_size = (bn, bk)
            self.w13_weight_scale_inv = nn.Parameter(
                torch.empty((e_local,) + scale_shape(2 * n, hidden_size, (bn, bk)), dtype=torch.float32), requires_grad=False)
            self.w2_weight_scale_inv = nn.Parameter(
                torch.empty((e_local,) + scale_shape(hidden_size, n, (bn, bk)), dtype=torch.float32), requires_grad=False)
            self.w13_weight_scale_inv.tp_full_shape = (E,) + scale_shape(2 * n, hidden_size, (bn, bk))
            self.w2_weight_scale_inv.tp_full_shape = (E,) + scale_shape(hidden_size, n, (bn, bk))
            self.w13_weight_scale_inv.tp_shard = self.w2_weight_scale_inv.tp_shard = lambda full: full[lo:hi].contiguous()

    # ------------------------------------------------------------------ expert parallelism through the all-to-all
    def all_to_all_comm(self):
        """The peer-memory communicator when this layer routes through the expert all-to-all, else None."""
        if not (self.expert_parallel and expert_all_to_all_enabled()):
            return None
        from semi_pd_amd.distributed import get_custom_all_reduce
        return get_custom_all_reduce()

    def forward_all_to_all(self, hidden_states: torch.Tensor, router_logits: torch.Tensor, comm) -> torch.Tensor:
        """[T, H] replicated in, [T, H] replicated out, COMPLETE (every expert's contribution, not a partial sum): rank r
        routes the tokens of its slice, ep_dispatch brings the rows routed to this rank's experts, the local experts run on
        them as a top-1 problem (each received row has one expert and one weight), ep_combine takes the weighted rows back
        and sums each token's k of them like moe_sum, the slices are all-gathered.  Chunked over tokens so that a chunk's
        rows fit a slot of the peer-memory region."""
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        T, H = hidden_states.shape
        topk_weights, topk_ids = select_experts(hidden_states, router_logits, self.top_k, self.use_grouped_topk,
                                                self.renormalize, self.topk_group, self.num_expert_group,
                                                self.correction_bias)
        k = topk_ids.shape[1]
        e_local = self.w13_weight.shape[0]
        cap = comm.capacity
        chunk = max(tp, (cap - 8192) // (k * H * hidden_states.element_size()))
        out = torch.empty_like(hidden_states)
        for c0 in range(0, T, chunk):
            c1 = min(T, c0 + chunk)
            n = c1 - c0
            per = -(-n // tp)                               # slice length (the last ranks' slices may be shorter or empty)
            lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
            x = hidden_states[c0 + lo: c0 + hi].contiguous()
            ids = topk_ids[c0 + lo: c0 + hi].to(torch.int32).contiguous()
            w = topk_weights[c0 + lo: c0 + hi].float().contiguous()
            # the chunk routes n x k rows in all, so no rank can receive more: overflow is impossible and the read-back of
            # recv_count (a host synchronisation per MoE layer -- and only on the ranks whose slice is a full one when n does
            # not divide by tp, i.e. ranks running out of step) is switched off
            max_recv = n * k
            st = comm.ep_dispatch(x, ids, w, e_local, max_recv, check_overflow=False)
            # rows that did not arrive (beyond recv_count) keep the out-of-range expert id and are dropped by moe_align
            recv_e = torch.where(torch.arange(max_recv, device=x.device) < st["recv_count"], st["recv_expert"],
                                 torch.full_like(st["recv_expert"], e_local)).view(-1, 1)
            recv_w = st["recv_weight"].view(-1, 1)
            if self.quant_config:
                y = fused_experts_fp8(st["recv_x"], self.w13_weight, self.w2_weight, self.w13_weight_scale_inv,
                                      self.w2_weight_scale_inv, recv_w, recv_e, self.quant_config.weight_block_size,
                                      partial_experts=True)
            else:
                y = fused_experts(st["recv_x"], self.w13_weight, self.w2_weight, recv_w, recv_e, partial_experts=True)
            mine = comm.ep_combine(y.contiguous(), st)      # [hi - lo, H]
            padded = torch.zeros((per, H), dtype=mine.dtype, device=mine.device)
            padded[: hi - lo] = mine
            out[c0:c1] = comm.all_gather(padded).view(tp * per, H)[:n]
        return out

    def route(self, hidden_states: torch.Tensor, router_logits):
        """(topk_weights, topk_ids) of this layer's routing rule; router_logits a tensor or the router GEMM's K-slice planes."""
        return select_experts(hidden_states, router_logits, self.top_k, self.use_grouped_topk, self.renormalize,
                              self.topk_group, self.num_expert_group, self.correction_bias)

    def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor, x_quant=None, out_scale: float = 1.0,
                out_addend: Optional[torch.Tensor] = None, topk=None) -> torch.Tensor:
        """out_scale / out_addend: `experts(x) * out_scale + out_addend` with the roundings of the separate element-wise
        ops, in the launch that sums the top-k rows (not combined with reduce_results: the caller reduces afterwards).
        topk: (topk_weights, topk_ids) from self.route when the caller has routed already."""
        if (out_scale != 1.0 or out_addend is not None) and self.reduce_results and get_tensor_model_parallel_world_size() > 1:
            raise RuntimeError("FusedMoE: out_scale / out_addend cannot be combined with reduce_results")
        topk_weights, topk_ids = topk if topk is not None else self.route(hidden_states, router_logits)
        if self.quant_config:
            out = fused_experts_fp8(hidden_states, self.w13_weight, self.w2_weight, self.w13_weight_scale_inv,
                                    self.w2_weight_scale_inv, topk_weights, topk_ids, self.quant_config.weight_block_size,
                                    expert_offset=self.expert_offset, partial_experts=self.expert_parallel,
                                    x_quant=x_quant, out_scale=out_scale, out_addend=out_addend)
        else:
            out = fused_experts(hidden_states, self.w13_weight, self.w2_weight, topk_weights, topk_ids,
                                expert_offset=self.expert_offset, partial_experts=self.expert_parallel,
                                out_scale=out_scale, out_addend=out_addend)
        if self.reduce_results and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out
