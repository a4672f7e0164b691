"""HipAttnBackend: the attention seam of the reference (`AttentionBackend`,
layers/attention/base_attn_backend.py:14-108) implemented on the HIP kernels.

Structure follows TritonAttnBackend (layers/attention/triton_backend.py:19-458): per forward it
builds kv_indptr / kv_indices (+ qo_indptr for extend) once, then every layer calls
forward_extend / forward_decode.  All metadata kernels are hipGraph-capturable, so for decode the
whole metadata build lives inside the captured graph and replay only refreshes the static input
buffers (init_forward_metadata_replay_cuda_graph is a no-op beyond that).
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional

import torch

from semi_pd_amd import ops
from semi_pd_amd.model_executor.forward_batch_info import ForwardBatch, ForwardMode


class AttentionBackend:
    """The ABC of the reference (base_attn_backend.py:14-108), kept verbatim in shape."""

    def init_forward_metadata(self, forward_batch: ForwardBatch):
        raise NotImplementedError()

    def init_cuda_graph_state(self, max_bs: int):
        raise NotImplementedError()

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                 encoder_lens, forward_mode, spec_info):
        raise NotImplementedError()

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                encoder_lens, forward_mode, spec_info, seq_lens_cpu):
        raise NotImplementedError()

    def get_cuda_graph_seq_len_fill_value(self):
        raise NotImplementedError()

    def forward(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache: bool = True):
        if forward_batch.forward_mode.is_decode():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache)
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        raise NotImplementedError()


@dataclass
class ForwardMetadata:
    attn_logits: Optional[torch.Tensor]
    kv_indptr: torch.Tensor
    kv_indices: torch.Tensor
    qo_indptr: Optional[torch.Tensor]
    max_extend_len: int
    num_kv_splits: int


MLA_SHARED_MIN_WORKGROUPS = 160   # csrc/mla_decode_shared.hip: launch_mla_decode_shared takes the shape from here up


def choose_kv_splits(batch: int, num_kv_heads: int, max_seq_len: int, num_cus: int, cap: int,
                     mla: bool = False, mla_heads: int = 0) -> int:
    """Split-KV factor: one full round of work items on the CUs this process owns, never cutting a
    split too short.  (The reference uses a fixed --triton-attention-num-kv-splits, 16 on HIP:
    server_args.py:321-323.)"""
    # GQA / MQA: one work item = one wave of decode_mfma_kernel; 2 waves per SIMD fit, i.e. 8 per CU.
    # Measured on 128 and 256 CUs at ctx ~ 1.1 k (profiles/r01_kbench_decode_small_batches.txt): the
    # best split puts ONE full round of waves on the CUs (B x Hkv x splits ~ 8 x CUs); a second round
    # costs more in per-wave prologue and stage-2 work than it gains in balance.
    # MLA: one work item = one 4-wave workgroup of mla_decode_kernel (the waves share the latent tile in
    # LDS), 2 per CU, and a split below ~128 tokens does not amortise staging Q for 16 heads
    # (profiles/r01_kbench_mla_small_batches.txt: best B x splits = 1 .. 2 x CUs).
    # MLA with 64 / 128 / 256 heads per rank (mla_decode_shared.hip): one work item = one workgroup for up to 128
    # heads, ONE per CU, and it wants >= 8 tiles of 32 rows; the launcher takes that kernel from 160 workgroups up,
    # below that the rule of the 16-head kernels applies (profiles/r02_kbench_mla_decode_short_contexts.txt: at
    # B = 32 .. 96 and ctx 1.1 k .. 4.4 k the best split count is min(CUs / B, ctx / 256))
    if mla and mla_heads >= 64 and mla_heads % 64 == 0:
        groups = (mla_heads + 127) // 128
        s = int(max(1, min(cap, num_cus // max(1, batch * groups), max_seq_len // 256)))
        if batch * groups * s >= MLA_SHARED_MIN_WORKGROUPS:
            return s
    target = (2 if mla else 8) * num_cus
    base = max(1, batch * num_kv_heads)
    want = max(1, target // base)
    by_len = max(1, max_seq_len // (128 if mla else 64))
    return int(max(1, min(cap, want, by_len)))


class HipAttnBackend(AttentionBackend):
    def __init__(self, model_runner, num_kv_splits_cap: int = 32):
        self.device = model_runner.device
        self.num_head = model_runner.num_attention_heads_local
        self.num_kv_head = model_runner.num_kv_heads_local
        self.v_head_dim = model_runner.v_head_dim
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        self.max_context_len = model_runner.max_context_len
        self.is_mla = model_runner.kv_geometry["kind"] == "mla"
        self.num_kv_splits_cap = num_kv_splits_cap
        # --triton-attention-num-kv-splits: a fixed count as in the reference (decode_attention.py's grid z)
        fixed = getattr(model_runner, "num_kv_splits", None)
        self.fixed_kv_splits = int(fixed) if fixed else None
        if self.fixed_kv_splits:
            self.num_kv_splits_cap = max(1, self.fixed_kv_splits)
        self.forward_metadata: Optional[ForwardMetadata] = None
        self.cuda_graph_attn_logits = None
        self.model_runner = model_runner
        self._algo = (0.0, 0.0)  # algorithmic (bytes, flops) per layer call of the current batch
        self._fused_decode_ok = {}

    @property
    def num_cus(self) -> int:
        # the CUs of the stream the instance is on right now (ModelRunner.set_owned_cus)
        return self.model_runner.num_cus_owned

    # ---- eager path ---------------------------------------------------------------------
    def init_forward_metadata(self, forward_batch: ForwardBatch):
        bs = forward_batch.batch_size
        dev = self.device
        kv_indptr = torch.empty(bs + 1, dtype=torch.int32, device=dev)
        if forward_batch.forward_mode.is_decode():
            kv_indices = torch.empty(max(forward_batch.seq_lens_sum, 1), dtype=torch.int32, device=dev)
            ops.create_flashinfer_kv_indices(self.req_to_token, forward_batch.req_pool_indices,
                                             forward_batch.seq_lens, kv_indptr, None, kv_indices)
            max_len = self.max_context_len if forward_batch.seq_lens_sum is None else max(
                1, forward_batch.seq_lens_sum // max(bs, 1))
            splits = self.fixed_kv_splits or choose_kv_splits(
                bs, self.num_kv_head, max_len, self.num_cus, self.num_kv_splits_cap, mla=self.is_mla,
                mla_heads=self.num_head if self.is_mla else 0)
            attn_logits = torch.empty((bs, self.num_head, splits, self.v_head_dim + 1), dtype=torch.float32,
                                      device=dev) if splits > 1 else None
            self.forward_metadata = ForwardMetadata(attn_logits, kv_indptr, kv_indices, None, 0, splits)
            # sum_len * Hkv * (Dk + Dv) * s  +  2 * B * Hq * D * s   per layer
            es = 2
            es_kv = getattr(self.model_runner, "kv_cache_dtype", torch.bfloat16).itemsize  # 1 for fp8 rows
            self._algo = (forward_batch.seq_lens_sum * self.num_kv_head * self._kv_row_elems() * es_kv
                          + 2 * bs * self.num_head * self.v_head_dim * es, 0.0)
        else:
            prefix_sum = int(sum(forward_batch.extend_prefix_lens_cpu))
            kv_indices = torch.empty(max(prefix_sum, 1), dtype=torch.int32, device=dev)
            ops.create_flashinfer_kv_indices(self.req_to_token, forward_batch.req_pool_indices,
                                             forward_batch.extend_prefix_lens, kv_indptr, None, kv_indices)
            qo_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
            qo_indptr[1:] = torch.cumsum(forward_batch.extend_seq_lens, dim=0)
            self.forward_metadata = ForwardMetadata(None, kv_indptr, kv_indices, qo_indptr,
                                                    max(forward_batch.extend_seq_lens_cpu), 1)
            # 2 * Hq * (Dk + Dv) * sum_i ext_i * (pre_i + (ext_i + 1) / 2) flop
            pairs = sum(e * (p + (e + 1) / 2.0) for e, p in zip(forward_batch.extend_seq_lens_cpu,
                                                                 forward_batch.extend_prefix_lens_cpu))
            T = forward_batch.extend_num_tokens
            es = 2
            self._algo = ((2 * T * self.num_head * self.v_head_dim + T * self.num_kv_head * self._kv_row_elems()
                           + prefix_sum * self.num_kv_head * self._kv_row_elems()) * es,
                          2.0 * self.num_head * self._kv_row_elems() * pairs)

    # ---- hipGraph path (decode) ------------------------------------------------------------
    def init_cuda_graph_state(self, max_bs: int):
        dev = self.device
        self.cuda_graph_max_bs = max_bs
        self.cuda_graph_kv_indptr = torch.zeros(max_bs + 1, dtype=torch.int32, device=dev)
        self.cuda_graph_kv_indices = torch.zeros(max_bs * self.max_context_len, dtype=torch.int32, device=dev)
        self.cuda_graph_attn_logits = torch.empty(
            (max_bs, self.num_head, self.num_kv_splits_cap, self.v_head_dim + 1), dtype=torch.float32, device=dev)

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                 encoder_lens=None, forward_mode=ForwardMode.DECODE,
                                                 spec_info=None, num_kv_splits: Optional[int] = None):
        """Called inside the capture: the kv_indices build kernel is recorded into the graph and reads
        the static req_pool_indices / seq_lens buffers on every replay."""
        assert forward_mode.is_decode()
        splits = num_kv_splits or self.fixed_kv_splits or choose_kv_splits(bs, self.num_kv_head, self.max_context_len, self.num_cus,
                                                   self.num_kv_splits_cap, mla=self.is_mla,
                                                   mla_heads=self.num_head if self.is_mla else 0)
        kv_indptr = self.cuda_graph_kv_indptr[: bs + 1]
        ops.create_flashinfer_kv_indices(self.req_to_token, req_pool_indices, seq_lens, kv_indptr, None,
                                         self.cuda_graph_kv_indices)
        logits = self.cuda_graph_attn_logits.view(-1)[: bs * self.num_head * splits * (self.v_head_dim + 1)].view(
            bs, self.num_head, splits, self.v_head_dim + 1)
        self.forward_metadata = ForwardMetadata(logits if splits > 1 else None, kv_indptr,
                                                self.cuda_graph_kv_indices, None, 0, splits)

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                encoder_lens=None, forward_mode=ForwardMode.DECODE,
                                                spec_info=None, seq_lens_cpu=None):
        return  # metadata kernels are part of the graph

    def get_cuda_graph_seq_len_fill_value(self):
        return 1

    def _kv_row_elems(self) -> int:
        geo = self.model_runner.kv_geometry
        if geo["kind"] == "mla":
            return geo["kv_lora_rank"] + geo["qk_rope_head_dim"]  # K and V share the latent row
        return geo["head_dim"] + geo["v_head_dim"]

    def _timing(self):
        kt = getattr(self.model_runner, "kernel_timing", None)
        return kt if (kt is not None and kt.active) else None

    # ---- per-layer calls ----------------------------------------------------------------------
    def forward_extend(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache: bool = True):
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        md = self.forward_metadata
        T = q.shape[0]
        o = torch.empty((T, layer.tp_q_head_num * layer.v_head_dim), dtype=q.dtype, device=q.device)
        kt = self._timing()
        t0 = kt.start() if kt else None
        ops.extend_attention_fwd(
            q.view(T, layer.tp_q_head_num, layer.qk_head_dim),
            k.view(T, layer.tp_k_head_num, layer.qk_head_dim),
            v.view(T, layer.tp_v_head_num, layer.v_head_dim),
            o.view(T, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            md.qo_indptr, md.kv_indptr, md.kv_indices, None, None, md.max_extend_len,
            layer.scaling, layer.logit_cap)
        if kt:
            kt.stop("extend_attention", t0, *self._algo)
        return o

    def forward_decode(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache: bool = True):
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        md = self.forward_metadata
        B = q.shape[0]
        o = torch.empty((B, layer.tp_q_head_num * layer.v_head_dim), dtype=q.dtype, device=q.device)
        kt = self._timing()
        t0 = kt.start() if kt else None
        ops.decode_attention_fwd(
            q.view(B, layer.tp_q_head_num, layer.qk_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            o.view(B, layer.tp_q_head_num, layer.v_head_dim),
            md.kv_indptr, md.kv_indices, md.attn_logits, md.num_kv_splits, layer.scaling, layer.logit_cap)
        if kt:
            kt.stop("decode_attention", t0, *self._algo, n_kernels=2 if md.num_kv_splits > 1 else 1)
        return o


    # ---- decode: RoPE + KV store + attention + split merge in one launch -------------------------
    def fused_decode_waves(self, bs: int, head_dim: int, num_kv_splits: int = 0) -> int:
        """How a decode batch of `bs` requests goes from the qkv GEMM's planes to o_proj's input: 0 = the separate
        launches (rope_and_store_kv_planes, decode_attention_fwd: stage 1 + stage 2), 4 / 8 = the launch whose
        workgroups are (request, kv head) pairs with that many waves as kv splits (ops.decode_rope_attention_planes,
        csrc/decode_attention_fused.hip).  From about half a workgroup per CU up that launch is the whole step between
        the two GEMMs; a smaller batch gets several workgroups per pair (fused_decode_zsplits) and the stage-2 launch
        behind them: two launches instead of three.
        SEMIPD_FUSED_DECODE_ATTN: 0 = off; 2 = one workgroup per pair at every batch size (tests); 4 = only where one
        workgroup per pair fills the chip (the form of the round's first measurements); a fixed
        --triton-attention-num-kv-splits keeps the reference's form."""
        knob = os.environ.get("SEMIPD_FUSED_DECODE_ATTN", "1")
        if self.is_mla or self.fixed_kv_splits or knob == "0":
            return 0
        key = (head_dim,)
        ok = self._fused_decode_ok.get(key)
        if ok is None:
            kv_dtype = getattr(self.model_runner, "kv_cache_dtype", None) or self.model_runner.dtype
            ok = self._fused_decode_ok[key] = ops.decode_rope_attention_planes_supported(
                self.num_head, self.num_kv_head, head_dim, self.model_runner.dtype, kv_dtype)
        if not ok:
            return 0
        wgs, cus = bs * self.num_kv_head, max(1, self.num_cus)
        if wgs >= 2 * cus:
            return 4
        if 2 * wgs >= cus or knob == "2":
            return 8
        # several workgroups per pair: worth it from about one workgroup per eight CUs (Llama-3-8B from one request: step
        # alone 3.32 -> 3.25 ms, at 8 requests 3.65 -> 3.44; the 70B TP = 8 rank's single kv head at 1 .. 8 requests: 16 .. 32
        # workgroups, 0.5-1 % slower than the separate launches, profiles/r06_decode_step_small_batches.txt)
        z = max(1, min(64, (int(num_kv_splits or 32) + 7) // 8))
        return 8 if (knob != "4" and 8 * wgs * z >= cus) else 0

    def fused_decode_zsplits(self, bs: int, num_kv_splits: int) -> int:
        """Workgroups per (request, kv head) of the fused launch: 1 where that fills the chip, otherwise what brings its
        8-wave workgroups to the split count choose_kv_splits picked for this batch (its rule: about eight waves per CU,
        no split shorter than 64 tokens)."""
        wgs, cus = bs * self.num_kv_head, max(1, self.num_cus)
        if 2 * wgs >= cus or os.environ.get("SEMIPD_FUSED_DECODE_ATTN", "1") == "2":
            return 1
        return max(1, min(64, (int(num_kv_splits) + 7) // 8))

    def forward_decode_rope_planes(self, positions, qkv_planes, rotary_emb, layer, forward_batch: ForwardBatch, waves: int):
        md = self.forward_metadata
        pool = forward_batch.token_to_kv_pool
        zsplits = self.fused_decode_zsplits(qkv_planes.rows, md.num_kv_splits) if waves == 8 else 1
        if zsplits > 1 and md.attn_logits is None:
            zsplits = 1
        kt = self._timing()
        t0 = kt.start() if kt else None
        o = ops.decode_rope_attention_planes(
            positions, qkv_planes, layer.tp_q_head_num, layer.tp_k_head_num, layer.head_dim, rotary_emb.cos_sin_cache,
            pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id), forward_batch.out_cache_loc,
            md.kv_indptr, md.kv_indices, waves, layer.scaling, layer.logit_cap, zsplits=zsplits,
            attn_logits=md.attn_logits if zsplits > 1 else None)
        if kt:
            kt.stop("decode_attention", t0, *self._algo, n_kernels=2 if zsplits > 1 else 1)
        return o


class RadixAttention:
    """Per-layer attention descriptor (layers/radix_attention.py:21-70): head counts, scaling,
    layer id; forward() reshapes nothing and dispatches to the batch's backend."""

    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int, layer_id: int,
                 logit_cap: float = 0.0, v_head_dim: int = -1):
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.head_dim = head_dim
        self.qk_head_dim = head_dim
        self.v_head_dim = v_head_dim if v_head_dim != -1 else head_dim
        self.scaling = scaling
        self.layer_id = layer_id
        self.logit_cap = logit_cap

    def forward(self, q, k, v, forward_batch: ForwardBatch, save_kv_cache: bool = True):
        return forward_batch.attn_backend.forward(q, k, v, self, forward_batch, save_kv_cache)

    __call__ = forward
