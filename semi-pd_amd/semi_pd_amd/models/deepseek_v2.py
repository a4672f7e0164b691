"""DeepSeek-V2 / V2-Lite (MLA attention + MoE) on the HIP hot-path operators.

Reference: models/deepseek_v2.py — MoEGate :119-138, DeepseekV2MoE :141-210, DeepseekV2AttentionMLA
:393-850 (forward_normal :591-631 for a prefill without prefix, forward_absorb :633-706 for decode and
for prefill over a prefix), decoder layer / model :853-1060, w_kc / w_vc construction :1215-1249.

MLA keeps ONE latent row [kv_lora_rank + qk_rope_head_dim] per token and layer in the KV pool.
* no prefix (forward_normal): expand the latent to per-head K/V with kv_b_proj and run ordinary MHA
  attention (Dk = 192, Dv = 128) over the new tokens only;
* otherwise (forward_absorb): fold W_kc into the query and W_vc into the output, so attention is MQA
  over the 576-wide latent rows (Dv = 512), straight out of the paged pool.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import os

import torch
import torch.nn.functional as F
from torch import nn

from semi_pd_amd import ops
from semi_pd_amd.distributed import (get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce)
from semi_pd_amd.layers.attention_backend import RadixAttention
from semi_pd_amd.layers.basic import (stream_linear_enabled, ColumnParallelLinear, LogitsProcessor, MergedColumnParallelLinear,
                                      ParallelLMHead, ReplicatedLinear, RMSNorm, RowParallelLinear, SiluAndMul,
                                      VocabParallelEmbedding, gate_up_silu, get_rope, yarn_get_mscale)
from semi_pd_amd.layers.fp8 import (FP8_DTYPE, Fp8Config, apply_w8a8_block_fp8_linear, block_dequantize_weight,
                                    block_quant_to_tensor_quant,
                                    quantize_activation)
from semi_pd_amd.layers.moe import FusedMoE


@dataclass
class DeepseekV2Config:
    """HF DeepseekV2Config fields read by the reference; defaults = DeepSeek-V2-Lite."""
    vocab_size: int = 102400
    hidden_size: int = 2048
    intermediate_size: int = 10944
    moe_intermediate_size: int = 1408
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    n_shared_experts: Optional[int] = 2
    n_routed_experts: int = 64
    num_experts_per_tok: int = 6
    routed_scaling_factor: float = 1.0
    topk_method: str = "greedy"
    n_group: int = 1
    topk_group: int = 1
    norm_topk_prob: bool = False
    first_k_dense_replace: int = 1
    moe_layer_freq: int = 1
    kv_lora_rank: int = 512
    q_lora_rank: Optional[int] = None
    qk_rope_head_dim: int = 64
    qk_nope_head_dim: int = 128
    v_head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    rope_scaling: Optional[Dict[str, Any]] = field(default_factory=lambda: {
        "type": "yarn", "factor": 40, "beta_fast": 32, "beta_slow": 1, "mscale": 0.707,
        "mscale_all_dim": 0.707, "original_max_position_embeddings": 4096})
    max_position_embeddings: int = 163840
    quantization_config: Optional[Dict[str, Any]] = None  # {"quant_method": "fp8", "weight_block_size": [128, 128]}
    tie_word_embeddings: bool = False
    architectures: tuple = ("DeepseekV2ForCausalLM",)


DEEPSEEK_V2_LITE = DeepseekV2Config()


def quant_config_of(config) -> Optional[Fp8Config]:
    """models/deepseek_v2.py passes `quant_config` to every linear and to FusedMoE; the gate, the norms,
    the embedding and lm_head stay in the activation dtype."""
    return Fp8Config.from_hf(getattr(config, "quantization_config", None))


class DeepseekV2MLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, dtype, reduce_results: bool = True, quant_config=None):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, params_dtype=dtype,
                                                       quant_config=quant_config)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, reduce_results=reduce_results,
                                           params_dtype=dtype, quant_config=quant_config)
        self.act_fn = SiluAndMul()

    def forward(self, x, x_quant=None, defer_down: bool = False):
        """defer_down: the caller sums down_proj's K-slice planes itself (ops.SplitKPlanes for decode batches, a tensor
        otherwise)."""
        qc = self.down_proj.quant_config
        if qc is None and self.gate_up_proj.quant_config is None:
            return self.down_proj(gate_up_silu(x, self.gate_up_proj, self.act_fn), defer_reduce=defer_down)
        gate_up = self.gate_up_proj(x, x_quant=x_quant)
        if qc is not None and gate_up.dim() == 2:
            # block-fp8: SiLU * mul and the quantisation in front of down_proj in one kernel
            q, s = ops.silu_and_mul_quant_fp8(gate_up, qc.weight_block_size[1])
            return self.down_proj.forward_prequantized(q, s, x.dtype)
        return self.down_proj(self.act_fn(gate_up))


class MoEGate(nn.Module):
    def __init__(self, config: DeepseekV2Config, dtype):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(config.n_routed_experts, config.hidden_size, dtype=dtype),
                                   requires_grad=False)
        if config.topk_method == "noaux_tc":
            self.e_score_correction_bias = nn.Parameter(torch.zeros(config.n_routed_experts, dtype=torch.float32),
                                                        requires_grad=False)
        else:
            self.e_score_correction_bias = None

    def forward(self, hidden_states, planes_ok: bool = False):
        # decode batches: the weight-streaming GEMM, stopped before its K-slice reduction -- the routing kernel sums the
        # planes (ops.grouped_topk on a SplitKPlanes): the library GEMM takes 10-18 us for these shapes
        if (planes_ok and hidden_states.dim() == 2 and stream_linear_enabled()
                and hidden_states.shape[0] <= ops.STREAM_LINEAR_MAX_ROWS and self.weight.shape[0] % 8 == 0
                and ops.stream_linear_is_supported(hidden_states, self.weight)
                and os.environ.get("SEMIPD_MOE_GATE_PLANES", "1") != "0"):
            return ops.stream_linear_planes(hidden_states, self.weight)
        return F.linear(hidden_states, self.weight, None)


class DeepseekV2MoE(nn.Module):
    def __init__(self, config: DeepseekV2Config, dtype):
        super().__init__()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.routed_scaling_factor = config.routed_scaling_factor
        self.n_shared_experts = config.n_shared_experts
        self.gate = MoEGate(config, dtype)
        self.experts = FusedMoE(config.n_routed_experts, config.num_experts_per_tok, config.hidden_size,
                                config.moe_intermediate_size, renormalize=config.norm_topk_prob,
                                use_grouped_topk=True, num_expert_group=config.n_group,
                                topk_group=config.topk_group, correction_bias=self.gate.e_score_correction_bias,
                                params_dtype=dtype, quant_config=quant_config_of(config))
        self.shared_experts = None
        if config.n_shared_experts is not None:
            self.shared_experts = DeepseekV2MLP(config.hidden_size,
                                                config.moe_intermediate_size * config.n_shared_experts, dtype,
                                                reduce_results=False, quant_config=quant_config_of(config))

    def forward(self, hidden_states: torch.Tensor, x_quant=None) -> torch.Tensor:
        # block-fp8: the shared experts and the routed experts read the same activation: quantise it once
        qc = self.experts.quant_config
        if x_quant is None and qc is not None and self.shared_experts is not None and hidden_states.dim() == 2:
            x_quant = quantize_activation(hidden_states, qc.weight_block_size)
        comm = self.experts.all_to_all_comm() if (self.tp_size > 1 and hidden_states.dim() == 2) else None
        if (comm is None and self.experts.use_grouped_topk and qc is None and hidden_states.dim() == 2 and self.tp_size == 1
                and hidden_states.shape[0] <= ops.STREAM_LINEAR_MAX_ROWS and stream_linear_enabled()
                and os.environ.get("SEMIPD_MOE_SHARED_PLANES", "1") != "0"):
            # decode batch of an unquantised model on one GPU: route first (the routing kernel sums the router GEMM's planes),
            # then the shared experts, whose down_proj planes are summed by the launch that sums the top-k rows -- nothing
            # else touches the GEMM workspace in between (the expert GEMMs have none)
            topk = self.experts.route(hidden_states, self.gate(hidden_states, planes_ok=True))
            shared = self.shared_experts(hidden_states, defer_down=True) if self.shared_experts is not None else None
            return self.experts(hidden_states, None, out_scale=float(self.routed_scaling_factor), out_addend=shared, topk=topk)
        shared_output = self.shared_experts(hidden_states, x_quant=x_quant) if self.shared_experts is not None else None
        # (the planes must be consumed by the next user of the GEMM workspace: the routing kernel at the top of self.experts)
        router_logits = self.gate(hidden_states, planes_ok=comm is None and self.experts.use_grouped_topk)
        if comm is not None:
            # --enable-ep-all-to-all: the routed part comes back COMPLETE and replicated; only the (tensor-parallel)
            # shared experts still hold a partial sum
            out = self.experts.forward_all_to_all(hidden_states, router_logits, comm)
            if self.routed_scaling_factor != 1.0:
                out = out * self.routed_scaling_factor
            if shared_output is not None:
                out = out + tensor_model_parallel_all_reduce(shared_output)
            return out
        if hidden_states.dim() == 2 and (shared_output is None or shared_output.is_contiguous()):
            # `* routed_scaling_factor` and `+ shared_output` ride in the launch that sums the top-k rows (same roundings)
            out = self.experts(hidden_states, router_logits, x_quant=x_quant, out_scale=float(self.routed_scaling_factor),
                               out_addend=shared_output)
        else:
            out = self.experts(hidden_states, router_logits, x_quant=x_quant)
            if self.routed_scaling_factor != 1.0:
                out = out * self.routed_scaling_factor
            if shared_output is not None:
                out = out + shared_output
        if self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class DeepseekV2AttentionMLA(nn.Module):
    def __init__(self, config: DeepseekV2Config, layer_id: int, dtype):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.layer_id = layer_id
        self.qk_nope_head_dim, self.qk_rope_head_dim = config.qk_nope_head_dim, config.qk_rope_head_dim
        self.qk_head_dim = self.qk_nope_head_dim + self.qk_rope_head_dim
        self.v_head_dim, self.kv_lora_rank, self.q_lora_rank = config.v_head_dim, config.kv_lora_rank, config.q_lora_rank
        assert config.num_attention_heads % tp == 0
        self.num_local_heads = config.num_attention_heads // tp
        H = config.num_attention_heads
        qc = self.quant_config = quant_config_of(config)
        if self.q_lora_rank is not None:
            self.q_a_proj = ReplicatedLinear(config.hidden_size, self.q_lora_rank, params_dtype=dtype, quant_config=qc)
            self.q_a_layernorm = RMSNorm(self.q_lora_rank, eps=config.rms_norm_eps)
            self.q_b_proj = ColumnParallelLinear(self.q_lora_rank, H * self.qk_head_dim, params_dtype=dtype,
                                                 quant_config=qc)
        else:
            self.q_proj = ColumnParallelLinear(config.hidden_size, H * self.qk_head_dim, params_dtype=dtype,
                                               quant_config=qc)
        self.kv_a_proj_with_mqa = ReplicatedLinear(config.hidden_size, self.kv_lora_rank + self.qk_rope_head_dim,
                                                   params_dtype=dtype, quant_config=qc)
        self.kv_a_layernorm = RMSNorm(self.kv_lora_rank, eps=config.rms_norm_eps)
        self.kv_b_proj = ColumnParallelLinear(self.kv_lora_rank, H * (self.qk_nope_head_dim + self.v_head_dim),
                                              params_dtype=dtype, quant_config=qc)
        self.o_proj = RowParallelLinear(H * self.v_head_dim, config.hidden_size, params_dtype=dtype, quant_config=qc)
        self.params_dtype = dtype
        rope_scaling = dict(config.rope_scaling) if config.rope_scaling else None
        self.scaling = self.qk_head_dim ** -0.5
        if rope_scaling:
            rope_scaling["rope_type"] = "deepseek_yarn"
            mscale = yarn_get_mscale(rope_scaling["factor"], float(rope_scaling.get("mscale_all_dim", False)))
            self.scaling = self.scaling * mscale * mscale
        self.rotary_emb = get_rope(self.qk_rope_head_dim, self.qk_rope_head_dim, config.max_position_embeddings,
                                   config.rope_theta, False, rope_scaling, dtype)
        self.attn_mqa = RadixAttention(self.num_local_heads, self.kv_lora_rank + self.qk_rope_head_dim, self.scaling,
                                       num_kv_heads=1, layer_id=layer_id, v_head_dim=self.kv_lora_rank)
        self.attn_mha = RadixAttention(self.num_local_heads, self.qk_head_dim, self.scaling,
                                       num_kv_heads=self.num_local_heads, layer_id=layer_id,
                                       v_head_dim=self.v_head_dim)
        # Semi-PD: buffers so that they are exported / imported through IPC (deepseek_v2.py:532-536)
        # a block-fp8 model keeps them in fp8 with ONE scale for both (bmm_fp8 path, deepseek_v2.py:659-665, 690-700);
        # SEMIPD_MLA_ABSORB_BF16=1 restores the dequantise-at-load + torch.bmm form
        self.absorb_fp8 = qc is not None and os.environ.get("SEMIPD_MLA_ABSORB_BF16", "0") != "1"
        wdt = FP8_DTYPE if self.absorb_fp8 else dtype
        self.register_buffer("w_kc", torch.empty(0, dtype=wdt), persistent=False)
        self.register_buffer("w_vc", torch.empty(0, dtype=wdt), persistent=False)
        if self.absorb_fp8:
            self.register_buffer("w_scale", torch.empty(1, dtype=torch.float32), persistent=False)
        # decode batches of an unquantised model without q_lora: q_proj and kv_a_proj_with_mqa read the same row, so they
        # run as ONE weight-streaming GEMM on the concatenated weight (post_load_weights) whose K-slice planes go straight
        # into ops.mla_decode_prep; a buffer like W_kc / W_vc so that both instances see it through IPC
        self.merged_qkv_a = (qc is None and self.q_lora_rank is None and self.kv_lora_rank <= 512 and
                             self.kv_lora_rank % 8 == 0 and self.qk_nope_head_dim % 8 == 0 and
                             self.qk_rope_head_dim % 16 == 0 and os.environ.get("SEMIPD_MLA_MERGED_QKV_A", "1") != "0")
        # block-fp8 with q_lora (DeepSeek-V3): q_a_proj and kv_a_proj_with_mqa read the same quantised row -- one block-fp8
        # GEMM on the concatenated weight for decode batches (q_lora_rank is a multiple of the scale block, so the two
        # scale tables concatenate as well): one GEMM and one K-slice reduction less per layer
        self.merged_qkv_a_fp8 = (qc is not None and self.q_lora_rank is not None and
                                 self.q_lora_rank % int(qc.weight_block_size[0]) == 0 and
                                 os.environ.get("SEMIPD_MLA_MERGED_QKV_A", "1") != "0")
        if self.merged_qkv_a_fp8:
            self.register_buffer("w_qkv_a_fp8", torch.empty(0, dtype=FP8_DTYPE), persistent=False)
            self.register_buffer("s_qkv_a_fp8", torch.empty(0, dtype=torch.float32), persistent=False)
        if self.merged_qkv_a:
            self.register_buffer("w_qkv_a", torch.empty(0, dtype=dtype), persistent=False)
            # W_kc / W_vc once more with K contiguous ([H, 512, 128], [H, 128, 512]) for ops.bmm_nk: decode batches
            self.register_buffer("w_kc_nk", torch.empty(0, dtype=dtype), persistent=False)
            self.register_buffer("w_vc_nk", torch.empty(0, dtype=dtype), persistent=False)

    def post_load_weights(self):
        """deepseek_v2.py:1228-1249: W_kc [H,128,512] and W_vc [H,512,128] out of kv_b_proj."""
        w = self.kv_b_proj.weight
        if self.merged_qkv_a_fp8:
            self.w_qkv_a_fp8 = torch.cat([self.q_a_proj.weight.data, self.kv_a_proj_with_mqa.weight.data], 0).contiguous()
            self.s_qkv_a_fp8 = torch.cat([self.q_a_proj.weight_scale_inv.data,
                                          self.kv_a_proj_with_mqa.weight_scale_inv.data], 0).contiguous()
        if self.merged_qkv_a:
            # rows: this rank's q heads, then the (replicated) latent + k_pe rows
            self.w_qkv_a = torch.cat([self.q_proj.weight.data, self.kv_a_proj_with_mqa.weight.data], 0).contiguous()
        if self.absorb_fp8:
            # deepseek_v2.py:1195-1209: the block-quantised kv_b_proj re-quantised per tensor; W_kc / W_vc stay fp8 and
            # go through the fp8 matrix cores (ops.bmm_fp8).  Stored column-major for the product they are used in:
            # w_kc [H, n = 512, k = 128], w_vc [H, n = 128, k = 512] (the reference's buffers, deepseek_v2.py:1234-1236)
            wq, w_scale = block_quant_to_tensor_quant(w, self.kv_b_proj.weight_scale_inv, self.quant_config.weight_block_size)
            w3 = wq.unflatten(0, (-1, self.qk_nope_head_dim + self.v_head_dim))
            self.w_kc = w3[:, : self.qk_nope_head_dim, :].transpose(1, 2).contiguous()
            self.w_vc = w3[:, self.qk_nope_head_dim:, :].contiguous()
            self.w_scale = w_scale.reshape(1).to(torch.float32)
            return
        if self.quant_config:
            w = block_dequantize_weight(w, self.kv_b_proj.weight_scale_inv, self.quant_config.weight_block_size,
                                        self.params_dtype)
        w_kc, w_vc = w.unflatten(0, (-1, self.qk_nope_head_dim + self.v_head_dim)).split(
            [self.qk_nope_head_dim, self.v_head_dim], dim=1)
        self.w_kc = w_kc.contiguous()                    # [H, 128, 512]
        self.w_vc = w_vc.transpose(1, 2).contiguous()    # [H, 512, 128]
        if self.merged_qkv_a:
            self.w_kc_nk = w_kc.transpose(1, 2).contiguous()   # [H, 512, 128]
            self.w_vc_nk = w_vc.contiguous()                   # [H, 128, 512]

    def _x_quant(self, hidden_states):
        """block-fp8: q_(a_)proj and kv_a_proj_with_mqa read the same activation: quantise it once."""
        if self.quant_config is None or hidden_states.dim() != 2:
            return None
        return quantize_activation(hidden_states, self.quant_config.weight_block_size)

    def _qkv_a_fp8(self, hidden_states, x_quant):
        """(q_a [T, q_lora], latent [T, 576]) as views of ONE block-fp8 GEMM's output for decode batches, or (None, None)."""
        if not (self.merged_qkv_a_fp8 and self.w_qkv_a_fp8.numel() and hidden_states.dim() == 2
                and 0 < hidden_states.shape[0] <= ops.STREAM_LINEAR_MAX_ROWS):
            return None, None
        y = apply_w8a8_block_fp8_linear(hidden_states, self.w_qkv_a_fp8, self.quant_config.weight_block_size,
                                        self.s_qkv_a_fp8, None, x_quant=x_quant)
        return y[:, : self.q_lora_rank], y[:, self.q_lora_rank:]

    def _q(self, hidden_states, x_quant=None, q_a=None):
        if self.q_lora_rank is not None:
            if q_a is None:
                q_a = self.q_a_proj(hidden_states, x_quant=x_quant)
            qc = self.quant_config
            if (qc is not None and q_a.dim() == 2 and q_a.shape[0] > 0 and self.q_lora_rank % int(qc.weight_block_size[1]) == 0
                    and int(qc.weight_block_size[1]) in (64, 128, 256, 512) and self.q_lora_rank <= 8192
                    and os.environ.get("SEMIPD_MLA_QA_NORM_QUANT", "1") != "0"):
                # block-fp8: q_a_layernorm and the quantisation in front of q_b_proj in one kernel
                x, xq = ops.rmsnorm_quant_fp8(q_a, self.q_a_layernorm.weight.data, self.q_a_layernorm.variance_epsilon,
                                              int(qc.weight_block_size[1]))
                q = self.q_b_proj(x, x_quant=xq)
            else:
                q = self.q_b_proj(self.q_a_layernorm(q_a))
        else:
            q = self.q_proj(hidden_states, x_quant=x_quant)
        return q.view(-1, self.num_local_heads, self.qk_head_dim)

    def _latent(self, hidden_states, positions, q, x_quant=None, latent=None):
        """kv_a_proj -> RMSNorm on the 512 latent dims -> RoPE on the 64 rope dims (and on q_pe);
        returns the finished latent rows [T, 1, 576]."""
        if latent is None:
            latent = self.kv_a_proj_with_mqa(hidden_states, x_quant=x_quant)  # [T, 576]
        kv_a = ops.rmsnorm(latent[:, : self.kv_lora_rank], self.kv_a_layernorm.weight.data,
                           self.kv_a_layernorm.variance_epsilon, out=latent[:, : self.kv_lora_rank])
        del kv_a
        latent = latent.unsqueeze(1)
        ops.apply_rope_strided_inplace(positions, q[..., self.qk_nope_head_dim:], latent[..., self.kv_lora_rank:],
                                       self.rotary_emb.cos_sin_cache, False)
        return latent

    def forward(self, positions, hidden_states, forward_batch, x_quant=None):
        no_absorb = forward_batch.forward_mode.is_extend() and sum(forward_batch.extend_prefix_lens_cpu) == 0
        if no_absorb:
            return self.forward_normal(positions, hidden_states, forward_batch, x_quant)
        return self.forward_absorb(positions, hidden_states, forward_batch, x_quant)

    def forward_normal(self, positions, hidden_states, forward_batch, x_quant=None):
        xq = x_quant if x_quant is not None else self._x_quant(hidden_states)
        q_a, lat = self._qkv_a_fp8(hidden_states, xq)
        q = self._q(hidden_states, xq, q_a)
        latent = self._latent(hidden_states, positions, q, xq, lat)
        forward_batch.token_to_kv_pool.set_kv_buffer(self.attn_mha, forward_batch.out_cache_loc, latent, None)
        kv = self.kv_b_proj(latent[:, 0, : self.kv_lora_rank])
        kv = kv.view(-1, self.num_local_heads, self.qk_nope_head_dim + self.v_head_dim)
        k = torch.empty_like(q)
        k[..., : self.qk_nope_head_dim] = kv[..., : self.qk_nope_head_dim]
        k[..., self.qk_nope_head_dim:] = latent[..., self.kv_lora_rank:]
        v = kv[..., self.qk_nope_head_dim:].contiguous()
        attn_output = self.attn_mha(q.reshape(q.shape[0], -1), k.view(k.shape[0], -1), v.view(v.shape[0], -1),
                                    forward_batch, save_kv_cache=False)
        return self.o_proj(attn_output)

    def _decode_prep_fused(self, positions, hidden_states, forward_batch):
        """Decode batch: ONE GEMM for q and the latent, then ONE launch for everything up to the attention
        (ops.mla_decode_prep): returns (q_nope [T, H, nope], q_input with its rope columns filled), or None when this
        call is not eligible."""
        if not (self.merged_qkv_a and forward_batch.forward_mode.is_decode() and self.w_qkv_a.numel()
                and hidden_states.dim() == 2 and hidden_states.shape[0] <= ops.STREAM_LINEAR_MAX_ROWS
                and stream_linear_enabled() and ops.stream_linear_is_supported(hidden_states, self.w_qkv_a)
                and self.w_qkv_a.shape[0] % 8 == 0):
            return None
        T = hidden_states.shape[0]
        planes = ops.stream_linear_planes(hidden_states, self.w_qkv_a)
        q_input = torch.empty((T, self.num_local_heads, self.kv_lora_rank + self.qk_rope_head_dim),
                              dtype=hidden_states.dtype, device=hidden_states.device)
        pool = forward_batch.token_to_kv_pool
        q_nope = ops.mla_decode_prep(planes, positions, self.rotary_emb.cos_sin_cache, self.kv_a_layernorm.weight.data,
                                     self.kv_a_layernorm.variance_epsilon, self.num_local_heads, self.qk_nope_head_dim,
                                     self.qk_rope_head_dim, self.kv_lora_rank, pool.get_key_buffer(self.layer_id),
                                     forward_batch.out_cache_loc, q_input)
        return q_nope, q_input

    def forward_absorb(self, positions, hidden_states, forward_batch, x_quant=None):
        fused = self._decode_prep_fused(positions, hidden_states, forward_batch) if x_quant is None else None
        if fused is not None:
            q_nope, q_input = fused
            T = q_nope.shape[0]
            own_bmm = self.w_kc_nk.numel() and os.environ.get("SEMIPD_MLA_OWN_BMM", "1") != "0"
            if own_bmm:
                ops.bmm_nk(q_nope.transpose(0, 1), self.w_kc_nk, out=q_input[..., : self.kv_lora_rank].transpose(0, 1))
            else:
                torch.bmm(q_nope.transpose(0, 1), self.w_kc, out=q_input[..., : self.kv_lora_rank].transpose(0, 1))
            attn_output = self.attn_mqa(q_input.view(T, -1), None, None, forward_batch, save_kv_cache=False)
            attn_output = attn_output.view(T, self.num_local_heads, self.kv_lora_rank)
            out = torch.empty((T, self.num_local_heads, self.v_head_dim), dtype=q_nope.dtype, device=q_nope.device)
            if own_bmm:
                ops.bmm_nk(attn_output.transpose(0, 1), self.w_vc_nk, out=out.transpose(0, 1))
            else:
                torch.bmm(attn_output.transpose(0, 1), self.w_vc, out=out.transpose(0, 1))
            return self.o_proj(out.view(T, -1), defer_reduce=True)
        xq = x_quant if x_quant is not None else self._x_quant(hidden_states)
        q_a, lat = self._qkv_a_fp8(hidden_states, xq)
        q = self._q(hidden_states, xq, q_a)
        T = q.shape[0]
        q_input = torch.empty((T, self.num_local_heads, self.kv_lora_rank + self.qk_rope_head_dim),
                              dtype=q.dtype, device=q.device)
        # decode batches: kv_a_layernorm, RoPE (q_pe, k_pe), q_pe -> q_input and the KV-pool store in ONE launch
        # (ops.mla_decode_prep_rows; the bf16 path without q_lora goes through _decode_prep_fused above)
        prep_rows = (forward_batch.forward_mode.is_decode() and hidden_states.dim() == 2 and self.kv_lora_rank <= 512
                     and self.kv_lora_rank % 8 == 0 and self.qk_nope_head_dim % 8 == 0 and self.qk_rope_head_dim % 16 == 0
                     and q.dtype in (torch.bfloat16, torch.float16) and os.environ.get("SEMIPD_MLA_PREP_ROWS", "1") != "0")
        if prep_rows:
            lat_raw = lat if lat is not None else self.kv_a_proj_with_mqa(hidden_states, x_quant=xq)
            ops.mla_decode_prep_rows(q, lat_raw, positions, self.rotary_emb.cos_sin_cache, self.kv_a_layernorm.weight.data,
                                     self.kv_a_layernorm.variance_epsilon, self.qk_nope_head_dim, self.kv_lora_rank,
                                     forward_batch.token_to_kv_pool.get_key_buffer(self.layer_id),
                                     forward_batch.out_cache_loc, q_input)
            latent = None
        else:
            latent = self._latent(hidden_states, positions, q, xq, lat)
        if self.absorb_fp8:
            # q_nope quantised per tensor, fp8 x fp8 on the matrix cores, written straight into q_input's layout
            q_val, q_scale = ops.input_to_float8(q[..., : self.qk_nope_head_dim].transpose(0, 1), FP8_DTYPE)
            ops.bmm_fp8(q_val, self.w_kc.transpose(1, 2), q_scale, self.w_scale, q.dtype,
                        out=q_input[..., : self.kv_lora_rank].transpose(0, 1))
        else:
            # [H, T, 512], written through the strides of q_input's layout (no copy behind the batched GEMM)
            torch.bmm(q[..., : self.qk_nope_head_dim].transpose(0, 1), self.w_kc,
                      out=q_input[..., : self.kv_lora_rank].transpose(0, 1))
        if latent is not None:
            q_input[..., self.kv_lora_rank:] = q[..., self.qk_nope_head_dim:]
            forward_batch.token_to_kv_pool.set_kv_buffer(self.attn_mqa, forward_batch.out_cache_loc, latent, None)
            attn_output = self.attn_mqa(q_input.view(T, -1), latent.view(T, -1), latent[..., : self.kv_lora_rank],
                                        forward_batch, save_kv_cache=False)
        else:
            attn_output = self.attn_mqa(q_input.view(T, -1), None, None, forward_batch, save_kv_cache=False)
        attn_output = attn_output.view(T, self.num_local_heads, self.kv_lora_rank)
        if self.absorb_fp8:
            a_val, a_scale = ops.input_to_float8(attn_output.transpose(0, 1), FP8_DTYPE)
            out = torch.empty((T, self.num_local_heads, self.v_head_dim), dtype=q.dtype, device=q.device)
            ops.bmm_fp8(a_val, self.w_vc.transpose(1, 2), a_scale, self.w_scale, q.dtype, out=out.transpose(0, 1))
            return self.o_proj(out.view(T, -1))
        out = torch.empty((T, self.num_local_heads, self.v_head_dim), dtype=q.dtype, device=q.device)
        torch.bmm(attn_output.transpose(0, 1), self.w_vc, out=out.transpose(0, 1))  # [H, T, 128] in [T, H, 128]'s memory
        # (its K-slice planes are summed by post_attention_layernorm: one launch less for decode batches)
        return self.o_proj(out.view(T, -1), defer_reduce=True)


class DeepseekV2DecoderLayer(nn.Module):
    def __init__(self, config: DeepseekV2Config, layer_id: int, dtype):
        super().__init__()
        self.self_attn = DeepseekV2AttentionMLA(config, layer_id, dtype)
        is_moe = (config.n_routed_experts is not None and layer_id >= config.first_k_dense_replace
                  and layer_id % config.moe_layer_freq == 0)
        self.mlp = DeepseekV2MoE(config, dtype) if is_moe else DeepseekV2MLP(
            config.hidden_size, config.intermediate_size, dtype, quant_config=quant_config_of(config))
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        qc = quant_config_of(config)
        self.quant_group = int(qc.weight_block_size[1]) if (qc and config.hidden_size <= 8192) else None

    def forward(self, positions, hidden_states, forward_batch, residual):
        # block-fp8: the norms quantise their output for the layers behind them in the same kernel
        group = self.quant_group if hidden_states.dim() == 2 else None
        attn_quant = mlp_quant = None
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        elif group:
            hidden_states, residual, attn_quant = self.input_layernorm.forward_quant(hidden_states, residual, group)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch, x_quant=attn_quant)
        if group:
            hidden_states, residual, mlp_quant = self.post_attention_layernorm.forward_quant(hidden_states, residual, group)
        else:
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states, x_quant=mlp_quant)
        return hidden_states, residual


class DeepseekV2Model(nn.Module):
    def __init__(self, config: DeepseekV2Config, dtype):
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size, params_dtype=dtype)
        self.layers = nn.ModuleList([DeepseekV2DecoderLayer(config, i, dtype)
                                     for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class DeepseekV2ForCausalLM(nn.Module):
    def __init__(self, config: DeepseekV2Config, dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.model = DeepseekV2Model(config, dtype)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, params_dtype=dtype)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    @property
    def kv_geometry(self):
        a = self.model.layers[0].self_attn
        return dict(kind="mla", num_kv_heads=1, num_heads=a.num_local_heads,
                    head_dim=a.kv_lora_rank + a.qk_rope_head_dim, v_head_dim=a.kv_lora_rank,
                    kv_lora_rank=a.kv_lora_rank, qk_rope_head_dim=a.qk_rope_head_dim,
                    num_layers=len(self.model.layers))

    def post_load_weights(self):
        for layer in self.model.layers:
            layer.self_attn.post_load_weights()

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.model(input_ids, positions, forward_batch)
        return self.logits_processor(input_ids, hidden_states, self.lm_head, forward_batch)
