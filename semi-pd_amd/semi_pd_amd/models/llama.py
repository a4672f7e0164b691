"""Llama-family decoder on the HIP hot-path operators (models/llama.py:60-330 in the reference).

Per layer (models/llama.py:177-203, 237-253):
  input_layernorm (fused add) -> qkv_proj -> RoPE fused with the KV-pool store -> attention backend
  -> o_proj (+TP all-reduce) -> post_attention_layernorm (fused add) -> gate_up_proj -> SiLU*mul
  -> down_proj (+TP all-reduce); final norm; LogitsProcessor.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch
from torch import nn

from semi_pd_amd.distributed import get_tensor_model_parallel_world_size
from semi_pd_amd.layers.attention_backend import RadixAttention
from semi_pd_amd.layers.basic import (LogitsProcessor, MergedColumnParallelLinear, ParallelLMHead,
                                      QKVParallelLinear, RMSNorm, RowParallelLinear, SiluAndMul,
                                      VocabParallelEmbedding, gate_up_silu, get_rope)


@dataclass
class LlamaConfig:
    """The HF config fields the reference reads (models/llama.py:91-131, 223-236)."""
    vocab_size: int = 128256
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: Optional[int] = None
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[Dict[str, Any]] = None
    max_position_embeddings: int = 8192
    tie_word_embeddings: bool = False
    architectures: tuple = ("LlamaForCausalLM",)

    @property
    def head_size(self) -> int:
        return self.head_dim or self.hidden_size // self.num_attention_heads


LLAMA3_8B = LlamaConfig()
LLAMA3_70B = LlamaConfig(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80,
                         num_attention_heads=64, num_key_value_heads=8)


class LlamaMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, dtype):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, params_dtype=dtype)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, params_dtype=dtype)
        self.act_fn = SiluAndMul()

    def forward(self, x):
        # the output goes into the next fused add + RMSNorm (next layer's input_layernorm or the final norm)
        return self.down_proj(gate_up_silu(x, self.gate_up_proj, self.act_fn), defer_reduce=True)


class LlamaAttention(nn.Module):
    def __init__(self, config: LlamaConfig, layer_id: int, dtype):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.head_dim = config.head_size
        self.qkv_proj = QKVParallelLinear(config.hidden_size, self.head_dim, config.num_attention_heads,
                                          config.num_key_value_heads, params_dtype=dtype)
        self.num_heads = self.qkv_proj.num_heads
        self.num_kv_heads = self.qkv_proj.num_kv_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.o_proj = RowParallelLinear(config.num_attention_heads * self.head_dim, config.hidden_size,
                                        params_dtype=dtype)
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, config.max_position_embeddings,
                                   config.rope_theta, True, config.rope_scaling, dtype)
        self.attn = RadixAttention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads, layer_id)
        del tp

    def forward(self, positions, hidden_states, forward_batch):
        pool = forward_batch.token_to_kv_pool
        if forward_batch.forward_mode.is_decode() and self.rotary_emb.supports_planes():
            # decode: the qkv GEMM stops before its K-slice reduction and the RoPE + KV-store kernel sums the planes
            planes = self.qkv_proj.forward_planes(hidden_states)
            if planes is not None:
                backend = forward_batch.attn_backend
                waves = backend.fused_decode_waves(planes.rows, self.head_dim, backend.forward_metadata.num_kv_splits)
                if waves:
                    # ... and the same kernel walks the KV rows and merges its splits: one launch up to o_proj's input
                    attn_output = backend.forward_decode_rope_planes(positions, planes, self.rotary_emb, self.attn,
                                                                     forward_batch, waves)
                    return self.o_proj(attn_output, defer_reduce=True)
                q = self.rotary_emb.forward_and_store_planes(positions, planes, self.num_heads, self.num_kv_heads,
                                                             pool.get_key_buffer(self.attn.layer_id),
                                                             pool.get_value_buffer(self.attn.layer_id),
                                                             forward_batch.out_cache_loc)
                attn_output = self.attn(q, None, None, forward_batch, save_kv_cache=False)
                return self.o_proj(attn_output, defer_reduce=True)
        qkv = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        # RoPE + set_kv_buffer in one launch (rotary_embedding.py:143-169 + memory_pool.py:316-346)
        self.rotary_emb.forward_and_store(positions, q, k, v, pool.get_key_buffer(self.attn.layer_id),
                                          pool.get_value_buffer(self.attn.layer_id), forward_batch.out_cache_loc)
        attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False)
        return self.o_proj(attn_output, defer_reduce=True)   # consumed by post_attention_layernorm(x, residual)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config: LlamaConfig, layer_id: int, dtype):
        super().__init__()
        self.self_attn = LlamaAttention(config, layer_id, dtype)
        self.mlp = LlamaMLP(config.hidden_size, config.intermediate_size, dtype)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, forward_batch, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class LlamaModel(nn.Module):
    def __init__(self, config: LlamaConfig, dtype):
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size, params_dtype=dtype)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, i, dtype) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class LlamaForCausalLM(nn.Module):
    def __init__(self, config: LlamaConfig, dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, dtype)
        if config.tie_word_embeddings:
            self.lm_head = self.model.embed_tokens
        else:
            self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, params_dtype=dtype)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    # geometry the model runner needs for the KV pool and the attention backend
    @property
    def kv_geometry(self):
        a = self.model.layers[0].self_attn
        return dict(kind="mha", num_kv_heads=a.num_kv_heads, head_dim=a.head_dim, v_head_dim=a.head_dim,
                    num_heads=a.num_heads, num_layers=len(self.model.layers))

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.model(input_ids, positions, forward_batch)
        return self.logits_processor(input_ids, hidden_states, self.lm_head, forward_batch)
