"""OPT decoder (BASELINE.json config 1: OPT-125m) on the HIP attention path.

The reference has no OPT model (python/sglang/srt/models/ holds no opt.py; the closest
learned-position / LayerNorm model is models/gpt2.py:42-230).  Semantics follow HF OPTForCausalLM
(pre-LN, learned positions with offset 2, biased projections, ReLU MLP, tied lm_head), which is also
the parity oracle for this model.  Attention (prefill + paged decode), the KV-pool store and the
logits/argmax run on the HIP kernels; LayerNorm / ReLU / biased GEMMs are plain torch.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import nn

from semi_pd_amd.layers.attention_backend import RadixAttention
from semi_pd_amd.layers.basic import (ColumnParallelLinear, LogitsProcessor, QKVParallelLinear,
                                      RowParallelLinear, VocabParallelEmbedding)


@dataclass
class OPTConfig:
    vocab_size: int = 50272
    hidden_size: int = 768
    ffn_dim: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 2048
    architectures: tuple = ("OPTForCausalLM",)

    @property
    def head_size(self):
        return self.hidden_size // self.num_attention_heads


OPT_125M = OPTConfig()


class OPTDecoderLayer(nn.Module):
    def __init__(self, config: OPTConfig, layer_id: int, dtype):
        super().__init__()
        h, d = config.hidden_size, config.head_size
        self.qkv_proj = QKVParallelLinear(h, d, config.num_attention_heads, config.num_attention_heads,
                                          bias=True, params_dtype=dtype)
        self.out_proj = RowParallelLinear(h, h, bias=True, params_dtype=dtype)
        self.self_attn_layer_norm = nn.LayerNorm(h, dtype=dtype)
        self.fc1 = ColumnParallelLinear(h, config.ffn_dim, bias=True, params_dtype=dtype)
        self.fc2 = RowParallelLinear(config.ffn_dim, h, bias=True, params_dtype=dtype)
        self.final_layer_norm = nn.LayerNorm(h, dtype=dtype)
        self.num_heads = self.qkv_proj.num_heads
        self.head_dim = d
        self.attn = RadixAttention(self.num_heads, d, d ** -0.5, self.num_heads, layer_id)

    def forward(self, hidden_states, forward_batch):
        residual = hidden_states
        x = self.self_attn_layer_norm(hidden_states)
        qkv = self.qkv_proj(x)
        q, k, v = qkv.split([self.num_heads * self.head_dim] * 3, dim=-1)
        x = self.attn(q, k, v, forward_batch, save_kv_cache=True)
        hidden_states = residual + self.out_proj(x)
        residual = hidden_states
        x = self.final_layer_norm(hidden_states)
        x = self.fc2(F.relu(self.fc1(x)))
        return residual + x


class OPTForCausalLM(nn.Module):
    def __init__(self, config: OPTConfig, dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size, params_dtype=dtype)
        self.embed_positions = nn.Embedding(config.max_position_embeddings + 2, config.hidden_size, dtype=dtype)
        self.layers = nn.ModuleList([OPTDecoderLayer(config, i, dtype) for i in range(config.num_hidden_layers)])
        self.final_layer_norm = nn.LayerNorm(config.hidden_size, dtype=dtype)
        self.lm_head = self.embed_tokens  # tied
        self.logits_processor = LogitsProcessor(config.vocab_size)
        for p in self.parameters():
            p.requires_grad_(False)

    @property
    def kv_geometry(self):
        l0 = self.layers[0]
        return dict(kind="mha", num_kv_heads=l0.num_heads, head_dim=l0.head_dim, v_head_dim=l0.head_dim,
                    num_heads=l0.num_heads, num_layers=len(self.layers))

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch):
        hidden_states = self.embed_tokens(input_ids) + self.embed_positions(positions + 2)
        for layer in self.layers:
            hidden_states = layer(hidden_states, forward_batch)
        hidden_states = self.final_layer_norm(hidden_states)
        return self.logits_processor(input_ids, hidden_states, self.lm_head, forward_batch)
