"""HF state_dict -> the product's parameter names (TEST INFRASTRUCTURE ONLY).
Plays the role of the reference's load_weights stacked-params mapping (models/llama.py:370-420:
q/k/v -> qkv_proj, gate/up -> gate_up_proj)."""
from __future__ import annotations

from typing import Dict

import torch


def llama_from_hf(sd: Dict[str, torch.Tensor], num_layers: int, tie: bool = False) -> Dict[str, torch.Tensor]:
    out = {"model.embed_tokens.weight": sd["model.embed_tokens.weight"], "model.norm.weight": sd["model.norm.weight"]}
    if not tie:
        out["lm_head.weight"] = sd["lm_head.weight"]
    for i in range(num_layers):
        p = f"model.layers.{i}."
        out[p + "self_attn.qkv_proj.weight"] = torch.cat(
            [sd[p + f"self_attn.{x}_proj.weight"] for x in ("q", "k", "v")], 0)
        out[p + "self_attn.o_proj.weight"] = sd[p + "self_attn.o_proj.weight"]
        out[p + "mlp.gate_up_proj.weight"] = torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0)
        out[p + "mlp.down_proj.weight"] = sd[p + "mlp.down_proj.weight"]
        out[p + "input_layernorm.weight"] = sd[p + "input_layernorm.weight"]
        out[p + "post_attention_layernorm.weight"] = sd[p + "post_attention_layernorm.weight"]
    return out


def opt_from_hf(sd: Dict[str, torch.Tensor], num_layers: int) -> Dict[str, torch.Tensor]:
    d = "model.decoder."
    out = {"embed_tokens.weight": sd[d + "embed_tokens.weight"],
           "embed_positions.weight": sd[d + "embed_positions.weight"],
           "final_layer_norm.weight": sd[d + "final_layer_norm.weight"],
           "final_layer_norm.bias": sd[d + "final_layer_norm.bias"]}
    for i in range(num_layers):
        s, p = d + f"layers.{i}.", f"layers.{i}."
        for wb in ("weight", "bias"):
            out[p + f"qkv_proj.{wb}"] = torch.cat([sd[s + f"self_attn.{x}_proj.{wb}"] for x in ("q", "k", "v")], 0)
            out[p + f"out_proj.{wb}"] = sd[s + f"self_attn.out_proj.{wb}"]
            out[p + f"self_attn_layer_norm.{wb}"] = sd[s + f"self_attn_layer_norm.{wb}"]
            out[p + f"fc1.{wb}"] = sd[s + f"fc1.{wb}"]
            out[p + f"fc2.{wb}"] = sd[s + f"fc2.{wb}"]
            out[p + f"final_layer_norm.{wb}"] = sd[s + f"final_layer_norm.{wb}"]
    return out


def pad_vocab(sd: Dict[str, torch.Tensor], keys, multiple: int = 64) -> Dict[str, torch.Tensor]:
    """The product pads vocab-parallel tables to a multiple of 64 rows (zero rows)."""
    out = dict(sd)
    for k in keys:
        w = out[k]
        pad = (-w.shape[0]) % multiple
        if pad:
            out[k] = torch.cat([w, torch.zeros(pad, w.shape[1], dtype=w.dtype)], 0)
    return out


def deepseek_v2_from_hf(sd: Dict[str, torch.Tensor], cfg) -> Dict[str, torch.Tensor]:
    """HF DeepseekV2ForCausalLM names -> product names (models/deepseek_v2.py load_weights mapping:
    gate/up -> gate_up_proj, experts -> w13_weight / w2_weight)."""
    out = {"model.embed_tokens.weight": sd["model.embed_tokens.weight"], "model.norm.weight": sd["model.norm.weight"],
           "lm_head.weight": sd["lm_head.weight"]}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        q_names = (("self_attn.q_a_proj.weight", "self_attn.q_a_layernorm.weight", "self_attn.q_b_proj.weight")
                   if p + "self_attn.q_a_proj.weight" in sd else ("self_attn.q_proj.weight",))
        for k in q_names + ("self_attn.kv_a_proj_with_mqa.weight", "self_attn.kv_a_layernorm.weight",
                            "self_attn.kv_b_proj.weight", "self_attn.o_proj.weight", "input_layernorm.weight",
                            "post_attention_layernorm.weight"):
            out[p + k] = sd[p + k]
        if p + "mlp.gate.weight" in sd:
            out[p + "mlp.gate.weight"] = sd[p + "mlp.gate.weight"]
            out[p + "mlp.experts.w13_weight"] = sd[p + "mlp.experts.gate_up_proj"]
            out[p + "mlp.experts.w2_weight"] = sd[p + "mlp.experts.down_proj"]
            sp = p + "mlp.shared_experts."
            if sp + "gate_proj.weight" in sd:
                out[sp + "gate_up_proj.weight"] = torch.cat([sd[sp + "gate_proj.weight"], sd[sp + "up_proj.weight"]], 0)
                out[sp + "down_proj.weight"] = sd[sp + "down_proj.weight"]
        else:
            out[p + "mlp.gate_up_proj.weight"] = torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0)
            out[p + "mlp.down_proj.weight"] = sd[p + "mlp.down_proj.weight"]
    return out
