"""Checkpoint loading for `--load-format auto`: HF config.json -> model config, *.safetensors shards ->
the product's (stacked, tensor-parallel) parameters.

Reference: model_loader/loader.py (DefaultModelLoader: config + safetensors iterator) and the
per-model `load_weights` stacked-parameter mappings: models/llama.py:370-420 (q/k/v -> qkv_proj,
gate/up -> gate_up_proj), models/deepseek_v2.py:1150-1227 (experts.{e}.{gate,up,down}_proj ->
w13_weight / w2_weight), OPT per HF naming.  The pieces of a stacked parameter are collected on the
host until complete, then the rank's shard is cut with the same `tp_shard` the dummy initialiser
uses, so a loaded model and a dummy one are laid out identically (and exported through IPC the same
way)."""
from __future__ import annotations

import dataclasses
import glob
import json
import os
import re
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import torch
from torch import nn


# ----------------------------------------------------------------------------------- config
def _config_classes():
    from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
    from semi_pd_amd.models.llama import LlamaConfig
    from semi_pd_amd.models.opt import OPTConfig
    return {"LlamaForCausalLM": LlamaConfig, "OPTForCausalLM": OPTConfig,
            "DeepseekV2ForCausalLM": DeepseekV2Config, "DeepseekV3ForCausalLM": DeepseekV2Config}


def config_from_hf_dict(d: dict):
    archs = d.get("architectures") or []
    classes = _config_classes()
    arch = next((a for a in archs if a in classes), None)
    if arch is None:
        raise ValueError(f"unsupported architectures {archs}; supported: {sorted(classes)}")
    cls = classes[arch]
    if arch == "OPTForCausalLM" and d.get("word_embed_proj_dim", d.get("hidden_size")) != d.get("hidden_size"):
        raise ValueError("OPT checkpoints with word_embed_proj_dim != hidden_size are not supported")
    names = {f.name for f in dataclasses.fields(cls)}
    if d.get("quantization_config") and "quantization_config" not in names:
        raise ValueError(f"{arch}: quantised checkpoints are supported for the DeepSeek family only")
    # config keys that change the computation and have no implementation here must not be dropped silently:
    # the checkpoint would load (unmatched bias tensors) and produce wrong logits
    if arch == "LlamaForCausalLM":
        for key in ("attention_bias", "mlp_bias"):
            if d.get(key):
                raise ValueError(f"{arch}: {key}=true checkpoints are not supported (linears run without bias)")
        if d.get("hidden_act", "silu") != "silu":
            raise ValueError(f"{arch}: hidden_act={d['hidden_act']!r} is not supported (SiLU * mul only)")
    if arch in ("DeepseekV2ForCausalLM", "DeepseekV3ForCausalLM") and d.get("attention_bias"):
        raise ValueError(f"{arch}: attention_bias=true checkpoints are not supported")
    kw = {k: v for k, v in d.items() if k in names and k != "architectures"}
    return cls(architectures=(arch,), **kw)


def load_hf_config(model_path: str):
    with open(os.path.join(model_path, "config.json")) as f:
        return config_from_hf_dict(json.load(f))


# ----------------------------------------------------------------------------------- weights
def safetensors_weights_iterator(model_path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """loader.py _get_weights_iterator: every tensor of every *.safetensors shard (host memory)."""
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if not files:
        bins = sorted(glob.glob(os.path.join(model_path, "pytorch_model*.bin")))
        if not bins:
            raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {model_path}")
        for b in bins:
            for k, v in torch.load(b, map_location="cpu", weights_only=True).items():
                yield k, v
        return
    for fpath in files:
        with safe_open(fpath, framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


class _Stacker:
    """Collects the parts of stacked parameters; `add` returns the finished (name, tensor) pairs."""

    def __init__(self):
        self.pending: Dict[str, Dict[int, torch.Tensor]] = {}

    def add(self, name: str, index: int, count: int, t: torch.Tensor, dim: Optional[int] = 0):
        parts = self.pending.setdefault(name, {})
        parts[index] = t
        if len(parts) < count:
            return []
        del self.pending[name]
        ordered = [parts[i] for i in range(count)]
        return [(name, torch.cat(ordered, dim) if dim is not None else torch.stack(ordered, 0))]


_QKV = {"q_proj": 0, "k_proj": 1, "v_proj": 2}
_GATE_UP = {"gate_proj": 0, "up_proj": 1}
_EXPERT_RE = re.compile(r"^(.*\.mlp\.experts)\.(\d+)\.(gate_proj|up_proj|down_proj)\.(weight|weight_scale_inv)$")


def product_items(config, weights: Iterable[Tuple[str, torch.Tensor]]) -> Iterator[Tuple[str, torch.Tensor]]:
    """HF tensor names -> (product parameter name, full unsharded tensor)."""
    arch = config.architectures[0]
    st = _Stacker()
    tie = bool(getattr(config, "tie_word_embeddings", False))
    n_exp = int(getattr(config, "n_routed_experts", 0) or 0)
    expert_parts: Dict[str, Dict[Tuple[int, str], torch.Tensor]] = {}
    for name, t in weights:
        if "rotary_emb.inv_freq" in name:
            continue
        if arch == "OPTForCausalLM":
            if name == "lm_head.weight":
                continue  # tied to embed_tokens
            name = name[len("model.decoder."):] if name.startswith("model.decoder.") else name
            name = name.replace(".self_attn.out_proj.", ".out_proj.")
            m = re.match(r"^(layers\.\d+)\.self_attn\.([qkv]_proj)\.(weight|bias)$", name)
            if m:
                yield from st.add(f"{m.group(1)}.qkv_proj.{m.group(3)}", _QKV[m.group(2)], 3, t)
            else:
                yield name, t
            continue
        if name == "lm_head.weight" and tie:
            continue
        m = _EXPERT_RE.match(name)
        if m:
            prefix, e, which, leaf = m.group(1), int(m.group(2)), m.group(3), m.group(4)
            # block-quantised checkpoints carry experts.{e}.*.weight_scale_inv next to the fp8 weights; they stack
            # like the weights (models/deepseek_v2.py:1150-1227 expert_params_mapping covers both)
            parts = expert_parts.setdefault(prefix + "|" + leaf, {})
            parts[(e, which)] = t
            if len(parts) == 3 * n_exp:
                del expert_parts[prefix + "|" + leaf]
                w13 = torch.stack([torch.cat([parts[(i, "gate_proj")], parts[(i, "up_proj")]], 0)
                                   for i in range(n_exp)], 0)
                w2 = torch.stack([parts[(i, "down_proj")] for i in range(n_exp)], 0)
                yield prefix + ".w13_" + leaf, w13
                yield prefix + ".w2_" + leaf, w2
            continue
        if name.endswith(".mlp.experts.gate_up_proj"):  # checkpoints that already store fused experts
            yield name[: -len("gate_up_proj")] + "w13_weight", t
            continue
        if name.endswith(".mlp.experts.down_proj"):
            yield name[: -len("down_proj")] + "w2_weight", t
            continue
        head, _, leaf = name.rpartition(".")
        mod = head.rsplit(".", 1)[-1]
        base = head[: -len(mod)]
        if mod in _QKV and ".self_attn." in name and arch == "LlamaForCausalLM":
            yield from st.add(f"{base}qkv_proj.{leaf}", _QKV[mod], 3, t)
        elif mod in _GATE_UP and ".experts." not in name:
            yield from st.add(f"{base}gate_up_proj.{leaf}", _GATE_UP[mod], 2, t)
        else:
            yield name, t
    if st.pending or expert_parts:
        missing = list(st.pending) + list(expert_parts)
        raise RuntimeError(f"checkpoint ended with incomplete stacked parameters: {missing[:4]}")


@torch.no_grad()
def load_weights(model: nn.Module, config, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
    """Copy every checkpoint tensor into the model (this rank's shard).  Returns the loaded names and
    raises if a parameter of the model stays uninitialised."""
    params = dict(model.named_parameters(remove_duplicate=False))
    loaded = []
    for name, full in product_items(config, weights):
        if name not in params:
            if "rotary_emb" in name:
                continue  # recomputed caches (inv_freq, cos / sin); behaviour-changing extras are rejected above
            raise KeyError(f"checkpoint tensor {name} has no counterpart in {type(model).__name__}")
        p = params[name]
        if (p.dtype == torch.float8_e4m3fn) != (full.dtype == torch.float8_e4m3fn):
            raise RuntimeError(f"{name}: checkpoint dtype {full.dtype} vs parameter {p.dtype} — block-quantised models "
                               "need a block-quantised checkpoint (and the other way round)")
        src = full.to(p.device, p.dtype)
        if tuple(src.shape) != tuple(p.shape):
            if not hasattr(p, "tp_shard"):
                raise RuntimeError(f"{name}: checkpoint shape {tuple(src.shape)} vs parameter {tuple(p.shape)}")
            src = p.tp_shard(src)
        p.copy_(src)
        loaded.append(name)
    filled = {id(params[n]) for n in loaded}
    # a tied lm_head is the same Parameter object as the embedding it was loaded through
    missing = sorted(n for n, p in params.items() if id(p) not in filled)
    if missing:
        raise RuntimeError(f"parameters not found in the checkpoint: {missing[:8]}{' ...' if len(missing) > 8 else ''}")
    return loaded
