"""CPU oracle models (TEST INFRASTRUCTURE ONLY — see oracle/ops.py header).

OracleLlama / OracleOPT re-state the model forward of the reference
(models/llama.py:177-253 layer order; layers/attention/torch_native_backend.py:26-262 per-request
SDPA over the cached keys/values; layers/logits_processor.py:394-445; layers/sampler.py:72-74) in
plain torch on CPU, driven like python/sglang/bench_one_batch.py:229-256 (one prefill of the whole
batch, then one decode step per new token).  Weights come from a state_dict with the product's
parameter names, so the oracle and the HIP engine run the same numbers.

Pinned by: HF LlamaForCausalLM / OPTForCausalLM on seeded random tiny configs
(tests/test_oracle_models.py), which is the independent cross-check BASELINE.md §3 asks for (the
reference itself has no OPT model and its Llama cannot be imported without vLLM).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from oracle import ops as O


def _sdpa_per_request(q, k_cache, v_cache, g, scale):
    """torch_native_backend.py:_run_sdpa_forward_decode/extend for one request:
    q [Tq,Hq,D], k_cache/v_cache [Tk,Hkv,D]; causal with the query block at the end."""
    Tq, Hq, D = q.shape
    Tk = k_cache.shape[0]
    k = k_cache.repeat_interleave(g, dim=1) if g > 1 else k_cache
    v = v_cache.repeat_interleave(g, dim=1) if g > 1 else v_cache
    s = torch.einsum("qhd,khd->hqk", q.float(), k.float()) * scale
    if Tq > 1:
        mask = torch.ones(Tq, Tk, dtype=torch.bool).tril(diagonal=Tk - Tq)
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, v.float())


class _KVCache:
    def __init__(self, num_layers: int, batch: int):
        self.k = [[None] * batch for _ in range(num_layers)]
        self.v = [[None] * batch for _ in range(num_layers)]

    def append(self, layer, b, k, v):
        self.k[layer][b] = k if self.k[layer][b] is None else torch.cat([self.k[layer][b], k], 0)
        self.v[layer][b] = v if self.v[layer][b] is None else torch.cat([self.v[layer][b], v], 0)


class OracleLlama:
    def __init__(self, config, state_dict: Dict[str, torch.Tensor], act_dtype=torch.float32):
        """act_dtype=torch.float32: exact-arithmetic oracle; torch.bfloat16: rounds activations where
        the engine materialises them (closer to the GPU bit pattern)."""
        self.cfg = config
        self.w = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self.act = act_dtype
        self.D = config.head_size
        self.Hq, self.Hkv = config.num_attention_heads, config.num_key_value_heads
        self.g = self.Hq // self.Hkv
        rs = config.rope_scaling
        if rs and rs.get("rope_type", rs.get("type")) == "llama3":
            inv = O.llama3_inv_freq(self.D, config.rope_theta, rs["factor"], rs["low_freq_factor"],
                                    rs["high_freq_factor"], rs["original_max_position_embeddings"])
        else:
            inv = O.rope_inv_freq(self.D, config.rope_theta)
        self.cache = O.cos_sin_cache_from_inv_freq(inv, config.max_position_embeddings)

    def _lin(self, x, name):
        return (x.float() @ self.w[name].float().T).to(self.act)

    def _layers(self, h, positions, lens: Sequence[int], kv: _KVCache):
        cfg = self.cfg
        res = None
        starts = [0]
        for n in lens:
            starts.append(starts[-1] + n)
        for l in range(cfg.num_hidden_layers):
            p = f"model.layers.{l}."
            if res is None:
                res = h
                x = O.rms_norm(h, self.w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
            else:
                x, res = O.fused_add_rms_norm(h, res, self.w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
            qkv = self._lin(x, p + "self_attn.qkv_proj.weight")
            q, k, v = qkv.split([self.Hq * self.D, self.Hkv * self.D, self.Hkv * self.D], dim=-1)
            q, k = O.apply_rope(positions, q, k, self.D, self.cache, True)
            outs = []
            for b, n in enumerate(lens):
                sl = slice(starts[b], starts[b + 1])
                kv.append(l, b, k[sl].view(n, self.Hkv, self.D), v[sl].view(n, self.Hkv, self.D))
                outs.append(_sdpa_per_request(q[sl].view(n, self.Hq, self.D), kv.k[l][b], kv.v[l][b], self.g,
                                              self.D ** -0.5).reshape(n, -1))
            attn = torch.cat(outs, 0).to(self.act)
            h = self._lin(attn, p + "self_attn.o_proj.weight")
            x, res = O.fused_add_rms_norm(h, res, self.w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
            gu = self._lin(x, p + "mlp.gate_up_proj.weight")
            h = self._lin(O.silu_and_mul(gu), p + "mlp.down_proj.weight")
        x, _ = O.fused_add_rms_norm(h, res, self.w["model.norm.weight"], cfg.rms_norm_eps)
        return x

    def _logits(self, hidden):
        head = self.w.get("lm_head.weight", self.w["model.embed_tokens.weight"])
        return (hidden.float() @ head.float().T)[:, : self.cfg.vocab_size]

    def prefill(self, prompts: Sequence[Sequence[int]]):
        lens = [len(p) for p in prompts]
        ids = torch.tensor([t for p in prompts for t in p], dtype=torch.long)
        pos = torch.cat([torch.arange(n) for n in lens])
        kv = _KVCache(self.cfg.num_hidden_layers, len(prompts))
        h = self.w["model.embed_tokens.weight"][ids].to(self.act)
        hidden = self._layers(h, pos, lens, kv)
        last = torch.cumsum(torch.tensor(lens), 0) - 1
        return self._logits(hidden[last]), kv, lens

    def decode_step(self, tokens: Sequence[int], kv: _KVCache, lens: List[int]):
        ids = torch.tensor(list(tokens), dtype=torch.long)
        pos = torch.tensor(lens, dtype=torch.long)
        h = self.w["model.embed_tokens.weight"][ids].to(self.act)
        hidden = self._layers(h, pos, [1] * len(tokens), kv)
        for i in range(len(lens)):
            lens[i] += 1
        return self._logits(hidden)

    def generate(self, prompts, max_new_tokens: int, forced: Optional[List[List[int]]] = None):
        """Greedy decode.  Returns (tokens, per-step logits [B, steps, V]).  With `forced`, the given
        tokens are fed back instead of the oracle's own argmax (teacher forcing for tie-margin checks)."""
        logits, kv, lens = self.prefill(prompts)
        lens = list(lens)
        all_logits = [logits]
        toks = [[] for _ in prompts]
        cur = []
        for b in range(len(prompts)):
            t = forced[b][0] if forced else int(torch.argmax(logits[b]))
            toks[b].append(t)
            cur.append(t)
        for step in range(1, max_new_tokens):
            logits = self.decode_step(cur, kv, lens)
            all_logits.append(logits)
            cur = []
            for b in range(len(prompts)):
                t = forced[b][step] if forced else int(torch.argmax(logits[b]))
                toks[b].append(t)
                cur.append(t)
        return toks, torch.stack(all_logits, 1)


class OracleOPT:
    """HF OPTForCausalLM semantics with the product's parameter names (models/opt.py)."""

    def __init__(self, config, state_dict, act_dtype=torch.float32):
        self.cfg = config
        self.w = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self.act = act_dtype
        self.H, self.D = config.num_attention_heads, config.head_size

    def _lin(self, x, name):
        return (x.float() @ self.w[name + ".weight"].float().T + self.w[name + ".bias"].float()).to(self.act)

    def _ln(self, x, name):
        return F.layer_norm(x.float(), (x.shape[-1],), self.w[name + ".weight"].float(),
                            self.w[name + ".bias"].float(), 1e-5).to(self.act)

    def _layers(self, h, lens, kv):
        starts = [0]
        for n in lens:
            starts.append(starts[-1] + n)
        for l in range(self.cfg.num_hidden_layers):
            p = f"layers.{l}."
            x = self._ln(h, p + "self_attn_layer_norm")
            qkv = self._lin(x, p + "qkv_proj")
            q, k, v = qkv.split([self.H * self.D] * 3, dim=-1)
            outs = []
            for b, n in enumerate(lens):
                sl = slice(starts[b], starts[b + 1])
                kv.append(l, b, k[sl].view(n, self.H, self.D), v[sl].view(n, self.H, self.D))
                outs.append(_sdpa_per_request(q[sl].view(n, self.H, self.D), kv.k[l][b], kv.v[l][b], 1,
                                              self.D ** -0.5).reshape(n, -1))
            h = h + self._lin(torch.cat(outs, 0).to(self.act), p + "out_proj")
            x = self._ln(h, p + "final_layer_norm")
            h = h + self._lin(F.relu(self._lin(x, p + "fc1")), p + "fc2")
        return self._ln(h, "final_layer_norm")

    def _embed(self, ids, pos):
        return (self.w["embed_tokens.weight"][ids].float() + self.w["embed_positions.weight"][pos + 2].float()).to(self.act)

    def _logits(self, hidden):
        return (hidden.float() @ self.w["embed_tokens.weight"].float().T)[:, : self.cfg.vocab_size]

    def prefill(self, prompts):
        lens = [len(p) for p in prompts]
        ids = torch.tensor([t for p in prompts for t in p], dtype=torch.long)
        pos = torch.cat([torch.arange(n) for n in lens])
        kv = _KVCache(self.cfg.num_hidden_layers, len(prompts))
        hidden = self._layers(self._embed(ids, pos), lens, kv)
        last = torch.cumsum(torch.tensor(lens), 0) - 1
        return self._logits(hidden[last]), kv, lens

    def decode_step(self, tokens, kv, lens):
        ids = torch.tensor(list(tokens), dtype=torch.long)
        pos = torch.tensor(lens, dtype=torch.long)
        hidden = self._layers(self._embed(ids, pos), [1] * len(tokens), kv)
        for i in range(len(lens)):
            lens[i] += 1
        return self._logits(hidden)

    generate = OracleLlama.generate
