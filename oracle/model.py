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
    def __init__(self, config, state_dict: Dict[str, torch.Tensor], act_dtype=torch.float32, kv_cache_dtype=None):
        """act_dtype=torch.float32: exact-arithmetic oracle; torch.bfloat16: rounds activations where
        the engine materialises them (closer to the GPU bit pattern).  kv_cache_dtype = torch.float8_e5m2 /
        float8_e4m3fn: rows go through `.to(fp8)` when they enter the cache and `.to(act)` when read
        (mem_cache/memory_pool.py:205-209, 326-336); the tokens of the running forward attend to each
        other unquantised, as extend_attention_fwd reads them from k_extend / v_extend."""
        self.cfg = config
        self.w = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self.act = act_dtype
        self.kv_cache_dtype = kv_cache_dtype
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
                kc, vc = k[sl].view(n, self.Hkv, self.D), v[sl].view(n, self.Hkv, self.D)
                k_all = kc if kv.k[l][b] is None else torch.cat([kv.k[l][b], kc], 0)
                v_all = vc if kv.v[l][b] is None else torch.cat([kv.v[l][b], vc], 0)
                outs.append(_sdpa_per_request(q[sl].view(n, self.Hq, self.D), k_all, v_all, self.g,
                                              self.D ** -0.5).reshape(n, -1))
                if self.kv_cache_dtype is not None:
                    kc, vc = (O.kv_cache_round_trip(kc, self.kv_cache_dtype),
                              O.kv_cache_round_trip(vc, self.kv_cache_dtype))
                kv.append(l, b, kc, vc)
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


class OracleDeepseekV2:
    """DeepSeek-V2(-Lite) forward in the *non-absorbed* form for every step (models/deepseek_v2.py
    forward_normal :591-631 generalised to cached tokens): per-head K/V are re-expanded from the cached
    latent with kv_b_proj, attention is plain MHA.  Mathematically equal to forward_absorb (:633-706),
    so it checks the engine's absorbed decode path independently.  MoE: MoEGate + grouped_topk /
    biased_grouped_topk + naive experts (deepseek_v2.py:119-210, layers/moe/topk.py:79-160,
    fused_moe_native.py:58-131)."""

    def __init__(self, config, state_dict, act_dtype=torch.float32, kv_cache_dtype=None, absorb_fp8=False):
        """kv_cache_dtype = torch.float8_e5m2 / float8_e4m3fn: the latent rows [kv_a | k_pe] go through `.to(fp8)`
        when they enter the cache (MLATokenToKVPool.set_kv_buffer, memory_pool.py:439-452); cached keys / values
        are expanded from the rounded rows, the tokens of the running forward attend to each other unrounded.
        absorb_fp8 (block-quantised models): decode steps run forward_absorb the way the reference's CUDA path does
        (deepseek_v2.py:633-706, 1195-1209): kv_b_proj re-quantised per tensor, q_nope and the latent attention
        output quantised per tensor on the fly (input_to_float8 over the whole [H, T, d] tensor of the step), both
        products through bmm_fp8.  Prefills without a prefix stay in the normal form, as in the engine."""
        self.absorb_fp8 = bool(absorb_fp8)
        self.cfg = config
        self.w = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self.act = act_dtype
        self.kv_cache_dtype = kv_cache_dtype
        c = config
        qc = getattr(c, "quantization_config", None)
        self.block = tuple(qc["weight_block_size"]) if qc else None
        # quantised layers see the activations in the engine's dtype: the group maxima, hence the fp8 codes,
        # depend on it
        self.q_act = torch.bfloat16
        self.H = c.num_attention_heads
        self.nope, self.rope, self.vd, self.lora = c.qk_nope_head_dim, c.qk_rope_head_dim, c.v_head_dim, c.kv_lora_rank
        self.scaling = (self.nope + self.rope) ** -0.5
        rs = c.rope_scaling
        if rs:
            extra = {k: rs[k] for k in ("beta_fast", "beta_slow", "mscale", "mscale_all_dim") if k in rs}
            self.cache = O.deepseek_yarn_cos_sin_cache(self.rope, rs["original_max_position_embeddings"],
                                                       c.rope_theta, rs["factor"], **extra)
            m = O.yarn_get_mscale(rs["factor"], float(rs.get("mscale_all_dim", False)))
            self.scaling = self.scaling * m * m
        else:
            self.cache = O.cos_sin_cache_from_inv_freq(O.rope_inv_freq(self.rope, c.rope_theta),
                                                       c.max_position_embeddings)

    def _lin(self, x, name):
        scale = self.w.get(name + "_scale_inv")
        if scale is not None:
            # block-quantised layer (fp8_utils.py:91-134): activations per token and group of 128, block matmul
            q, s = O.per_token_group_quant_fp8(x.to(self.q_act), self.block[1])
            return O.w8a8_block_fp8_matmul(q, self.w[name], s, scale, self.block, self.q_act).to(self.act)
        return (x.float() @ self.w[name].float().T).to(self.act)

    def _dense(self, name):
        """A weight as a float matrix whether or not it is stored block-quantised (kv_b_proj in _layers)."""
        scale = self.w.get(name + "_scale_inv")
        w = self.w[name].float()
        if scale is None:
            return w
        bn, bk = self.block
        s = scale.repeat_interleave(bn, 0)[: w.shape[0]].repeat_interleave(bk, 1)[:, : w.shape[1]]
        return w * s

    def _mlp(self, x, prefix):
        return self._lin(O.silu_and_mul(self._lin(x, prefix + "gate_up_proj.weight")), prefix + "down_proj.weight")

    def _moe(self, x, p):
        c = self.cfg
        logits = self._lin(x, p + "gate.weight")
        bias = self.w.get(p + "gate.e_score_correction_bias")
        if bias is not None:
            tw, ti = O.biased_grouped_topk(logits, bias.float(), c.num_experts_per_tok, c.norm_topk_prob,
                                           c.n_group, c.topk_group)
        else:
            tw, ti = O.grouped_topk(logits, c.num_experts_per_tok, c.norm_topk_prob, c.n_group, c.topk_group)
        if p + "experts.w13_weight_scale_inv" in self.w:
            out = O.fused_moe_block_fp8(x.to(self.q_act), self.w[p + "experts.w13_weight"], self.w[p + "experts.w2_weight"],
                                        self.w[p + "experts.w13_weight_scale_inv"],
                                        self.w[p + "experts.w2_weight_scale_inv"], tw, ti, self.block).float()
        else:
            out = O.fused_moe(x, self.w[p + "experts.w13_weight"], self.w[p + "experts.w2_weight"], tw, ti)
        out = out * c.routed_scaling_factor
        if c.n_shared_experts is not None:
            out = out + self._mlp(x, p + "shared_experts.").float()
        return out.to(self.act)

    def _absorb_weights(self, name):
        """(W_kc [H, 128, 512] fp8, W_vc [H, 512, 128] fp8, 1 / scale): block_quant_to_tensor_quant of kv_b_proj."""
        if not hasattr(self, "_absorb_cache"):
            self._absorb_cache = {}
        if name not in self._absorb_cache:
            wq = self.w[name]
            if wq.dtype != torch.float8_e4m3fn:   # a state dict widened to fp32 holds the same (fp8-representable) values
                wq = wq.to(torch.float8_e4m3fn)
            wt, ws = O.block_quant_to_tensor_quant(wq, self.w[name + "_scale_inv"].float(), self.block)
            w3 = wt.unflatten(0, (self.H, self.nope + self.vd))
            self._absorb_cache[name] = (w3[:, : self.nope, :], w3[:, self.nope:, :].transpose(1, 2), ws)
        return self._absorb_cache[name]

    def _attend_absorbed(self, p, q, kv_a, k_pe, lens, starts, kv, l):
        """forward_absorb with the per-tensor fp8 products; the cache holds latent rows [kv_a | k_pe]."""
        T = q.shape[0]
        w_kc, w_vc, w_s = self._absorb_weights(p + "self_attn.kv_b_proj.weight")
        q8, q_s = O.input_to_float8(q[..., : self.nope].to(self.q_act).transpose(0, 1), torch.float8_e4m3fn)
        q_abs = O.bmm_fp8(q8, w_kc, q_s, w_s, self.q_act).float()                      # [H, T, 512]
        lat_new = torch.cat([kv_a.float(), k_pe.float()], -1)                             # [T, 576]
        o_lat = torch.empty(T, self.H, self.lora)
        for b, n in enumerate(lens):
            sl = slice(starts[b], starts[b + 1])
            old = kv.lat[l][b]
            lat = lat_new[sl] if old is None else torch.cat([old, lat_new[sl]], 0)
            kv.lat[l][b] = lat
            s = (torch.einsum("hqc,kc->hqk", q_abs[:, sl], lat[:, : self.lora])
                 + torch.einsum("qhr,kr->hqk", q[sl, :, self.nope:].float(), lat[:, self.lora:])) * self.scaling
            Tk = lat.shape[0]
            if n > 1:
                s = s.masked_fill(~torch.ones(n, Tk, dtype=torch.bool).tril(diagonal=Tk - n), float("-inf"))
            o_lat[sl] = torch.einsum("hqk,kc->qhc", torch.softmax(s, -1), lat[:, : self.lora])
        a8, a_s = O.input_to_float8(o_lat.to(self.q_act).transpose(0, 1), torch.float8_e4m3fn)
        out = O.bmm_fp8(a8, w_vc, a_s, w_s, self.q_act)                                   # [H, T, 128]
        return out.transpose(0, 1).reshape(T, -1)

    def _layers(self, h, positions, lens, kv):
        c = self.cfg
        res = None
        starts = [0]
        for n in lens:
            starts.append(starts[-1] + n)
        if self.absorb_fp8 and not hasattr(kv, "lat"):
            kv.lat = [[None] * len(lens) for _ in range(c.num_hidden_layers)]
        # the engine absorbs whenever cached tokens are attended to (decode, later chunks); a first prefill does not
        absorb = self.absorb_fp8 and self.block is not None and any(x is not None for x in kv.k[0])
        for l in range(c.num_hidden_layers):
            p = f"model.layers.{l}."
            if res is None:
                res = h
                x = O.rms_norm(h, self.w[p + "input_layernorm.weight"], c.rms_norm_eps)
            else:
                x, res = O.fused_add_rms_norm(h, res, self.w[p + "input_layernorm.weight"], c.rms_norm_eps)
            T = x.shape[0]
            if p + "self_attn.q_a_proj.weight" in self.w:
                # low-rank query (q_lora_rank; models/deepseek_v2.py: q_a_proj -> q_a_layernorm -> q_b_proj)
                qa = O.rms_norm(self._lin(x, p + "self_attn.q_a_proj.weight"), self.w[p + "self_attn.q_a_layernorm.weight"],
                                c.rms_norm_eps)
                q = self._lin(qa, p + "self_attn.q_b_proj.weight").view(T, self.H, self.nope + self.rope)
            else:
                q = self._lin(x, p + "self_attn.q_proj.weight").view(T, self.H, self.nope + self.rope)
            latent = self._lin(x, p + "self_attn.kv_a_proj_with_mqa.weight")
            kv_a = O.rms_norm(latent[:, : self.lora], self.w[p + "self_attn.kv_a_layernorm.weight"], c.rms_norm_eps)
            q_pe, k_pe = O.apply_rope(positions, q[..., self.nope:].reshape(T, -1), latent[:, self.lora:],
                                      self.rope, self.cache, False)
            q = torch.cat([q[..., : self.nope], q_pe.view(T, self.H, self.rope)], -1)
            if absorb:
                h = self._lin(self._attend_absorbed(p, q, kv_a, k_pe, lens, starts, kv, l).to(self.act),
                              p + "self_attn.o_proj.weight")
                for b, n in enumerate(lens):   # keep the per-head cache of the normal form in step (length bookkeeping)
                    kv.append(l, b, torch.zeros(n, self.H, self.nope + self.rope), torch.zeros(n, self.H, self.vd))
                x, res = O.fused_add_rms_norm(h, res, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
                is_moe = (c.n_routed_experts is not None and l >= c.first_k_dense_replace and l % c.moe_layer_freq == 0)
                h = self._moe(x, p + "mlp.") if is_moe else self._mlp(x, p + "mlp.")
                continue
            # (with absorb_fp8 off, a block-quantised kv_b_proj is used dequantised, like the engine's
            # SEMIPD_MLA_ABSORB_BF16 form and the reference's HIP branch, deepseek_v2.py:655-658, 1195-1249)
            kvb = (kv_a.float() @ self._dense(p + "self_attn.kv_b_proj.weight").T).to(self.act)
            kvb = kvb.view(T, self.H, self.nope + self.vd)
            k = torch.cat([kvb[..., : self.nope], k_pe.view(T, 1, self.rope).expand(T, self.H, self.rope)], -1)
            v = kvb[..., self.nope:]
            if self.kv_cache_dtype is not None:
                rt = lambda t: O.kv_cache_round_trip(t.to(torch.bfloat16), self.kv_cache_dtype).to(t.dtype)  # noqa: E731
                kvb_c = (rt(kv_a).float() @ self._dense(p + "self_attn.kv_b_proj.weight").T).to(self.act)
                kvb_c = kvb_c.view(T, self.H, self.nope + self.vd)
                k_c = torch.cat([kvb_c[..., : self.nope], rt(k_pe).view(T, 1, self.rope).expand(T, self.H, self.rope)], -1)
                v_c = kvb_c[..., self.nope:]
            else:
                k_c, v_c = k, v
            outs = []
            for b, n in enumerate(lens):
                sl = slice(starts[b], starts[b + 1])
                kk = k[sl] if kv.k[l][b] is None else torch.cat([kv.k[l][b], k[sl]], 0)
                vv = v[sl] if kv.v[l][b] is None else torch.cat([kv.v[l][b], v[sl]], 0)
                kv.append(l, b, k_c[sl], v_c[sl])
                if self.absorb_fp8:
                    row = torch.cat([kv_a[sl].float(), k_pe[sl].float()], -1)
                    kv.lat[l][b] = row if kv.lat[l][b] is None else torch.cat([kv.lat[l][b], row], 0)
                s = torch.einsum("qhd,khd->hqk", q[sl].float(), kk.float()) * self.scaling
                Tk = kk.shape[0]
                if n > 1:
                    s = s.masked_fill(~torch.ones(n, Tk, dtype=torch.bool).tril(diagonal=Tk - n), float("-inf"))
                outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv.float()).reshape(n, -1))
            h = self._lin(torch.cat(outs, 0).to(self.act), p + "self_attn.o_proj.weight")
            x, res = O.fused_add_rms_norm(h, res, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            is_moe = (c.n_routed_experts is not None and l >= c.first_k_dense_replace and l % c.moe_layer_freq == 0)
            h = self._moe(x, p + "mlp.") if is_moe else self._mlp(x, p + "mlp.")
        x, _ = O.fused_add_rms_norm(h, res, self.w["model.norm.weight"], c.rms_norm_eps)
        return x

    _logits = OracleLlama._logits
    prefill = OracleLlama.prefill
    decode_step = OracleLlama.decode_step
    generate = OracleLlama.generate
