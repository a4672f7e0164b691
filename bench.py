#!/usr/bin/env python
"""Semi-PD serving benchmark (BASELINE.json metric: output tokens/s + p50 TTFT / TBT, Semi-PD mode).

A "step" is one wave of synthetic fixed-length requests with Poisson arrivals served end to end by
the Semi-PD engine (prefill process + decode process per GPU, shared weights / KV, CU split).
Client-side timing follows python/sglang/bench_serving.py:884-970 (TTFT = first token - send,
ITL = gaps between tokens, output tok/s = sum(out) / duration); request generation follows
:771-782 (ids[i][j] = (o_i + i + j) mod vocab) and :896-899 (exponential inter-arrival times).

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (TP = N)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "semi-pd_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# the masked Semi-PD operating point of the default line (see the --prefill-cu / --decode-cu help): nested CU masks like the
# reference's overlapping MPS percentages (semi_pd/utils.py:10-11: prefill 80 %, decode 100 %), work-conserving
# (--cu-mask-mode dynamic).  Shares come in whole groups of 32 CUs (one per shader engine of every XCD), so the choice on 256
# CUs is 192 or 224 for the prefill instance: 224 (88 %) with the decode-step deadline below
DEFAULT_PREFILL_CU = 88
DEFAULT_DECODE_CU = 100
DEFAULT_BACKLOG_FULL_TOKENS = 8192
# decode-step deadline gate (semi_pd/step_pacer.py): a decode step older than this holds the prefill instance at its next
# layer boundary until the step is over.  0 = off
DEFAULT_DEADLINE_MS = 8.5
DEFAULT_TBT_SLO_MS = 12.0
# BASELINE config 2: "Poisson QPS sweep".  The default line sweeps a fixed grid of rates for BOTH engines -- Semi-PD under the
# default policy and the unified engine -- at the headline's lengths (in = 1024 / out = 128), and states the GOODPUT of each:
# the highest rate of the grid that meets a service-level objective (the reference's result form: latency against rate for
# both engines and the rate at which each breaks its SLO, README.md:105, evaluation/show_result.py:50-66; sweep loop
# python/sglang/bench_serving.py:1413-1438).  (SURVEY 8d.2's lambda 2 .. 16 x two length pairs x two policies tables are
# profiles/r05_qps_sweep_config2.txt.)  Points below 16 req/s send 16 s worth of requests instead of all of them.
DEFAULT_SWEEP_RATES = "8,16,24,32,40,48"
SWEEP_OUTPUT_LEN = None          # = --output-len
# the objectives of `goodput` (ms): the 99th percentile of the time to the first token, and either the 99th percentile of
# every gap between two tokens ("itl": streaming smoothness -- what Semi-PD's isolation is about) or the 99th percentile over
# requests of a request's MEAN gap ("tpot": the reference's evaluation metric, show_result.py:36-45, 55-58)
SLO_TTFT_P99_MS = 200.0
SLO_ITL_P99_MS = 15.0
SLO_TPOT_P99_MS = 15.0
# token check of the timed engines (bench_one_batch.py:16-41 `--correct` keeps a known-answer probe for the same purpose):
# requests of these lengths, this many greedy tokens each, Semi-PD against the unified engine; they may part ways only at a
# near-tie among the unified engine's own best log-probabilities (tests/test_gpu_full_depth.py: same margin; here up to
# TOKEN_CHECK_TOP-way: with flat random-weight logits and 128 steps per engine a three-way tie inside the margin was seen)
# sixteen requests sent together: the decode batch reaches 16 requests, from where a Llama decode step takes the fused
# RoPE + attention launch (layers/attention_backend.py: fused_decode_waves) -- the check must run what the timed waves ran
TOKEN_CHECK_LENS = (64, 200, 1024, 7) * 4
TOKEN_CHECK_STEPS = 8
TOKEN_CHECK_TOP = 8     # log-probabilities the reference engine returns per step: a near-tie may be three- or four-way
# The margin: the two engines run bf16 kernels on the same random weights but batch the prompts differently (the unified
# engine takes the sixteen prompts as one 5.2 k-token chunk, the Semi-PD prefill instance as whatever had arrived), so the
# GEMMs behind a token differ in tile shape and summation order.  With N(0, 0.02) weights the logits of a step have a
# standard deviation of ~1.3 and its best candidates sit 0.0-0.3 apart; the engines' logits differ by ~0.05 (one sigma)
# after 32 layers.  Over 128 steps per engine gaps of up to 0.149 were seen at an accepted flip
# (profiles/r06_bench_n1_default_line_v3.json: 7 of 16 requests parted, all within the reference's top 3); 0.15 was the margin
# of the four-request check of the round's first lines (no flip seen in 32 steps).  0.3 is ~5 sigma: a wrong KV row, a
# mis-rotated head or a dropped K slice moves the logits by O(1) and lands outside the reference's top 8 altogether.
TOKEN_CHECK_MARGIN = 0.3

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def model_config(name: str):
    from semi_pd_amd.models.llama import LLAMA3_8B, LLAMA3_70B, LlamaConfig
    from semi_pd_amd.models.opt import OPT_125M
    if name == "llama3-8b":
        return LLAMA3_8B
    if name == "llama3-70b":
        return LLAMA3_70B
    if name == "opt-125m":
        return OPT_125M
    if name == "deepseek-v2-lite":
        from semi_pd_amd.models.deepseek_v2 import DEEPSEEK_V2_LITE
        return DEEPSEEK_V2_LITE
    if name == "deepseek-v3-slice":
        # DeepSeek-V3 geometry (hidden 7168, 128 heads, q_lora 1536, kv_lora 512, 256 experts top-8 in
        # 8 groups top-4, moe_inter 2048, sigmoid routing with bias) cut to 1 dense + 3 MoE layers so that it
        # fits one GPU: a shape check of config 5's kernels, not a headline number
        from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
        return DeepseekV2Config(
            vocab_size=129280, hidden_size=7168, intermediate_size=18432, moe_intermediate_size=2048,
            num_hidden_layers=4, num_attention_heads=128, n_shared_experts=1, n_routed_experts=256,
            num_experts_per_tok=8, routed_scaling_factor=2.5, topk_method="noaux_tc", n_group=8, topk_group=4,
            norm_topk_prob=True, first_k_dense_replace=1, kv_lora_rank=512, q_lora_rank=1536, qk_rope_head_dim=64,
            qk_nope_head_dim=128, v_head_dim=128, rope_theta=10000.0,
            rope_scaling={"type": "yarn", "factor": 40, "beta_fast": 32, "beta_slow": 1, "mscale": 1.0,
                          "mscale_all_dim": 1.0, "original_max_position_embeddings": 4096},
            architectures=("DeepseekV3ForCausalLM",))
    if name == "deepseek-v3-tp8-rank":
        # what ONE rank of BASELINE config 5 (DeepSeek-V3, block-fp8, TP = 8) holds and computes, without the collectives:
        # all 61 layers at hidden 7168 with 128 / 8 = 16 MLA heads, every MLP / expert intermediate width / 8 (dense 2304,
        # experts 256: the reference's column split, fused_moe_triton/layer.py), 256 routed experts top-8 in 8 groups, 1 / 8
        # of the vocabulary: ~85 GB of fp8 weights on one GPU.  Use with --quantization fp8.  A one-GPU measurement of the
        # per-rank kernel shapes, not a claim about config 5
        from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
        return DeepseekV2Config(
            vocab_size=16160, hidden_size=7168, intermediate_size=2304, moe_intermediate_size=256,
            num_hidden_layers=61, num_attention_heads=16, n_shared_experts=1, n_routed_experts=256,
            num_experts_per_tok=8, routed_scaling_factor=2.5, topk_method="noaux_tc", n_group=8, topk_group=4,
            norm_topk_prob=True, first_k_dense_replace=3, kv_lora_rank=512, q_lora_rank=1536, qk_rope_head_dim=64,
            qk_nope_head_dim=128, v_head_dim=128, rope_theta=10000.0,
            rope_scaling={"type": "yarn", "factor": 40, "beta_fast": 32, "beta_slow": 1, "mscale": 1.0,
                          "mscale_all_dim": 1.0, "original_max_position_embeddings": 4096},
            architectures=("DeepseekV3ForCausalLM",))
    if name == "llama3-70b-tp8-rank":
        # what ONE rank of BASELINE config 4 (Llama-3-70B, TP = 8) computes, without the collectives: 80 layers of hidden
        # 8192 with 1/8 of the heads (8 q, 1 kv, head size 128), 1/8 of the MLP columns (3584) and 1/8 of the vocabulary:
        # 17.4 GB of weights per decode step = 2.2 ms at 8 TB/s.  A one-GPU measurement of the per-rank kernel shapes and
        # launch count (9 per layer x 80), not a claim about config 4
        return LlamaConfig(vocab_size=16032, hidden_size=8192, intermediate_size=3584, num_hidden_layers=80,
                           num_attention_heads=8, num_key_value_heads=1, head_dim=128, max_position_embeddings=8192)
    if name == "llama-tiny":
        return LlamaConfig(vocab_size=32000, hidden_size=1024, intermediate_size=2816, num_hidden_layers=4,
                           num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=8192)
    raise ValueError(name)


def make_requests(num, input_len, vocab, seed):
    """bench_serving.py:771-782 random-ids rule."""
    rs = np.random.RandomState(seed)
    offsets = rs.randint(0, vocab, size=num)
    return [[int((offsets[i] + i + j) % vocab) for j in range(input_len)] for i in range(num)]


def arrival_times(num, rate, seed):
    if rate <= 0 or rate == float("inf"):
        return np.zeros(num)
    rs = np.random.RandomState(seed + 1)
    return np.cumsum(rs.exponential(1.0 / rate, size=num)) - 0.0


def run_wave(engine, prompts, arrivals, output_len):
    """Send requests on their Poisson schedule, stream tokens, return per-request records."""
    from semi_pd_amd.managers.io_struct import SamplingParams
    t0 = time.time()
    rids = []
    nxt = 0
    n = len(prompts)
    done = 0
    while done < n:
        now = time.time() - t0
        while nxt < n and arrivals[nxt] <= now:
            rids.append(engine.add_request(prompts[nxt], SamplingParams(max_new_tokens=output_len, ignore_eos=True)))
            nxt += 1
        wait = 0.0005 if nxt >= n else min(0.0005, max(0.0, arrivals[nxt] - now))
        if not engine.poll(timeout=wait):
            engine.check_children()
        done = sum(1 for r in rids if engine._finished[r] is not None)
    dur = time.time() - t0
    recs = [engine.request_record(r) for r in rids]
    return recs, dur


def combine_ranks(records, elapsed, rank, world, replicas):
    """N > 1: the timed region is as long as the slowest rank's (MAX over ranks); with independent replicas rank 0
    also gets every replica's request records, so that `value` counts the tokens of the whole job and the latency
    percentiles run over all requests.  (A tensor-parallel engine is driven by rank 0 alone: nothing to gather.)"""
    if world <= 1:
        return records, elapsed
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if replicas:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(records, gathered, dst=0)
        if rank == 0:
            records = [r for part in gathered for r in part]
    return records, elapsed


def summarize(records, duration):
    ttft, itl, tpot, out_tokens = [], [], [], 0
    for r in records:
        tt = r["token_times"]
        out_tokens += len(r["output_ids"])
        if tt:
            ttft.append(tt[0] - r["send"])
            gaps = np.diff(tt)
            itl.extend(gaps.tolist())
            if len(gaps):
                tpot.append(float(gaps.mean()))     # a request's mean gap (show_result.py:36-45)
    return {"output_tokens": out_tokens, "duration_s": duration,
            "output_tok_s": out_tokens / duration if duration > 0 else 0.0,
            "p50_ttft_ms": float(np.median(ttft) * 1e3) if ttft else None,
            "p99_ttft_ms": float(np.percentile(ttft, 99) * 1e3) if ttft else None,
            "p50_tbt_ms": float(np.median(itl) * 1e3) if itl else None,
            "p99_tbt_ms": float(np.percentile(itl, 99) * 1e3) if itl else None,
            "p99_tpot_ms": float(np.percentile(tpot, 99) * 1e3) if tpot else None}


def _rounded(sm):
    return {k: (round(v, 2) if isinstance(v, float) else v) for k, v in sm.items()}


def sweep_point(sm, rate, n, input_len, output_len, source=None):
    """One row of a rate sweep, with the verdicts of the two objectives."""
    ttft_ok = sm["p99_ttft_ms"] is not None and sm["p99_ttft_ms"] <= SLO_TTFT_P99_MS
    row = {"request_rate": rate, "num_requests": n, "input_len": input_len, "output_len": output_len, **_rounded(sm),
           "meets_slo_itl": bool(ttft_ok and sm["p99_tbt_ms"] is not None and sm["p99_tbt_ms"] <= SLO_ITL_P99_MS),
           "meets_slo_tpot": bool(ttft_ok and sm["p99_tpot_ms"] is not None and sm["p99_tpot_ms"] <= SLO_TPOT_P99_MS)}
    if source:
        row["source"] = source
    return row


def goodput_of(points, key):
    """The highest rate of the grid whose point meets the objective (0 when none does)."""
    ok = [p["request_rate"] for p in points if p.get(key)]
    return max(ok) if ok else 0.0


def sweep_requests(rate, num_requests):
    """Requests of one sweep point: all of them from 16 req/s up, 16 s worth below (a 256-request wave at 8 req/s is 32 s)."""
    return max(1, min(num_requests, int(round(16 * rate)))) if rate > 0 else num_requests


def token_probe(engine, vocab, seed, logprobs=False):
    """TOKEN_CHECK_STEPS greedy tokens for requests of TOKEN_CHECK_LENS tokens (ids by bench_serving.py's rule)."""
    from semi_pd_amd.managers.io_struct import SamplingParams
    rs = np.random.RandomState(seed + 77)
    prompts = [[int((o + j) % vocab) for j in range(n)] for o, n in zip(rs.randint(0, vocab, size=len(TOKEN_CHECK_LENS)), TOKEN_CHECK_LENS)]
    sp = SamplingParams(max_new_tokens=TOKEN_CHECK_STEPS, ignore_eos=True)
    if logprobs:
        return engine.generate(prompts, sp, timeout=300, return_logprob=True, top_logprobs_num=TOKEN_CHECK_TOP)
    return engine.generate(prompts, sp, timeout=300), None


def compare_tokens(name, got, ref, ref_lps):
    """`got` (an engine under test) against `ref` (the unified engine, with its TOKEN_CHECK_TOP best log-probabilities per step):
    equal, or parting at a step where the other token is one the reference rates within TOKEN_CHECK_MARGIN of its own."""
    out = {"engine": name, "requests": len(ref), "tokens_per_request": TOKEN_CHECK_STEPS, "equal_requests": 0,
           "near_tie_divergences": [], "errors": []}
    for i, (a, b) in enumerate(zip(ref, got)):
        if list(a) == list(b):
            out["equal_requests"] += 1
            continue
        if len(a) != len(b):
            out["errors"].append(f"request {i}: {len(b)} tokens instead of {len(a)}")
            continue
        s = next(j for j in range(len(a)) if a[j] != b[j])
        try:
            top = [(float(lp), int(t)) for lp, t in (e[:2] for e in ref_lps[i]["top"][s])]   # the reference's best tokens, best first
            lp1, t1 = top[0]
        except Exception as e:   # no log-probabilities to judge the divergence by: it counts as an error
            out["errors"].append(f"request {i} step {s}: {a[s]} vs {b[s]} and no top log-probabilities ({e!r})")
            continue
        # the engine's token must be one the reference itself rates within the margin of its own choice (its runner-up, or
        # -- flat random-weight logits, 128 steps per engine -- the third or fourth of a several-way near-tie)
        rank = next((r for r, (_, t) in enumerate(top) if t == b[s]), None)
        gap = lp1 - top[rank][0] if rank is not None else float("inf")
        if t1 == a[s] and rank is not None and rank >= 1 and gap < TOKEN_CHECK_MARGIN:
            out["near_tie_divergences"].append({"request": i, "step": s, "rank_in_reference": rank + 1, "logprob_gap": round(gap, 4)})
        else:
            out["errors"].append(f"request {i} step {s}: unified {a[s]} (its top {len(top)}: "
                                 + ", ".join(f"{t} {lp1 - lp:.4f}" for lp, t in top) + f"), {name} {b[s]}")
    out["ok"] = not out["errors"]
    return out


def prefill_accounting(s: dict, n: int, num_layers: int) -> dict:
    """Where a prefill batch's time on the GPU goes (ms per batch, means over the timed steps):
      gpu_owned            from the moment the batch has the GPU (its launch, or the end of the batch it was queued behind) to
                           its ids on the host -- what the prefill queue's service time is
      layers_without_hold  num_layers x the mean GPU time between two consecutive layer hooks at which no hold happened: one
                           decoder layer's kernels plus the launch gaps inside it (HIP events of the step pacer's run-ahead
                           bound; the per-kernel split is the prefill process's rocprofv3 table under profiles/)
      held_gpu_idle        what the decode-step deadline's holds cost the batch ON THE GPU: (hook intervals with a hold per
                           batch) x (their mean length - the mean length of an interval without one)
      outside_layers       the rest: embedding, final norm, lm_head + sampling of the last tokens, the copy of the ids, and
                           whatever the GPU idled between this batch and its predecessor
      held_host            wall time the hooks spent in holds (includes the tail of the layer still running when a hold began)
      pacer_wait_host      host time inside the hooks waiting for the GPU (bounded run-ahead): NOT GPU time, listed so that
                           nobody adds it"""
    if not s.get("t_gpu_owned_s"):
        return {}
    gate = s.get("step_gate") or {}
    owned = 1e3 * s["t_gpu_owned_s"] / n
    out = {"gpu_owned": round(owned, 3)}
    if gate.get("layer_ms_without_hold"):
        per_layer = gate["layer_ms_without_hold"]
        layers = num_layers * per_layer
        timed = gate.get("layer_intervals_timed", 0) + gate.get("layer_intervals_with_hold", 0)
        # (the last run-ahead intervals of a forward are not timed: scale the count of held intervals to all hooks)
        held_per_batch = gate.get("layer_intervals_with_hold", 0) * (gate.get("gates", timed) / max(timed, 1)) / n
        held_gpu = held_per_batch * max(0.0, gate.get("layer_ms_with_hold", per_layer) - per_layer)
        out.update({"layers_without_hold": round(layers, 3), "held_gpu_idle": round(held_gpu, 3),
                    "outside_layers": round(owned - layers - held_gpu, 3)})
    if gate:
        out["held_host"] = round(gate.get("held_ms", 0.0) / n, 3)
    if gate.get("run_ahead_waits_ms") is not None:
        out["pacer_wait_host"] = round(gate["run_ahead_waits_ms"] / n, 3)
    return out


def pmc_traffic(kernel: str, algorithmic_bytes: float) -> dict:
    """`traffic` of the roofline object: HBM bytes per launch from the committed PMC pass of the kernel (its measured
    traffic / algorithmic ratio times this launch's algorithmic bytes), with the file it comes from; null without one."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        for name, rec in table.items():
            if name in kernel:
                # (not counters of THIS run: the ratio measured by the committed rocprofv3 --pmc pass of the same kernel,
                #  times this launch's algorithmic bytes -- `traffic_estimated` says so)
                return {"traffic": int(rec["traffic_over_algorithmic"] * algorithmic_bytes), "traffic_estimated": True,
                        "traffic_over_algorithmic": rec["traffic_over_algorithmic"], "traffic_source": rec["source"]}
    except (OSError, ValueError, KeyError):
        pass
    return {"traffic": None}


def cpu_baseline(cfg, input_len, output_len, budget_s=30.0):
    """The CPU oracle (oracle/model.py, a port of the reference's torch_native path) timed on this host, on a bounded
    sample of the same workload, at the model's FULL depth: every one of the num_layers layers is executed (fp32; the
    layers share one layer's seeded weights -- 0.9 GB resident instead of 28 GB, the arithmetic and the memory traffic
    per layer are those of distinct weights bigger than any cache), embedding + final norm + lm_head at their real
    sizes; 2 requests of the workload's input length (1 if a one-layer probe says 2 would not fit the budget), one
    prefill, then 3 decode steps, driven like bench_one_batch.py; only the number of decode steps is extrapolated
    (a decode step's cost grows by < 1 % over the output length at these context sizes)."""
    from oracle.model import OracleLlama
    import copy
    # torch's intra-op pool stops scaling (and with 256 hardware threads collapses: 3.2 s per decode
    # step of ONE layer) well before a big host's core count: use at most 32 threads and say so
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    H, I, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_size
    L = cfg.num_hidden_layers
    layer = {"input_layernorm.weight": torch.ones(H),
             "post_attention_layernorm.weight": torch.ones(H),
             "self_attn.qkv_proj.weight": torch.randn((cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * D, H, generator=g) * 0.02,
             "self_attn.o_proj.weight": torch.randn(H, cfg.num_attention_heads * D, generator=g) * 0.02,
             "mlp.gate_up_proj.weight": torch.randn(2 * I, H, generator=g) * 0.02,
             "mlp.down_proj.weight": torch.randn(H, I, generator=g) * 0.02}
    sd = {"model.embed_tokens.weight": torch.randn(cfg.vocab_size, H, generator=g) * 0.02,
          "lm_head.weight": torch.randn(cfg.vocab_size, H, generator=g) * 0.02,
          "model.norm.weight": torch.ones(H)}
    for l in range(L):
        for k, v in layer.items():
            sd[f"model.layers.{l}.{k}"] = v          # the same storage under every layer's name
    # one-layer probe (one request): does the whole depth with 2 requests fit the budget?
    c1 = copy.copy(cfg)
    c1.num_hidden_layers = 1
    probe = OracleLlama(c1, sd)
    probe.prefill(make_requests(1, input_len, cfg.vocab_size, 3))    # first call: thread pool, allocator
    t0 = time.time()
    probe.prefill(make_requests(1, input_len, cfg.vocab_size, 3))
    t_probe = time.time() - t0
    nreq = 2 if 2 * t_probe * L * 1.15 <= budget_s else 1
    steps = 3
    oracle = OracleLlama(cfg, sd)
    prompts = make_requests(nreq, input_len, cfg.vocab_size, 3)
    t0 = time.time()
    logits, kv, lens = oracle.prefill(prompts)
    t_prefill = time.time() - t0
    lens = list(lens)
    cur = [int(torch.argmax(l)) for l in logits]
    t0 = time.time()
    for _ in range(steps):
        logits = oracle.decode_step(cur, kv, lens)
        cur = [int(torch.argmax(l)) for l in logits]
    t_decode = (time.time() - t0) / steps
    total = t_prefill + t_decode * (output_len - 1)
    return {"value": nreq * output_len / total, "unit": "output tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/model.py OracleLlama fp32, all {L} layers executed (one layer's weights under every layer's "
                      f"name) + embedding + lm_head; {nreq} request(s) in={input_len}, one prefill + {steps} decode steps "
                      f"measured, the decode step extrapolated to out={output_len}; measured prefill {t_prefill:.2f}s, "
                      f"decode step {t_decode * 1e3:.0f}ms (one-layer probe {t_probe:.2f}s)"}


def cpu_baseline_opt(cfg, num_requests, input_len, output_len, seed):
    """BASELINE config 1 end to end on the host: the CPU oracle of the OPT model (oracle/model.py OracleOPT, HF
    semantics, fp32) with the same seeded dummy weights and the same prompts serves ALL requests as one batch --
    one prefill, then output_len - 1 greedy decode steps (bench_one_batch.py:229-256 drives the reference like this)."""
    from oracle.model import OracleOPT
    from semi_pd_amd.model_executor.model_runner import build_model, dummy_init_weights
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    model = build_model(cfg, torch.float32)
    dummy_init_weights(model, torch.device("cpu"), seed)
    sd = {k: v.float() for k, v in model.state_dict().items()}
    oracle = OracleOPT(cfg, sd)
    prompts = make_requests(num_requests, input_len, cfg.vocab_size, seed)
    t0 = time.time()
    logits, kv, lens = oracle.prefill(prompts)
    t_prefill = time.time() - t0
    lens = list(lens)
    cur = [int(torch.argmax(l)) for l in logits]
    t0 = time.time()
    for _ in range(output_len - 1):
        logits = oracle.decode_step(cur, kv, lens)
        cur = [int(torch.argmax(l)) for l in logits]
    t_decode = time.time() - t0
    total = t_prefill + t_decode
    return {"value": num_requests * output_len / total, "unit": "output tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/model.py OracleOPT fp32, the WHOLE workload: {num_requests} requests in={input_len} "
                      f"out={output_len} as one batch; prefill {t_prefill:.2f}s, {output_len - 1} decode steps "
                      f"{t_decode:.2f}s ({1e3 * t_decode / max(output_len - 1, 1):.0f} ms per step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--num-requests", type=int, default=256)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--output-len", type=int, default=128)
    ap.add_argument("--request-rate", type=float, default=32.0, help="Poisson arrivals per second (0 = all at once)")
    ap.add_argument("--mode", choices=["semi-pd", "unified"], default="semi-pd")
    # CU shares of the two instances (prefill takes its share from the bottom of the CU range, decode from the top).  The
    # default is a MASKED policy -- the north star's compute isolation and BASELINE config 2 ("CU split") -- with the
    # nested shares, prefill 88 % / decode 100 % (the decode instance has the top 32 CUs to itself and may
    # use the rest), work-conserving (--cu-mask-mode dynamic; profiles/r04_policy_sweep.txt).  `--prefill-cu 50 --decode-cu 50
    # --cu-mask-mode env` is config 2's split as written: it is measured in the same invocation by a second engine (one
    # warm-up + one timed wave, "static_split_50_50"; --no-static-split-wave skips it); `--prefill-cu 100 --decode-cu 100`
    # is the unmasked round-2 default.
    ap.add_argument("--prefill-cu", type=int, default=DEFAULT_PREFILL_CU)
    ap.add_argument("--decode-cu", type=int, default=DEFAULT_DECODE_CU)
    ap.add_argument("--no-static-split-wave", action="store_true",
                    help="N = 1 Semi-PD default run: do not start the second engine with BASELINE config 2's literal 50 / 50 split")
    ap.add_argument("--no-unified-wave", action="store_true",
                    help="N = 1 Semi-PD run: do not start the unified engine on the same requests afterwards (unified_same_load)")
    ap.add_argument("--no-prefill-gemm-tuning", action="store_true",
                    help="prefill instance: the library's own GEMM choice instead of the solutions timed on its CU share")
    ap.add_argument("--tune-prefill-gemm", action="store_true",
                    help="time the library's GEMM solutions at start-up even when the prefill instance owns every CU")
    ap.add_argument("--cu-mask-mode", default="dynamic", choices=["env", "none", "dynamic"],
                    help="env: static HSA_CU_MASK per instance; dynamic: unmasked processes with a CU-masked stream over their "
                         "share and a stream over every CU, chosen per decode step / prefill batch (work-conserving shares)")
    ap.add_argument("--prefill-backlog-full-tokens", type=int, default=DEFAULT_BACKLOG_FULL_TOKENS,
                    help="dynamic mode: waiting prompt tokens from which a prefill batch takes every CU (0 = never)")
    ap.add_argument("--decode-step-deadline-ms", type=float, default=DEFAULT_DEADLINE_MS,
                    help="Semi-PD: a decode step older than this makes the prefill instance yield at its next layer boundary "
                         "until the step is over (0 = no gate)")
    ap.add_argument("--decode-tbt-slo-ms", type=float, default=DEFAULT_TBT_SLO_MS,
                    help="with a deadline: adapt it so that the 99th percentile of the time between tokens meets this (0 = fixed)")
    ap.add_argument("--prefill-priority", type=int, default=0, help="HIP stream priority of the prefill instance (-1 = high)")
    ap.add_argument("--decode-priority", type=int, default=0, help="HIP stream priority of the decode instance (-1 = high)")
    ap.add_argument("--disable-stream-linear", action="store_true",
                    help="decode-batch dense layers through hipBLASLt instead of the persistent streaming kernel (A/B)")
    ap.add_argument("--library-gemm-grid", action="store_true",
                    help="size hipBLASLt's stream-K grids to each instance's CU share (TENSILE_STREAMK_MAX_CUS)")
    ap.add_argument("--context-length", type=int, default=0)
    ap.add_argument("--max-running-requests", type=int, default=256)
    ap.add_argument("--mem-fraction-static", type=float, default=None)
    ap.add_argument("--max-total-tokens", type=int, default=None)
    ap.add_argument("--chunked-prefill-size", type=int, default=None, help="tokens per prefill batch (default 8192)")
    ap.add_argument("--disable-cuda-graph", action="store_true")
    ap.add_argument("--kv-splits", type=int, default=None,
                    help="fixed split-KV count of the decode attention (--triton-attention-num-kv-splits); default: per batch")
    ap.add_argument("--disable-overlap-schedule", action="store_true",
                    help="plain (not overlapped / pipelined) decode and prefill loops, for A/B runs")
    ap.add_argument("--quantization", default=None, choices=[None, "fp8"],
                    help="fp8 = block-scaled e4m3fn weights (128 x 128) and per-token-group activations, DeepSeek family")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e5m2", "fp8_e4m3"],
                    help="KV pool rows: activation type (the measured default) or OCP fp8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-token-check", action="store_true",
                    help="N = 1 Semi-PD run: skip the comparison of the timed engines' first tokens with the unified engine's")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fixed-load", action="store_true", help="N > 1 (tensor parallel): do not scale requests / rate with N")
    ap.add_argument("--tp", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: N independent Semi-PD replicas, one per GPU (no data-path collective), instead of ONE "
                         "engine tensor-parallel over the N GPUs")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="default Llama-3-8B line: skip the one-wave runs of BASELINE configs 1 (OPT-125m, with its CPU "
                         "baseline) and 3 (DeepSeek-V2-Lite)")
    ap.add_argument("--no-saturation-wave", action="store_true",
                    help="skip the extra (untimed for `value`) wave with all requests sent at once")
    ap.add_argument("--rate-sweep", default=None, help="comma-separated Poisson rates; one extra (untimed for "
                    "`value`) wave per rate after the timed steps, reported under qps_sweep (BASELINE config 2) with the goodput "
                    "under the objectives named in config.slo; the unified engine of the default line runs the same grid "
                    f"(qps_sweep_unified).  Default: {DEFAULT_SWEEP_RATES} for the default N = 1 Llama-3-8B workload, none "
                    "otherwise; '' = none")
    ap.add_argument("--sweep-output-len", type=int, default=SWEEP_OUTPUT_LEN, help="default: --output-len")
    ap.add_argument("--sweep-num-requests", type=int, default=None, help="requests per sweep point (default: --num-requests)")
    args = ap.parse_args()

    if args.rate_sweep is None:
        default_workload = (args.gpus == 1 and args.model == "llama3-8b" and args.mode == "semi-pd"
                            and args.input_len == 1024 and args.output_len == 128 and args.request_rate == 32.0)
        args.rate_sweep = DEFAULT_SWEEP_RATES if default_workload else ""
    if args.sweep_output_len is None:
        args.sweep_output_len = args.output_len
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SEMIPD_BENCH_ALL_ON_GPU0") == "1":
        local_rank = 0  # functional check of the N > 1 code path on a one-GPU box (replicas share the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # N > 1, default: ONE engine, tensor parallel over the N GPUs (heads and MLP columns sharded, an
    # all-reduce of [tokens, hidden] after o_proj and down_proj of every layer over RCCL / the peer-memory kernels),
    # serving N times the offered load unless --fixed-load ("scaling": "weak").  --replicas: N independent Semi-PD
    # engines, one per GPU, each with the N = 1 workload (no data-path collective).
    args.tp = world > 1 and not args.replicas
    tp_world = world if args.tp else 1
    replica = rank if (world > 1 and not args.tp) else 0
    if world > 1 and args.tp and not args.fixed_load:
        args.num_requests *= world
        args.request_rate *= world
        args.max_running_requests *= world
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)  # launcher-level barrier / max only

    def barrier():
        if world > 1:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.server_args import ServerArgs
    cfg = model_config(args.model)
    if args.quantization == "fp8":
        import dataclasses
        extra = {}
        if args.model == "deepseek-v2-lite":
            # the dense FFN of layer 0 is 10944 wide = 85.5 quantisation groups; whole groups need 11008 (the
            # reference cannot block-quantise that layer either, fp8_kernel.py:183-186)
            extra["intermediate_size"] = 11008
        cfg = dataclasses.replace(cfg, quantization_config={"quant_method": "fp8", "weight_block_size": [128, 128],
                                                            "activation_scheme": "dynamic"}, **extra)
    sweep_rates = [float(x) for x in args.rate_sweep.split(",") if x]
    ctx = args.context_length or (args.input_len + max(args.output_len, args.sweep_output_len if sweep_rates else 0) + 8)
    port_base = int(os.environ.get("MASTER_PORT", "29500")) + 100
    sa = ServerArgs(model_config=cfg, context_length=ctx, tp_size=tp_world, enable_semi_pd=(args.mode == "semi-pd"),
                    max_running_requests=args.max_running_requests, mem_fraction_static=args.mem_fraction_static,
                    max_total_tokens=args.max_total_tokens, prefill_cu_percent=args.prefill_cu,
                    decode_cu_percent=args.decode_cu, cu_mask_mode=args.cu_mask_mode,
                    prefill_backlog_full_tokens=args.prefill_backlog_full_tokens,
                    decode_step_deadline_ms=args.decode_step_deadline_ms,
                    decode_tbt_slo_ms=(args.decode_tbt_slo_ms if args.decode_step_deadline_ms > 0 else 0.0),
                    library_gemm_grid=args.library_gemm_grid, disable_stream_linear=args.disable_stream_linear,
                    tune_prefill_gemm=(False if args.no_prefill_gemm_tuning else (True if args.tune_prefill_gemm else None)),
                    prefill_stream_priority=args.prefill_priority, decode_stream_priority=args.decode_priority,
                    disable_cuda_graph=args.disable_cuda_graph, nccl_port_base=port_base,
                    disable_overlap_schedule=args.disable_overlap_schedule,
                    triton_attention_num_kv_splits=args.kv_splits,
                    **({"chunked_prefill_size": args.chunked_prefill_size} if args.chunked_prefill_size else {}),
                    kv_cache_dtype=args.kv_cache_dtype,
                    # (the one-GPU check of the TP path reduces over gloo, whose collectives cannot be captured beyond the
                    # peer-memory kernels' 16 MB: small graphs there)
                    cuda_graph_max_bs=(64 if os.environ.get("SEMIPD_BENCH_ALL_ON_GPU0") == "1" else min(1024, 256 * tp_world)),
                    collect_kernel_timing=not args.no_kernel_timing, random_seed=args.seed,
                    dist_init_addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), watchdog_timeout=600.0,
                    # (RCCL refuses two ranks on one device: the one-GPU functional check runs the TP groups over gloo +
                    # the peer-memory all-reduce, like tests/test_gpu_engine.py's TP = 2 case)
                    **({"dist_backend": "gloo"} if os.environ.get("SEMIPD_BENCH_ALL_ON_GPU0") == "1" else {}))
    if args.mode == "unified" and tp_world > 1:
        raise SystemExit("unified mode is single-GPU only in this round")
    if args.tp:
        engine = Engine(sa, local_tp_ranks=[rank], gpu_ids={rank: local_rank})
    else:
        engine = Engine(sa, gpu_ids={0: local_rank})
    driver = rank == 0 or not args.tp      # who sends requests: rank 0 of a TP engine, every replica otherwise
    prompts = make_requests(args.num_requests, args.input_len, cfg.vocab_size, args.seed + 1000 * replica)
    arrivals = arrival_times(args.num_requests, args.request_rate, args.seed + 1000 * replica)

    try:
        for _ in range(args.warmup):
            barrier()
            if driver:
                run_wave(engine, prompts, arrivals, args.output_len)
        if driver and not args.no_kernel_timing:
            engine.get_stats(reset=True)
        barrier()
        t0 = time.time()
        all_records, wave_summaries = [], []
        for _ in range(args.steps):
            if driver:
                recs, dur = run_wave(engine, prompts, arrivals, args.output_len)
                all_records.extend(recs)
                wave_summaries.append(summarize(recs, dur))
        barrier()
        elapsed = time.time() - t0
        all_records, elapsed = combine_ranks(all_records, elapsed, rank, world, replicas=not args.tp)
        stats = engine.get_stats() if (rank == 0 and not args.no_kernel_timing) else []
        sweep = []
        n_sweep = args.sweep_num_requests or args.num_requests
        sweep_prompts = prompts if n_sweep == args.num_requests else make_requests(n_sweep, args.input_len, cfg.vocab_size,
                                                                                  args.seed + 1000 * replica)
        rate_scale = world if (args.tp and not args.fixed_load) else 1

        def run_sweep(eng, rates, skip_rate=None):
            """One wave per rate on `eng` (sweep_requests(rate) requests of the sweep's lengths); skip_rate: a rate that was
            measured already (the timed steps of the headline) and is filled in by the caller."""
            rows = []
            for rate in rates:
                if skip_rate is not None and rate == skip_rate:
                    continue
                n = sweep_requests(rate, n_sweep) if not args.sweep_num_requests else n_sweep
                recs, dur = run_wave(eng, sweep_prompts[:n], arrival_times(n, rate * rate_scale, args.seed), args.sweep_output_len)
                rows.append(sweep_point(summarize(recs, dur), rate, n, args.input_len, args.sweep_output_len))
            return rows

        # the headline's rate is a point of the grid that the timed steps have measured already (same lengths, all requests)
        headline_in_grid = (args.steps > 0 and args.sweep_output_len == args.output_len and not args.sweep_num_requests
                            and args.request_rate in sweep_rates)
        barrier()
        if driver and sweep_rates:
            sweep = run_sweep(engine, sweep_rates, skip_rate=args.request_rate if headline_in_grid else None)
            if headline_in_grid:
                sweep.append(sweep_point(summarize(all_records, sum(w["duration_s"] for w in wave_summaries)), args.request_rate,
                                         args.num_requests, args.input_len, args.output_len,
                                         source=f"the {args.steps} timed step(s) of the headline"))
                sweep.sort(key=lambda r: r["request_rate"])
        saturation = None
        if not args.no_saturation_wave and args.request_rate > 0:
            # capacity next to the load-bound headline: the same requests, all sent at once
            barrier()
            if driver:
                recs, dur = run_wave(engine, prompts, arrival_times(args.num_requests, 0.0, args.seed), args.output_len)
                sm = summarize(recs, dur)
                saturation = _rounded(sm)
        # token check, first half: the engine that was just timed produces TOKEN_CHECK_STEPS tokens for sixteen requests; the
        # unified engine below is the reference they are compared with
        probes = {}
        want_token_check = (world == 1 and args.mode == "semi-pd" and not args.no_unified_wave and not args.no_token_check)
        if want_token_check and driver:
            probes["semi-pd"] = token_probe(engine, cfg.vocab_size, args.seed)[0]
        barrier()
    finally:
        engine.shutdown()

    def side_waves(eng, n_timed):
        """One warm-up wave, then n_timed timed waves of the headline's requests on another engine; summary over them."""
        run_wave(eng, prompts, arrivals, args.output_len)
        recs_all, t0 = [], time.time()
        for _ in range(n_timed):
            recs, _ = run_wave(eng, prompts, arrivals, args.output_len)
            recs_all.extend(recs)
        sm = summarize(recs_all, time.time() - t0)
        return {"warmup_waves": 1, "timed_waves": n_timed, **_rounded(sm)}

    # side engines run the headline's load for at most this many timed waves (the driver's 20-step line would otherwise
    # spend as long on each of them as on the headline)
    n_side = max(1, min(args.steps, 2))
    static_split = None
    if (world == 1 and args.mode == "semi-pd" and not args.no_static_split_wave
            and (args.prefill_cu, args.decode_cu) != (50, 50)):
        # BASELINE config 2 as written: disjoint halves of the CUs, same requests and rate as the headline
        import dataclasses
        eng2 = Engine(dataclasses.replace(sa, prefill_cu_percent=50, decode_cu_percent=50, cu_mask_mode="env",
                                          decode_step_deadline_ms=0.0, decode_tbt_slo_ms=0.0, collect_kernel_timing=False),
                      gpu_ids={0: local_rank})
        try:
            static_split = {"workload": "same requests and rate, HSA_CU_MASK halves: prefill CUs 0-127, decode CUs 128-255",
                            **side_waves(eng2, n_side)}
            if want_token_check:
                probes["semi-pd 50/50"] = token_probe(eng2, cfg.vocab_size, args.seed)[0]
        except Exception as e:  # a side wave must never take the measured line down with it
            static_split = {"error": repr(e)}
        finally:
            eng2.shutdown()

    # The reference's claim is Semi-PD against the unified engine at equal load (README.md:105, evaluation/show_result.py:
    # 50-66): the same requests and arrival times through ONE process that interleaves prefill batches and decode steps
    # on every CU.
    unified, sweep_unified, token_check = None, [], None
    if world == 1 and args.mode == "semi-pd" and not args.no_unified_wave:
        import dataclasses
        eng5 = Engine(dataclasses.replace(sa, enable_semi_pd=False, collect_kernel_timing=False, mem_fraction_static=None,
                                          decode_step_deadline_ms=0.0, decode_tbt_slo_ms=0.0),
                      gpu_ids={0: local_rank})
        try:
            unified = {"workload": "same requests and rate, the unified engine (one process, chunked prefill and decode "
                                   "interleaved on every CU; --mode unified)", **side_waves(eng5, n_side)}
            if sweep_rates:
                # the same grid of rates as the Semi-PD engine above; the headline's rate = the waves just timed
                in_grid = headline_in_grid
                sweep_unified = run_sweep(eng5, sweep_rates, skip_rate=args.request_rate if in_grid else None)
                if in_grid:
                    pt = {k: unified[k] for k in ("output_tokens", "duration_s", "output_tok_s", "p50_ttft_ms", "p99_ttft_ms",
                                                  "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms")}
                    sweep_unified.append(sweep_point(pt, args.request_rate, args.num_requests, args.input_len, args.output_len,
                                                     source=f"the {n_side} timed wave(s) of unified_same_load"))
                    sweep_unified.sort(key=lambda r: r["request_rate"])
            if not args.no_saturation_wave and args.request_rate > 0:
                recs, dur = run_wave(eng5, prompts, arrival_times(args.num_requests, 0.0, args.seed), args.output_len)
                unified["saturation"] = _rounded(summarize(recs, dur))
            if want_token_check:
                ref, ref_lps = token_probe(eng5, cfg.vocab_size, args.seed, logprobs=True)
                checks = [compare_tokens(name, got, ref, ref_lps) for name, got in probes.items()]
                token_check = {"reference": f"the unified engine's tokens and top-{TOKEN_CHECK_TOP} log-probabilities",
                               "prompt_lens": list(TOKEN_CHECK_LENS), "margin": TOKEN_CHECK_MARGIN, "engines": checks,
                               "ok": all(c["ok"] for c in checks)}
        except Exception as e:
            unified = {"error": repr(e)}
            if want_token_check:
                token_check = {"ok": False, "error": f"the unified engine failed: {e!r}"}
        finally:
            eng5.shutdown()

    # BASELINE configs 1 and 3, one wave each, in the same invocation (they are parity-test cases, not the bench line:
    # extra keys only).  Config 1 runs the reference's own CPU-runnable case whole, with the CPU oracle timed on the same
    # workload beside it; config 3 is the MLA + MoE model at the config-2 load.
    side = {}
    default_line = (world == 1 and args.mode == "semi-pd" and args.model == "llama3-8b" and not args.no_side_configs)
    if default_line:
        import dataclasses
        for key, model, nreq, ilen, olen, rate, pcu, dcu in (
                ("config1_opt_125m", "opt-125m", 32, 128, 64, 0.0, args.prefill_cu, args.decode_cu),
                ("config3_deepseek_v2_lite", "deepseek-v2-lite", 256, 1024, 128, 32.0, args.prefill_cu, args.decode_cu)):
            try:
                scfg = model_config(model)
                ssa = dataclasses.replace(sa, model_config=scfg, context_length=ilen + olen + 8, prefill_cu_percent=pcu,
                                          decode_cu_percent=dcu, collect_kernel_timing=False, served_model_name=None)
                eng3 = Engine(ssa, gpu_ids={0: local_rank})
                try:
                    sp = make_requests(nreq, ilen, scfg.vocab_size, args.seed)
                    sarr = arrival_times(nreq, rate, args.seed)
                    run_wave(eng3, sp, sarr, olen)
                    recs, dur = run_wave(eng3, sp, sarr, olen)
                finally:
                    eng3.shutdown()
                sm = summarize(recs, dur)
                side[key] = {"workload": f"{model} bf16 TP=1 semi-pd, same CU policy as the headline, {nreq} synthetic requests "
                                         f"in={ilen} out={olen}, " + (f"Poisson {rate} req/s" if rate else "all sent at t = 0")
                                         + ", one warm-up wave + one timed wave",
                             **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in sm.items()}}
                if model == "opt-125m" and not args.no_cpu_baseline:
                    side[key]["cpu_baseline"] = cpu_baseline_opt(scfg, nreq, ilen, olen, args.seed)
            except Exception as e:  # a side wave must never take the measured line down with it
                side[key] = {"error": repr(e)}

    if rank != 0:
        return
    summ = summarize(all_records, elapsed)
    roofline = None
    extra = {}
    for s in stats:
        if s.get("role") == "DECODE" and s.get("decode_steps"):
            n = s["decode_steps"]
            extra["decode_step_ms"] = {"steps": int(n), "avg_batch": round(s["decode_tokens"] / n, 1),
                                       "schedule": round(1e3 * s.get("t_schedule_s", 0) / n, 3),
                                       "forward_and_sync": round(1e3 * s.get("t_forward_s", 0) / n, 3),
                                       "output": round(1e3 * s.get("t_output_s", 0) / n, 3),
                                       **{k: int(v) for k, v in s.items() if k.startswith("steps_on_")}}
        if s.get("role") == "PREFILL" and s.get("prefill_batches"):
            n = s["prefill_batches"]
            extra["prefill_batch_ms"] = {"batches": int(n), "avg_tokens": round(s["prefill_tokens"] / n, 1),
                                         "avg_requests": round(s.get("prefill_reqs", 0) / n, 2),
                                         "wait_admission": round(1e3 * s.get("t_wait_admission_s", 0) / n, 3),
                                         # from the launch to the ids on the host: for a batch launched behind a running one
                                         # this includes the rest of THAT batch (up to SEMIPD_PREFILL_LEAD_MS)
                                         "forward_and_sync": round(1e3 * s.get("t_forward_s", 0) / n, 3),
                                         **prefill_accounting(s, n, cfg.num_hidden_layers),
                                         # batches queued behind a running one by the late-binding loop, and how many
                                         # running batches had their first tokens sent from a layer hook of that launch
                                         "launched_behind_a_running_batch": int(s.get("late_bound_launches", 0)),
                                         "results_sent_from_layer_hook": int(s.get("results_sent_from_layer_hook", 0)),
                                         **{k: int(v) for k, v in s.items() if k.startswith("batches_on_")},
                                         # the decode-step deadline gate: launches, how many held the stream, for how long
                                         **({"step_gate": s["step_gate"]} if s.get("step_gate") else {})}
        kt = s.get("kernel_timing") or {}
        def hbm_line(k, kernel):
            return {"bound": "hbm", "kernel": kernel, "achieved": round(k["gbps"], 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(k["gbps"] / HBM_PEAK_GBPS, 4),
                    # PMC counters cannot be read from inside this process: the HBM bytes per launch come from the
                    # committed rocprofv3 --pmc pass of this kernel (profiles/pmc_traffic.json: FETCH_SIZE x 2 on gfx950
                    # + WRITE_SIZE over the algorithmic bytes of the profiled launch), scaled to this launch's bytes
                    **pmc_traffic(kernel, k["bytes_per_launch"]),
                    "avg_launch_us": round(k["avg_us"], 2),
                    "avg_launch_us_minus_event_overhead": round(k.get("avg_us_minus_event_overhead", k["avg_us"]), 2),
                    "event_pair_overhead_us": kt.get("_event_pair_overhead_us"),
                    "algorithmic_bytes_per_launch": int(k["bytes_per_launch"]), "launches_sampled": k["launches"]}
        if "decode_attention" in kt:
            if "Deepseek" in cfg.architectures[0]:
                kname = "mla_decode_kernel + decode_stage2_kernel (one decode_attention call)"
            elif os.environ.get("SEMIPD_FUSED_DECODE_ATTN", "1") != "0" and cfg.architectures[0] == "LlamaForCausalLM":
                # csrc/decode_attention_fused.hip; batches under half a workgroup per CU keep the separate launches
                kname = ("decode_rope_attn_kernel (qkv planes -> RoPE + KV store + paged attention + merge of the kv splits, "
                         "ONE launch per layer; the separate launches below 16 requests)")
            else:
                kname = "decode_mfma_kernel + decode_stage2_kernel (one decode_attention call)"
            attn_line = hbm_line(kt["decode_attention"], kname)
            if "stream_linear" in kt:
                # the dominant kernel of the decode instance by GPU time is the weight-streaming GEMM
                # (profiles/r02_bench_n1_decode_proc_kernel_stats_*.csv): qkv / o / gate_up + SiLU*mul / down of the
                # sampled step's first layer, algorithmic bytes = weight + activations in + result out per call
                roofline = hbm_line(kt["stream_linear"], "stream_gemm_glds_kernel (+ splitk_planes_reduce where K is "
                                    "sliced): the four dense layers of a decoder layer, decode batch")
                extra["decode_attention"] = attn_line
            else:
                roofline = attn_line
        if "prefill_gemm" in kt:
            # the dominant kernels of the PREFILL instance by GPU time: the four dense layers of a sampled batch's first
            # decoder layer, whatever serves them (layers/basic.py: _timed_gemm), flops = 2 rows n k
            k = kt["prefill_gemm"]
            extra["prefill_gemm"] = {"bound": "mfma", "kernel": "the dense layers of a prefill batch (qkv / o / gate_up / down of the "
                                     "sampled batch's first layer): hipBLASLt solutions timed on the share, gemm8p / gemm4w where "
                                     "they beat them (SiLU * mul in the epilogue there; the separate silu_and_mul is not in this "
                                     "interval)", "achieved": round(k["tflops"], 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": round(k["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
                                     "avg_launch_us": round(k["avg_us"], 1),
                                     "flops_per_launch": int(k["flops_per_launch"]), "launches_sampled": k["launches"]}
        if "extend_attention" in kt:
            k = kt["extend_attention"]
            extra["extend_attention"] = {"bound": "mfma", "achieved": round(k["tflops"], 1),
                                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(k["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4),
                                         "avg_launch_us": round(k["avg_us"], 1), "launches_sampled": k["launches"]}
    cpu = None
    if not args.no_cpu_baseline and world == 1 and cfg.architectures[0] in ("LlamaForCausalLM", "OPTForCausalLM"):
        try:
            if cfg.architectures[0] == "OPTForCausalLM":
                cpu = cpu_baseline_opt(cfg, args.num_requests, args.input_len, args.output_len, args.seed)
            else:
                cpu = cpu_baseline(cfg, args.input_len, args.output_len)
        except Exception as e:  # the baseline must never take the measured number down with it
            cpu = {"error": repr(e)}
    if args.mode != "semi-pd":
        mask_text = "one process on every CU"
    elif args.cu_mask_mode == "dynamic":
        mask_text = (f"work-conserving CU shares P{args.prefill_cu}/D{args.decode_cu} (the reference's MPS percentages as CU masks: "
                     f"prefill on a CU-masked stream over the lowest {args.prefill_cu} % of the CUs, decode on "
                     + ("every CU" if args.decode_cu >= 100 else f"the highest {args.decode_cu} %")
                     + "; an instance takes every CU while the other has nothing in flight"
                     + (f", the prefill instance also from {args.prefill_backlog_full_tokens} waiting prompt tokens" if args.prefill_backlog_full_tokens else "")
                     + (f"; a decode step older than {args.decode_step_deadline_ms:g} ms holds the prefill instance at its next "
                        "layer boundary until the step is over" if args.decode_step_deadline_ms > 0 else "")
                     + "; BASELINE config 2's literal 50 / 50 split is the side field static_split_50_50)")
    elif (args.prefill_cu, args.decode_cu) == (100, 100) or args.cu_mask_mode != "env":
        mask_text = "CU shares P100/D100 (no mask: both instances on every CU)"
    else:
        lo, hi = args.prefill_cu, 100 - args.decode_cu
        kind = "disjoint" if lo <= hi else "overlapping"
        mask_text = (f"CU masks P{args.prefill_cu}/D{args.decode_cu} ({kind} HSA_CU_MASK shares: prefill the lowest "
                     f"{args.prefill_cu} % of the CUs, decode the highest {args.decode_cu} %)")
    out = {
        "metric": "output tokens/s (Semi-PD mode; with p50 TTFT / TBT)" if args.mode == "semi-pd" else "output tokens/s (unified engine)",
        "value": round(summ["output_tok_s"], 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 2),
        "higher_is_better": True, "scaling": ("strong" if (args.fixed_load and args.tp) else "weak"), "vs_baseline": None, "dtype": ("fp8_e4m3fn" if args.quantization else "bf16"), "data": "synthetic",
        "p50_ttft_ms": summ["p50_ttft_ms"], "p50_tbt_ms": summ["p50_tbt_ms"],
        "p99_ttft_ms": summ["p99_ttft_ms"], "p99_tbt_ms": summ["p99_tbt_ms"],
        "config": {"workload": f"{args.model} {'block-fp8 (e4m3fn 128x128) linears + experts' if args.quantization else 'bf16'} TP={tp_world} {args.mode}"
                               + (f" x {world} independent replicas (one per GPU, each with its own {args.num_requests} requests)"
                                  if (world > 1 and not args.tp) else "") + f", {mask_text}, {args.num_requests} synthetic requests in={args.input_len} "
                               f"out={args.output_len}, Poisson {args.request_rate} req/s, dummy weights"
                               + ("" if args.kv_cache_dtype == "auto" else f", KV cache {args.kv_cache_dtype}"),
                   "num_requests": args.num_requests, "input_len": args.input_len, "output_len": args.output_len,
                   "request_rate": args.request_rate,
                   "parallelism": (f"tp{world}" if (args.tp or world == 1) else f"dp{world} (replicas, tp1 each)"),
                   "mode": args.mode, "prefill_cu_percent": args.prefill_cu, "decode_cu_percent": args.decode_cu,
                   "prefill_gemm": ("library solutions timed on the prefill share at start-up (csrc/dense_gemm.hip)"
                                    + (", next to a replaying decode step" if (world == 1 or not args.tp) and
                                       os.environ.get("SEMIPD_TUNE_UNDER_DECODE_LOAD", "1") != "0" else "")
                                    if (not args.no_prefill_gemm_tuning and args.prefill_cu < 100 and args.mode == "semi-pd"
                                        and args.cu_mask_mode in ("env", "dynamic"))
                                    else "library heuristic"),
                   "decode_step_deadline_ms": args.decode_step_deadline_ms, "decode_tbt_slo_ms": args.decode_tbt_slo_ms,
                   "kv_cache_dtype": args.kv_cache_dtype,
                   **({"slo": {"p99_ttft_ms": SLO_TTFT_P99_MS, "itl": {"p99_tbt_ms": SLO_ITL_P99_MS},
                               "tpot": {"p99_tpot_ms": SLO_TPOT_P99_MS},
                               "goodput": "highest Poisson rate of qps_sweep's grid whose wave meets p99 TTFT and the named "
                                          "objective for the gaps between tokens (itl: every gap; tpot: a request's mean gap)"}}
                      if sweep else {})},
        "roofline": roofline, "roofline_extra": extra, "cpu_baseline": cpu,
    }
    if static_split:
        out["static_split_50_50"] = static_split
    if unified is not None:
        out["unified_same_load"] = unified
    if sweep:
        gp = {"semi_pd": {"itl": goodput_of(sweep, "meets_slo_itl"), "tpot": goodput_of(sweep, "meets_slo_tpot")}}
        if sweep_unified:
            gp["unified"] = {"itl": goodput_of(sweep_unified, "meets_slo_itl"), "tpot": goodput_of(sweep_unified, "meets_slo_tpot")}
            out["qps_sweep_unified"] = sweep_unified
        out["goodput_req_s"] = gp["semi_pd"]["itl"]      # Semi-PD, the tail-of-every-gap objective (config.slo)
        out["goodput"] = {"unit": "requests/s", "rates": sweep_rates, **gp}
    if token_check is not None:
        out["token_check"] = token_check
    out.update(side)
    if saturation:
        out["saturation"] = {"note": "extra wave, every request sent at t = 0: output tok/s here is the engine's capacity; "
                                     "`value` above is measured at the Poisson rate named in config (load-bound)",
                             **saturation}
    if tp_world > 1:
        L, H = cfg.num_hidden_layers, cfg.hidden_size
        out["tensor_parallel"] = {"tp_world": tp_world, "ranks": world, "backend": ("gloo (one-GPU functional check)" if sa.dist_backend == "gloo" else "rccl (nccl backend of torch.distributed)") + " + peer-memory all-reduce kernels for payloads up to 16 MB",
                                  "all_reduce_calls_per_forward": 2 * L + 1,
                                  "all_reduce_bytes_per_token": (2 * L + 1) * H * 2,
                                  "logits_all_gather_bytes_per_request": cfg.vocab_size * 4}
    if sweep:
        out["qps_sweep"] = sweep
    print(json.dumps(out), flush=True)
    if token_check is not None and not token_check.get("ok"):
        # a timed engine whose tokens are not the unified engine's (beyond near-ties) has measured something else: the run fails
        print("bench.py: token check FAILED: " + json.dumps(token_check), file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
