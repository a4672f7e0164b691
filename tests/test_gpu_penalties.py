"""Frequency / presence / min_new_tokens penalties through the serving path on the GPU (reference:
sampling/penaltylib/*.py, sampling_batch_info.py:188-191 apply_logits_bias).

The engine's tokens are teacher-forced through the fp32 CPU oracle model; the reference's dense penalizer state
machine (restated in test_penalties_cpu.DensePenalizers-style below) is applied to the oracle's logits, and every
engine token must be the penalised argmax or within the bf16 tie margin of it.  Semi-PD must agree with the unified
engine: the prefill instance samples the first token (min_new_tokens applies there), the decode instance the rest
(its overlapped loop has to wait for each step's ids while a penalised request runs)."""
import pytest
import torch

from oracle.model import OracleLlama
from test_gpu_engine import MARGIN, make_prompts, server_args, tiny_llama

pytestmark = pytest.mark.gpu

NEW = 10


def penalised_margin_check(oracle, prompts, outs, sps, eos=()):
    """Every engine token must be within MARGIN of the best PENALISED oracle logit of its step."""
    n = max(len(o) for o in outs)
    padded = [o + [0] * (n - len(o)) for o in outs]
    _, logits = oracle.generate(prompts, n, forced=padded)
    flips = 0
    for b, (toks, sp) in enumerate(zip(outs, sps)):
        counts = torch.zeros(logits.shape[-1])
        stops = set(sp.stop_token_ids or ()) | set(eos)
        for s, t in enumerate(toks):
            row = logits[b, s].clone() - sp.frequency_penalty * counts - sp.presence_penalty * (counts > 0).float()
            if s < sp.min_new_tokens:
                for x in stops:
                    row[x] = float("-inf")
                assert t not in stops, f"request {b} sampled stop token {t} at step {s} < min_new_tokens"
            best = float(row.max())
            assert float(row[t]) >= best - MARGIN, (
                f"request {b} step {s}: token {t} has penalised oracle logit {float(row[t]):.4f}, "
                f"argmax {int(row.argmax())} has {best:.4f}")
            flips += int(int(row.argmax()) != int(logits[b, s].argmax()))
            counts[t] += 1
    return flips


def test_penalties_unified_and_semi_pd():
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg = tiny_llama()
    prompts = make_prompts(cfg.vocab_size, [5, 37, 128, 1, 64, 90], seed=11)
    eng = Engine(server_args(cfg))
    try:
        sd = {k: v.float().cpu() for k, v in eng.model_runner.model.state_dict().items()}
        oracle = OracleLlama(cfg, sd)
        plain_sp = SamplingParams(max_new_tokens=NEW, ignore_eos=True)
        plain = eng.generate(prompts, plain_sp)
        # a NEGATIVE frequency penalty rewards repetition: with -2 per occurrence the first token is repeated
        # for ever (random-weight logits are spread by much less than 2), a strong signal that the penalty is live
        rep = eng.generate(prompts, SamplingParams(max_new_tokens=NEW, ignore_eos=True, frequency_penalty=-2.0))
        flips = penalised_margin_check(oracle, prompts, rep,
                                       [SamplingParams(max_new_tokens=NEW, frequency_penalty=-2.0)] * 6)
        assert rep != plain and flips > 0, "the penalty never changed an argmax: the check above was vacuous"
        assert sum(len(set(o)) for o in rep) < sum(len(set(o)) for o in plain)
        # a mixed batch: positive frequency + presence, none, and min_new_tokens with the request's own natural
        # first tokens as stop ids (without min_new_tokens it would stop after one token)
        sps = [SamplingParams(max_new_tokens=NEW, ignore_eos=True, frequency_penalty=1.5, presence_penalty=0.5),
               plain_sp,
               SamplingParams(max_new_tokens=NEW, min_new_tokens=4, stop_token_ids=plain[2][:3]),
               SamplingParams(max_new_tokens=NEW, ignore_eos=True, presence_penalty=2.0),
               SamplingParams(max_new_tokens=NEW, min_new_tokens=NEW, stop_token_ids=plain[4][:1]),
               SamplingParams(max_new_tokens=NEW, stop_token_ids=plain[5][:1])]
        uni = eng.generate(prompts, sps)
        assert uni[1] == plain[1], "a request without penalties changed because its batch mates have them"
        assert uni[5] == plain[5][:1], "stop_token_ids without min_new_tokens must stop at the first token"  # incl. it
        assert len(uni[2]) >= 4 and len(uni[4]) == NEW and uni[2][0] != plain[2][0] and uni[4][0] != plain[4][0]
        assert len(set(uni[0])) == NEW and len(set(uni[3])) == NEW, "penalised requests repeated a token"
        penalised_margin_check(oracle, prompts, uni, sps)
    finally:
        eng.shutdown()
    semi = Engine(server_args(cfg, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50))
    try:
        got = semi.generate(prompts, sps, timeout=120)
        # request 2 may stop on any of three ids once min_new_tokens is reached: a flipped near-tie can move that step
        assert [len(g) for i, g in enumerate(got) if i != 2] == [len(u) for i, u in enumerate(uni) if i != 2]
        assert len(got[2]) >= 4
        penalised_margin_check(oracle, prompts, got, sps)   # near-ties may flip between the engines, not more
        assert got[5] == uni[5] and len(set(got[0])) == NEW
    finally:
        semi.shutdown()
