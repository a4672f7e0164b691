"""Frequency / presence / min_new_tokens penalties (reference: sampling/penaltylib/*.py).

The reference keeps dense [B, vocab] state per penalizer, updated by scatter after every step; the sparse per-step
rebuild in SamplingBatchInfo must give the same logits.  `dense_reference` below restates the reference's state
machine (cumulate after every sampled token, filter on finish) step by step."""
import pytest
import torch

from semi_pd_amd.managers.io_struct import SamplingParams
from semi_pd_amd.managers.schedule_batch import Req
from semi_pd_amd.managers.tokenizer_manager import sampling_params_from_dict
from semi_pd_amd.sampling_batch_info import SamplingBatchInfo

V = 97


class DensePenalizers:
    """frequency_penalty.py:36-57, presence_penalty.py:36-57, min_new_tokens.py:36-79 for a fixed set of requests."""

    def __init__(self, reqs):
        b = len(reqs)
        self.freq = torch.tensor([r.sampling_params.frequency_penalty for r in reqs]).view(b, 1)
        self.pres = torch.tensor([r.sampling_params.presence_penalty for r in reqs]).view(b, 1)
        self.cum_freq = torch.zeros(b, V)
        self.cum_pres = torch.zeros(b, V)
        self.min_new = torch.tensor([r.sampling_params.min_new_tokens for r in reqs], dtype=torch.int32).view(b, 1)
        self.stop_pen = torch.zeros(b, V)
        for i, r in enumerate(reqs):
            for t in set(r.sampling_params.stop_token_ids or ()) | set(r.eos_token_ids):
                self.stop_pen[i, t] = float("-inf")
        self.len_out = torch.zeros(b, 1, dtype=torch.int32)

    def cumulate(self, output_ids):
        idx = torch.tensor(output_ids).view(-1, 1)
        self.cum_freq.scatter_add_(1, idx, self.freq)
        self.cum_pres.scatter_(1, idx, self.pres)
        self.len_out += 1

    def apply(self, logits):
        logits.sub_(self.cum_freq)
        logits.sub_(self.cum_pres)
        mask = (self.len_out < self.min_new).expand_as(logits)
        logits[mask] += self.stop_pen[mask]


def test_sparse_entries_equal_the_dense_state_machine():
    g = torch.Generator().manual_seed(3)
    sps = [SamplingParams(max_new_tokens=32, frequency_penalty=0.7, presence_penalty=-0.3),
           SamplingParams(max_new_tokens=32),                                       # no penalties in the same batch
           SamplingParams(max_new_tokens=32, min_new_tokens=5, stop_token_ids=[11, 12]),
           SamplingParams(max_new_tokens=32, presence_penalty=1.5, min_new_tokens=2, ignore_eos=True)]
    reqs = [Req(f"r{i}", [1, 2, 3, 11], sp, eos_token_ids={7}) for i, sp in enumerate(sps)]
    dense = DensePenalizers(reqs)
    for step in range(10):
        logits = torch.randn(len(reqs), V, generator=g)
        want = logits.clone()
        dense.apply(want)
        info = SamplingBatchInfo.from_reqs(reqs, V, "cpu")
        assert info.is_all_greedy and info.has_penalties
        got = logits.clone()
        info.apply_penalties(got)
        # the reference sums f per occurrence in fp32 and subtracts twice; here f * count + p is one value: 1e-5
        assert torch.allclose(got, want, rtol=0, atol=1e-5), step
        assert torch.equal(torch.isinf(got), torch.isinf(want))
        if step < 5:
            assert got[2, 7] == got[2, 11] == got[2, 12] == float("-inf")   # EOS and stop ids are banned
        else:
            assert torch.isfinite(got[2]).all()
        assert torch.equal(got[1], logits[1])
        ids = torch.randint(0, 20, (len(reqs),), generator=g).tolist()     # small range: tokens repeat
        for r, t in zip(reqs, ids):
            r.output_ids.append(t)
        dense.cumulate(ids)


def test_no_penalties_means_no_tensors():
    reqs = [Req("a", [1], SamplingParams()), Req("b", [1], SamplingParams())]
    info = SamplingBatchInfo.from_reqs(reqs, V, "cpu")
    assert info.is_all_greedy and not info.has_penalties and info.pen_rows is None
    # a request that asks for a frequency penalty but has generated nothing yet contributes no entry
    reqs = [Req("a", [1], SamplingParams(frequency_penalty=1.0))]
    assert not SamplingBatchInfo.from_reqs(reqs, V, "cpu").has_penalties
    reqs[0].output_ids += [4, 4, 5]
    info = SamplingBatchInfo.from_reqs(reqs, V, "cpu")
    assert sorted(zip(info.pen_toks.tolist(), info.pen_vals.tolist())) == [(4, 2.0), (5, 1.0)]


def test_out_of_range_stop_ids_are_harmless():
    """The batch does not always know the vocabulary size on the host (vocab_size 0): ids beyond the logits' width
    are neutralised at application time instead of writing out of bounds."""
    r = Req("a", [1], SamplingParams(min_new_tokens=2, stop_token_ids=[5, V + 100, 10 ** 9]), eos_token_ids={V - 1})
    info = SamplingBatchInfo.from_reqs([r], 0, "cpu")
    logits = torch.zeros(1, V)
    info.apply_penalties(logits)
    assert logits[0, 5] == logits[0, V - 1] == float("-inf")
    assert torch.isfinite(logits).sum() == V - 2
    info = SamplingBatchInfo.from_reqs([Req("a", [1], SamplingParams(min_new_tokens=2, stop_token_ids=[V + 1]))], 0, "cpu")
    logits = torch.zeros(1, V)
    info.apply_penalties(logits)
    assert (logits == 0).all()


def test_validation_follows_the_reference():
    """sampling_params.py:100-128."""
    for kw in ({"frequency_penalty": 2.5}, {"presence_penalty": -2.1}, {"min_new_tokens": -1},
               {"min_new_tokens": 9, "max_new_tokens": 8}):
        with pytest.raises(ValueError):
            SamplingParams(**kw)
    sp = sampling_params_from_dict({"frequency_penalty": 0.5, "presence_penalty": 0.25, "min_new_tokens": 3})
    assert (sp.frequency_penalty, sp.presence_penalty, sp.min_new_tokens) == (0.5, 0.25, 3) and sp.needs_penalties
    assert not sampling_params_from_dict({}).needs_penalties
    with pytest.raises(ValueError):
        sampling_params_from_dict({"repetition_penalty": 1.3})


def test_against_the_reference_penalizers_golden():
    """tests/golden/penalties.npz was produced by the reference's own BatchedPenalizerOrchestrator (frequency,
    presence, min_new_tokens) driven through prefill sample -> decode steps -> merge of a new batch -> two filters
    (make_golden.py gen_penalties).  The per-step sparse rebuild must reproduce every penalised logit row: same
    -inf pattern, values within fp32 accumulation error (the reference adds f once per occurrence)."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "penalties.npz"))
    eos = {int(z["eos"])}
    reqs = []
    for (f, p, m), stops in zip(z["params"], z["stops"]):
        sp = SamplingParams(max_new_tokens=64, frequency_penalty=float(f), presence_penalty=float(p),
                            min_new_tokens=int(m), stop_token_ids=[int(s) for s in stops if s >= 0] or None)
        reqs.append(Req(f"g{len(reqs)}", [1, 2], sp, eos_token_ids=eos))
    changed = 0
    for k in range(int(z["n_steps"])):
        batch = [reqs[i] for i in z[f"rows_{k}"]]
        logits = torch.from_numpy(z[f"logits_{k}"]).clone()
        want = torch.from_numpy(z[f"want_{k}"])
        SamplingBatchInfo.from_reqs(batch, logits.shape[1], "cpu").apply_penalties(logits)
        assert torch.equal(torch.isinf(logits), torch.isinf(want)), k
        assert torch.allclose(logits, want, rtol=0, atol=1e-5), k
        changed += int((want != torch.from_numpy(z[f"logits_{k}"])).sum())
        for r, t in zip(batch, z[f"ids_{k}"]):
            r.output_ids.append(int(t))
    assert changed > 50    # the golden is not trivially the identity (93 penalised entries)


def test_retracted_request_keeps_its_history_on_the_prefill_instance():
    """After a retraction the decode instance re-sends prompt + generated tokens as the new prompt
    (semi_pd_decode_scheduler.py:117-139).  The message says how many of its last ids were generated, so the prefill
    instance's sample for the next token is penalised like the unified engine's (the reference loses that history)."""
    sp = dict(max_new_tokens=32, frequency_penalty=1.0, min_new_tokens=4, stop_token_ids=[9])
    whole = Req("u", [1, 2, 3], SamplingParams(**sp))
    whole.output_ids = [5, 6, 5]                       # the unified engine / the decode instance's view
    resent = Req("p", [1, 2, 3, 5, 6, 5], SamplingParams(**sp), is_retracted=True)
    resent.retracted_output_len = 3                    # the prefill instance's view
    a, b = (SamplingBatchInfo.from_reqs([r], V, "cpu") for r in (whole, resent))
    la, lb = torch.zeros(1, V), torch.zeros(1, V)
    a.apply_penalties(la), b.apply_penalties(lb)
    assert torch.equal(la, lb) and la[0, 5] == -2 and la[0, 6] == -1 and la[0, 9] == float("-inf") and la[0, 1] == 0
    resent.output_ids.append(7)                        # 4 generated tokens: stop ids are allowed again
    lc = torch.zeros(1, V)
    SamplingBatchInfo.from_reqs([resent], V, "cpu").apply_penalties(lc)
    assert lc[0, 9] == 0 and lc[0, 7] == -1
