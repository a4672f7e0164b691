"""GPU parity of the stochastic sampling kernels (csrc/sampling.hip) against the CPU oracle and the
reference's own sampling tests (sgl-kernel/tests/test_sampling.py: support masks over repeated trials,
renorm within rtol = atol = 1e-3) plus a distribution check against the reference sampler's golden
token counts (tests/golden/sampling.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from semi_pd_amd import ops as _ops
    return _ops


def _rand_probs(batch, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    pre = torch.rand(batch, vocab, generator=g)
    return pre / pre.sum(dim=-1, keepdim=True)


# sgl-kernel/tests/test_sampling.py:8-52 (batch 1/19/99/989 x vocab 111/500/32000/128256; trials reduced
# from 1000 so that the suite stays within minutes)
@pytest.mark.parametrize("batch,vocab", [(1, 111), (19, 500), (99, 32000), (19, 128256), (989, 111)])
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_top_k_top_p_joint_sampling_support(ops, device, batch, vocab, p):
    k = int(vocab * 0.5) if p == 0.1 else int(vocab * 0.1)
    probs = _rand_probs(batch, vocab, 42)
    mask = O.top_k_top_p_joint_mask(probs, k, p).to(device)
    dprobs = probs.to(device)
    top_p = torch.full((batch,), p, device=device)
    top_k = torch.full((batch,), k, dtype=torch.int32, device=device)
    rows = torch.arange(batch, device=device)
    gen = torch.Generator(device=device).manual_seed(1)
    for _ in range(60):
        u = torch.rand(32, batch, device=device, generator=gen)
        samples, success = ops.top_k_top_p_sampling_from_probs(dprobs, u, top_k, top_p, filter_apply_order="joint")
        assert bool(success.all())
        s = samples.long()
        assert bool((s >= 0).all()) and bool((s < vocab).all())
        assert bool((mask[rows, s] == 1).all())


# sgl-kernel/tests/test_sampling.py:57-81
@pytest.mark.parametrize("batch,vocab", [(1, 111), (19, 500), (99, 32000), (19, 128256), (989, 500)])
@pytest.mark.parametrize("p", [0.1, 0.5, 0.9])
def test_top_p_renorm_prob(ops, device, batch, vocab, p):
    probs = _rand_probs(batch, vocab, 7)
    got = ops.top_p_renorm_prob(probs.to(device), p).cpu()
    torch.testing.assert_close(got, O.top_p_renorm_prob(probs, p), rtol=1e-3, atol=1e-3)
    ps = torch.full((batch,), p)
    got = ops.top_p_renorm_prob(probs.to(device), ps.to(device)).cpu()
    torch.testing.assert_close(got, O.top_p_renorm_prob(probs, ps), rtol=1e-3, atol=1e-3)


# sgl-kernel/tests/test_sampling.py:84-109
@pytest.mark.parametrize("batch,vocab", [(1, 111), (19, 500), (99, 32000), (19, 128256), (989, 500)])
@pytest.mark.parametrize("k", [10, 100, 500])
def test_top_k_renorm_prob(ops, device, batch, vocab, k):
    if k > vocab:
        pytest.skip("k should be less than vocab_size")
    probs = _rand_probs(batch, vocab, 42)
    got = ops.top_k_renorm_prob(probs.to(device), k).cpu()
    want = O.top_k_renorm_prob(probs, k)
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-3)
    assert torch.equal(got > 0, want > 0)  # exactly the k largest (no ties in continuous random rows)


# sgl-kernel/tests/test_sampling.py:112-141
@pytest.mark.parametrize("batch,vocab", [(1, 111), (19, 500), (99, 32000), (19, 128256)])
@pytest.mark.parametrize("p", [0.05, 0.2, 0.7, 1.0])
def test_min_p_sampling_support(ops, device, batch, vocab, p):
    probs = _rand_probs(batch, vocab, 42)
    mask = O.min_p_mask(probs, p).to(device)
    dprobs = probs.to(device)
    min_p = torch.full((batch,), p, device=device)
    rows = torch.arange(batch, device=device)
    gen = torch.Generator(device=device).manual_seed(3)
    for _ in range(60):
        u = torch.rand(batch, device=device, generator=gen)
        s = ops.min_p_sampling_from_probs(dprobs, u, min_p).long()
        assert bool((mask[rows, s] == 1).all())


@pytest.mark.parametrize("batch,vocab", [(3, 111), (16, 32000), (5, 128256)])
def test_softmax_temperature(ops, device, batch, vocab):
    g = torch.Generator().manual_seed(vocab)
    logits = torch.randn(batch, vocab, generator=g) * 4
    temps = torch.rand(batch, 1, generator=g) * 1.5 + 0.3
    got = ops.softmax_temperature_(logits.to(device), temps.to(device)).cpu()
    torch.testing.assert_close(got, O.softmax_temperature(logits, temps), rtol=1e-5, atol=1e-8)


def _freq_check(counts: np.ndarray, dist: np.ndarray, draws: int):
    assert ((counts > 0) <= (dist > 0)).all(), "sampled a token outside the reference support"
    sigma = np.sqrt(draws * dist * (1 - dist))
    assert (np.abs(counts - draws * dist) <= 5 * sigma + 1).all()


def test_sampling_distribution_matches_reference(ops, device):
    """Draw many samples per golden row (each row replicated across the batch) and compare the
    frequencies with the oracle distribution, which tests/test_oracle_golden.py pins to the reference
    sampler's own counts."""
    g = load_golden("sampling")
    probs = torch.from_numpy(g["probs"])
    top_ks, top_ps, min_ps = (torch.from_numpy(g[k]) for k in ("top_ks", "top_ps", "min_ps"))
    B, V = probs.shape
    rep, launches = 2048, 8
    draws = rep * launches
    dist = O.top_k_top_p_min_p_filter(probs, top_ks, top_ps, torch.zeros(B), False).double().numpy()
    dist_mp = O.top_k_top_p_min_p_filter(probs, top_ks, top_ps, min_ps, True).double().numpy()
    gen = torch.Generator(device=device).manual_seed(9)
    for r in range(B):
        pr = probs[r:r + 1].repeat(rep, 1).contiguous().to(device)
        kk = top_ks[r:r + 1].repeat(rep).to(device)
        pp = top_ps[r:r + 1].repeat(rep).to(device)
        mp = min_ps[r:r + 1].repeat(rep).to(device)
        counts = torch.zeros(V, dtype=torch.int64, device=device)
        counts_mp = torch.zeros(V, dtype=torch.int64, device=device)
        for _ in range(launches):
            u = torch.rand(32, rep, device=device, generator=gen)
            s, ok = ops.top_k_top_p_sampling_from_probs(pr, u, kk, pp)
            assert bool(ok.all())
            counts += torch.bincount(s.long(), minlength=V)
            # the sampler's min-p path: top-k renorm -> top-p renorm -> min-p sampling (sampler.py:95-100)
            q = ops.top_p_renorm_prob(ops.top_k_renorm_prob(pr, kk), pp)
            s2 = ops.min_p_sampling_from_probs(q, u, mp)
            counts_mp += torch.bincount(s2.long(), minlength=V)
        _freq_check(counts.cpu().numpy(), dist[r], draws)
        if r != 2 and r != 5:  # rows with top_k AND top_p < 1: sequential renorm != joint filter by design
            _freq_check(counts_mp.cpu().numpy(), dist_mp[r], draws)


def test_sampler_layer_mixed_batch(ops, device):
    """Sampler.forward on a batch mixing greedy (top_k = 1) and stochastic rows: greedy rows return the
    argmax, stochastic rows stay inside their top-k set."""
    from semi_pd_amd.layers.basic import LogitsProcessorOutput, Sampler
    from semi_pd_amd.managers.io_struct import SamplingParams
    from semi_pd_amd.sampling_batch_info import SamplingBatchInfo

    class _R:
        def __init__(self, sp):
            self.sampling_params = sp

    torch.manual_seed(0)
    V = 32000
    logits = torch.randn(4, V) * 3
    reqs = [_R(SamplingParams(temperature=0.0)), _R(SamplingParams(temperature=0.8, top_k=5)),
            _R(SamplingParams(temperature=1.0, top_p=0.3)), _R(SamplingParams(temperature=1.0, top_k=3, min_p=0.1))]
    sampler = Sampler()
    top5 = torch.topk(logits[1], 5).indices.tolist()
    top3 = torch.topk(logits[3], 3).indices.tolist()
    for need_min_p in (False, True):
        rs = reqs if need_min_p else reqs[:3]
        info = SamplingBatchInfo.from_reqs(rs, V, device)
        assert not info.is_all_greedy and info.need_min_p_sampling == need_min_p
        for _ in range(20):
            out = LogitsProcessorOutput(logits[: len(rs)].clone().to(device))
            ids = sampler(out, info).tolist()
            assert ids[0] == int(torch.argmax(logits[0]))
            assert ids[1] in top5
            if need_min_p:
                assert ids[3] in top3
    info = SamplingBatchInfo.from_reqs(reqs[:1], V, device)
    assert info.is_all_greedy


@pytest.mark.parametrize("batch,vocab", [(1, 111), (7, 32000), (33, 128256)])
def test_token_logprobs(ops, device, batch, vocab):
    g = torch.Generator().manual_seed(batch)
    logits = torch.randn(batch, vocab, generator=g) * 5
    ids = torch.randint(0, vocab, (batch,), generator=g)
    lp, lse = ops.token_logprobs(logits.to(device), ids.to(device))
    want = torch.log_softmax(logits, -1)
    torch.testing.assert_close(lp.cpu(), want[torch.arange(batch), ids], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(logits, -1), rtol=1e-5, atol=1e-5)


def test_sampler_logprobs_greedy_and_stochastic(ops, device):
    """Sampler.forward with return_logprob (layers/sampler.py:74-75, 84-89, 139-155): greedy rows report
    log_softmax, stochastic batches log(top-p-normalised softmax(logits / T))."""
    from semi_pd_amd.layers.basic import LogitsProcessorOutput, Sampler
    from semi_pd_amd.managers.io_struct import SamplingParams
    from semi_pd_amd.sampling_batch_info import SamplingBatchInfo

    class _R:
        def __init__(self, sp):
            self.sampling_params = sp

    torch.manual_seed(1)
    V = 5000
    logits = torch.randn(3, V) * 3
    sampler = Sampler()
    out = LogitsProcessorOutput(logits.clone().to(device))
    ids = sampler(out, SamplingBatchInfo.from_reqs([_R(SamplingParams())] * 3, V, device), return_logprob=True,
                  top_logprobs_nums=[0, 2, 5])
    want = torch.log_softmax(logits, -1)
    assert ids.tolist() == logits.argmax(-1).tolist()
    torch.testing.assert_close(out.next_token_logprobs.cpu(), want.max(-1).values, rtol=1e-5, atol=1e-5)
    assert [len(v) for v in out.next_token_top_logprobs_val] == [0, 2, 5]
    tv, ti = torch.topk(want[2], 5)
    assert out.next_token_top_logprobs_idx[2] == ti.tolist()
    torch.testing.assert_close(torch.tensor(out.next_token_top_logprobs_val[2]), tv, rtol=1e-5, atol=1e-5)
    # stochastic batch
    reqs = [_R(SamplingParams(temperature=0.7, top_p=0.9)), _R(SamplingParams(temperature=1.3, top_k=50)),
            _R(SamplingParams(temperature=1.0))]
    info = SamplingBatchInfo.from_reqs(reqs, V, device)
    out = LogitsProcessorOutput(logits.clone().to(device))
    ids = sampler(out, info, return_logprob=True, top_logprobs_nums=[1, 1, 1]).cpu()
    probs = O.softmax_temperature(logits, torch.tensor([0.7, 1.3, 1.0]))
    want = torch.log(O.top_p_normalize_probs(probs, torch.tensor([0.9, 1.0, 1.0]))).clamp(min=torch.finfo(torch.float32).min)
    torch.testing.assert_close(out.next_token_logprobs.cpu(), want[torch.arange(3), ids.long()], rtol=1e-3, atol=1e-3)
