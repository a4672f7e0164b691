"""The oracle of the expert-parallel all-to-all (oracle/ops.py: ep_dispatch / ep_combine).  The reference has
no such collective (ep_moe/layer.py:190 all-reduces), so the oracle IS the definition; here its own invariants: every routed
entry arrives exactly once at the rank that owns its expert, in (sender, token, j) order; the way back is the inverse
permutation; dispatch -> identity experts -> combine is the single-GPU moe_sum of the same rows."""
import torch

from oracle import ops as O


def _case(seed, tokens_per_rank, k, H, E):
    g = torch.Generator().manual_seed(seed)
    xs = [(torch.randn(T, H, generator=g)).to(torch.bfloat16) for T in tokens_per_rank]
    ids = [torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32) if T
           else torch.zeros(0, k, dtype=torch.int32) for T in tokens_per_rank]
    ws = [torch.rand(T, k, generator=g) for T in tokens_per_rank]
    return xs, ids, ws


def test_dispatch_is_a_permutation_in_sender_token_j_order():
    world, k, H, E = 4, 3, 32, 8
    xs, ids, ws = _case(0, [5, 0, 9, 2], k, H, E)
    epr = E // world
    rx, re, rw, counts, pos = O.ep_dispatch(xs, ids, ws, epr)
    assert sum(r.shape[0] for r in rx) == sum(x.shape[0] for x in xs) * k
    for d in range(world):
        assert rx[d].shape[0] == sum(counts[s][d] for s in range(world)) == re[d].numel() == rw[d].numel()
        assert bool(((re[d] >= 0) & (re[d] < epr)).all())
        at = 0
        for s in range(world):                      # sender-major, then (t, j) row-major
            for t in range(xs[s].shape[0]):
                for j in range(k):
                    e = int(ids[s][t, j])
                    if e // epr == d:
                        assert int(pos[s][t, j]) == at and torch.equal(rx[d][at], xs[s][t])
                        assert int(re[d][at]) == e - d * epr and float(rw[d][at]) == float(ws[s][t, j])
                        at += 1
        assert at == rx[d].shape[0]


def test_dispatch_experts_combine_equals_the_single_gpu_sum():
    world, k, H, E = 8, 4, 64, 16
    xs, ids, ws = _case(1, [3, 7, 0, 1, 12, 4, 4, 9], k, H, E)
    epr = E // world
    rx, re, rw, counts, pos = O.ep_dispatch(xs, ids, ws, epr)
    # "experts": row * weight * (1 + global expert id), rounded once -- any row-wise function does
    ys = [(rx[d].float() * rw[d][:, None] * (1 + re[d].float() + d * epr)[:, None]).to(torch.bfloat16) for d in range(world)]
    out = O.ep_combine(ys, ids, pos, epr, torch.bfloat16)
    for s in range(world):
        T = xs[s].shape[0]
        staged = torch.stack([(xs[s].float() * ws[s][:, j:j + 1] * (1 + ids[s][:, j:j + 1].float())).to(torch.bfloat16)
                              for j in range(k)], 1) if T else torch.zeros(0, k, H, dtype=torch.bfloat16)
        # moe_sum's arithmetic (csrc/elementwise.hip): fp32 accumulation over j in order, one rounding
        want = torch.zeros(T, H)
        for j in range(k):
            want += staged[:, j].float()
        assert torch.equal(out[s], want.to(torch.bfloat16))
