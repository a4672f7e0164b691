"""One rank of the expert-parallel all-to-all test (tests/test_gpu_ep_all_to_all.py spawns `world` of these on one GPU,
each on its own CUs).  Checks semipd_ep_dispatch / semipd_ep_combine against the oracle permutation (oracle/ops.py:
ep_dispatch / ep_combine) bit for bit: ragged token counts per rank (also zero), several top-k and hidden widths, bf16 and
f16, back-to-back rounds without host synchronisation (double buffering), interleaved with all-reduces on the same regions,
and a hipGraph replay.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-pd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.set_num_threads(4)
    from oracle.ops import all_reduce_sum, ep_combine, ep_dispatch
    from semi_pd_amd.custom_all_reduce import CustomAllreduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ar = CustomAllreduce(dist.group.WORLD, dev, max_size=16 << 20)
    assert not ar.disabled
    report = {"rank": rank, "cases": 0, "bad": []}

    def same(name, got, want):
        report["cases"] += 1
        g = got.cpu()
        if g.shape != want.shape or not torch.equal(g.contiguous().view(torch.uint8), want.contiguous().view(torch.uint8)):
            report["bad"].append(name)

    def make_case(seed, tokens_per_rank, top_k, hidden, experts, dtype):
        g = torch.Generator().manual_seed(seed)
        xs, ids, ws = [], [], []
        for T in tokens_per_rank:
            xs.append((torch.randn(T, hidden, generator=g) * 2).to(dtype))
            # top_k DISTINCT experts per token, like a router
            ids.append(torch.stack([torch.randperm(experts, generator=g)[:top_k] for _ in range(T)]).to(torch.int32)
                       if T else torch.zeros(0, top_k, dtype=torch.int32))
            ws.append(torch.rand(T, top_k, generator=g))
        return xs, ids, ws

    def expert_fn(x, e, w, dtype):
        """A stand-in expert: row * (1 + local expert id / 8) * weight, rounded once (the real ones are GEMMs)."""
        return (x.float() * (1.0 + e.float()[:, None] / 8.0) * w[:, None]).to(dtype)

    cases = [
        (1, [5, 0, 17, 3, 9, 1, 30, 2][:world], 2, 64, 4 * world, torch.bfloat16),
        (2, [16] * world, 8, 7168, 8 * world, torch.bfloat16),            # DeepSeek-V3 row width, top-8
        (3, [1] + [0] * (world - 1), 6, 2048, 8 * world, torch.float16),  # one token in the whole group
        (4, [64, 7, 128, 1, 0, 50, 3, 96][:world], 2, 512, world, torch.bfloat16),   # one expert per rank
        (5, [0] * world, 2, 128, 2 * world, torch.bfloat16),              # nobody has a token
    ]
    launched = []
    for seed, tpr, k, H, E, dtype in cases:
        xs, ids, ws = make_case(seed, tpr, k, H, E, dtype)
        epr = E // world
        max_recv = max(1, sum(tpr) * k)
        want_x, want_e, want_w, counts, pos = ep_dispatch(xs, ids, ws, epr)
        st = ar.ep_dispatch(xs[rank].to(dev), ids[rank].to(dev), ws[rank].to(dev), epr, max_recv)
        n = int(st["recv_count"].item())
        same(f"case {seed}: count", torch.tensor([n]), torch.tensor([want_x[rank].shape[0]]))
        same(f"case {seed}: rows", st["recv_x"][:n], want_x[rank])
        same(f"case {seed}: experts", st["recv_expert"][:n], want_e[rank])
        same(f"case {seed}: weights", st["recv_weight"][:n], want_w[rank])
        same(f"case {seed}: counts", st["counts_all"], torch.tensor(counts, dtype=torch.int32))
        same(f"case {seed}: positions", st["send_within"] + 0, (pos[rank] - torch.tensor(
            [[sum(counts[s][int(ids[rank][t, j]) // epr] for s in range(rank)) for j in range(k)]
             for t in range(tpr[rank])], dtype=torch.int32).reshape(tpr[rank], k)))
        # the experts of this rank, then the way back
        y = torch.zeros(max_recv, H, dtype=dtype, device=dev)
        y[:n] = expert_fn(st["recv_x"][:n], st["recv_expert"][:n], st["recv_weight"][:n], dtype)
        ys = [expert_fn(want_x[d], want_e[d], want_w[d], dtype) for d in range(world)]
        out = ar.ep_combine(y, st)
        same(f"case {seed}: combined", out, ep_combine(ys, ids, pos, epr, dtype)[rank])
        # an all-reduce on the same regions in between: one call sequence for all collectives
        v = [(torch.arange(4096) % 13 + r).to(torch.bfloat16) for r in range(world)]
        same(f"case {seed}: all-reduce after", ar.all_reduce(v[rank].to(dev)), all_reduce_sum(v, torch.bfloat16))
        launched.append(seed)

    # overflow: max_recv (the same on every rank) below what the routing sends -- the dispatch keeps the first max_recv rows
    # and reports the full count, the way back counts the dropped entries as zero rows (never a read behind the staged rows),
    # and the wrapper's check raises on the ranks that overflowed
    seed, tpr, k, H, E, dtype = 6, [24] * world, 2, 128, 2 * world, torch.bfloat16
    xs, ids, ws = make_case(seed, tpr, k, H, E, dtype)
    epr, max_recv = E // world, 8
    want_x, want_e, want_w, counts, pos = ep_dispatch(xs, ids, ws, epr)
    st = ar.ep_dispatch(xs[rank].to(dev), ids[rank].to(dev), ws[rank].to(dev), epr, max_recv, check_overflow=False)
    n = int(st["recv_count"].item())
    same("overflow: count", torch.tensor([n]), torch.tensor([want_x[rank].shape[0]]))
    kept = min(n, max_recv)
    same("overflow: kept rows", st["recv_x"][:kept], want_x[rank][:kept])
    y = torch.zeros(max_recv, H, dtype=dtype, device=dev)
    y[:kept] = expert_fn(st["recv_x"][:kept], st["recv_expert"][:kept], st["recv_weight"][:kept], dtype)
    ys = []
    for d in range(world):
        yd = expert_fn(want_x[d], want_e[d], want_w[d], dtype)
        yd[max_recv:] = 0
        ys.append(yd)
    same("overflow: combined", ar.ep_combine(y, st), ep_combine(ys, ids, pos, epr, dtype)[rank])
    raised = False
    try:
        st = ar.ep_dispatch(xs[rank].to(dev), ids[rank].to(dev), ws[rank].to(dev), epr, max_recv)
    except RuntimeError as e:
        raised = "rows were dropped" in str(e)
    same("overflow: the wrapper raises where rows were dropped", torch.tensor([raised]), torch.tensor([n > max_recv]))
    # (keep the call sequence of the ranks aligned: the dispatch above was launched on every rank before any raise; `st`
    #  is the first dispatch's state where the second raised -- the same routing)
    ar.ep_combine(torch.zeros(max_recv, H, dtype=dtype, device=dev), st)

    # back to back without host synchronisation: 12 dispatch + combine rounds, alternating shapes, results checked at the end
    pending = []
    for i in range(12):
        tpr = [(7 * i + 3 * r) % 23 for r in range(world)]
        k, H, E, dtype = (2, 256, 2 * world, torch.bfloat16) if i % 2 else (4, 1024, 4 * world, torch.float16)
        xs, ids, ws = make_case(100 + i, tpr, k, H, E, dtype)
        epr = E // world
        max_recv = max(1, sum(tpr) * k)
        st = ar.ep_dispatch(xs[rank].to(dev), ids[rank].to(dev), ws[rank].to(dev), epr, max_recv)
        y = expert_fn(st["recv_x"], st["recv_expert"], st["recv_weight"], dtype)   # rows past recv_count: never read back
        out = ar.ep_combine(y, st)
        pending.append((i, xs, ids, ws, epr, dtype, st, out))
    torch.cuda.synchronize()
    for i, xs, ids, ws, epr, dtype, st, out in pending:
        want_x, want_e, want_w, counts, pos = ep_dispatch(xs, ids, ws, epr)
        ys = [expert_fn(want_x[d], want_e[d], want_w[d], dtype) for d in range(world)]
        same(f"round {i}: combined", out, ep_combine(ys, ids, pos, epr, dtype)[rank])

    # hipGraph: one captured dispatch + expert + combine, replayed on fresh inputs in the same static buffers
    T, k, H, E, dtype = 16, 4, 512, 4 * world, torch.bfloat16
    epr, max_recv = E // world, T * world * k
    xb = torch.zeros(T, H, dtype=dtype, device=dev)
    ib = torch.zeros(T, k, dtype=torch.int32, device=dev)
    wb = torch.zeros(T, k, dtype=torch.float32, device=dev)

    def step():
        st = ar.ep_dispatch(xb, ib, wb, epr, max_recv)
        y = expert_fn(st["recv_x"], st["recv_expert"], st["recv_weight"], dtype)
        return ar.ep_combine(y, st)

    xs, ids, ws = make_case(900, [T] * world, k, H, E, dtype)
    xb.copy_(xs[rank]); ib.copy_(ids[rank]); wb.copy_(ws[rank])
    step()
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=cap):
        out_g = step()
    torch.cuda.synchronize()
    for rep in range(3):
        xs, ids, ws = make_case(901 + rep, [T] * world, k, H, E, dtype)
        xb.copy_(xs[rank]); ib.copy_(ids[rank]); wb.copy_(ws[rank])
        g.replay()
        torch.cuda.synchronize()
        want_x, want_e, want_w, counts, pos = ep_dispatch(xs, ids, ws, epr)
        ys = [expert_fn(want_x[d], want_e[d], want_w[d], dtype) for d in range(world)]
        same(f"graph replay {rep}", out_g, ep_combine(ys, ids, pos, epr, dtype)[rank])
    dist.barrier()
    print("AR_REPORT " + json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
