"""world_size-2 tests of the N>1 path on CPU over gloo (the GPU boxes available to the build are
single-GPU): TP linear / embedding / logits sharding + all-reduce / all-gather against the unsharded
computation, scheduler-message broadcast, and the Semi-PD P<->D protocol with two TP ranks per
instance (rank 0 owns the sockets, rank 1 follows the broadcasts; semi_pd_prefill_scheduler.py:140-147,
semi_pd_decode_scheduler.py:363-364)."""
import multiprocessing as mp
import io
import os
import socket
import sys
import traceback

import pytest
import torch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn_name, q, paths):
    try:
        for p in paths:
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(1)
        from semi_pd_amd import distributed as D
        D.init_distributed_environment(world, rank, f"tcp://127.0.0.1:{port}", backend="gloo")
        out = globals()[fn_name](rank, world)
        # as bytes: a tensor on a multiprocessing queue travels as a file descriptor that the parent fetches from THIS
        # process's resource sharer -- gone if the worker has exited before the parent unpickles (seen as a
        # FileNotFoundError on the listener socket, one run in a few dozen)
        buf = io.BytesIO()
        torch.save(out, buf)
        q.put((rank, "ok", buf.getvalue()))
        D.destroy_distributed_environment()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def _spawn(fn_name, world=2, timeout=120):
    from conftest import PKG, ROOT
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, fn_name, q, [ROOT, PKG, os.path.join(ROOT, "tests")]))
             for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, status, out = q.get(timeout=timeout)
        assert status == "ok", f"rank {rank}:\n{out}"
        results[rank] = torch.load(io.BytesIO(out), weights_only=False)
    for p in procs:
        p.join(30)
    return results


# ------------------------------------------------------------------------------ worker bodies
def _tp_layers(rank, world):
    import torch.nn.functional as F
    from semi_pd_amd.layers.basic import (MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear,
                                          VocabParallelEmbedding)
    from semi_pd_amd.distributed import tensor_model_parallel_all_gather
    g = torch.Generator().manual_seed(0)
    H, Hq, Hkv, D, I, V = 64, 4, 2, 16, 96, 100
    x = torch.randn(7, H, generator=g)
    full = {"qkv": torch.randn((Hq + 2 * Hkv) * D, H, generator=g), "o": torch.randn(H, Hq * D, generator=g),
            "gu": torch.randn(2 * I, H, generator=g), "down": torch.randn(H, I, generator=g),
            "emb": torch.randn(V, H, generator=g), "ids": torch.randint(0, V, (7,), generator=g)}
    qkv = QKVParallelLinear(H, D, Hq, Hkv, params_dtype=torch.float32)
    o = RowParallelLinear(Hq * D, H, params_dtype=torch.float32)
    gu = MergedColumnParallelLinear(H, [I, I], params_dtype=torch.float32)
    down = RowParallelLinear(I, H, params_dtype=torch.float32)
    emb = VocabParallelEmbedding(V, H, params_dtype=torch.float32)
    for layer, key in ((qkv, "qkv"), (o, "o"), (gu, "gu"), (down, "down"), (emb, "emb")):
        layer.weight.data.copy_(layer.weight.tp_shard(full[key]))
    # attention-less block: per-head "attention" = identity on v, so q/k shards only need the right shape
    y = qkv(x)
    q, k, v = y.split([qkv.num_heads * D, qkv.num_kv_heads * D, qkv.num_kv_heads * D], dim=-1)
    rep = qkv.num_heads // qkv.num_kv_heads
    attn = v.view(7, qkv.num_kv_heads, 1, D).expand(7, qkv.num_kv_heads, rep, D).reshape(7, -1) + q
    h = o(attn)
    a = gu(h)
    h2 = down(F.silu(a[:, : a.shape[1] // 2]) * a[:, a.shape[1] // 2:])
    e = emb(full["ids"])
    logits = tensor_model_parallel_all_gather(torch.matmul(h2, emb.weight.T))[:, :V]
    # unsharded reference
    yq, yk, yv = (x @ full["qkv"].T).split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    attn_f = yv.view(7, Hkv, 1, D).expand(7, Hkv, Hq // Hkv, D).reshape(7, -1) + yq
    h_f = attn_f @ full["o"].T
    a_f = h_f @ full["gu"].T
    h2_f = (F.silu(a_f[:, :I]) * a_f[:, I:]) @ full["down"].T
    torch.testing.assert_close(h, h_f, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(h2, h2_f, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(e, full["emb"][full["ids"]], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(logits, h2_f @ full["emb"].T, rtol=1e-4, atol=1e-2)
    return True


def _row_parallel_overlap(rank, world):
    """RowParallelLinear on a prefill-sized batch: the all-reduce of token chunk i is issued without blocking
    (async collective) while the GEMM of chunk i + 1 runs; the result has the bits of the blocking form on the
    same chunks and matches the unsharded layer."""
    import torch.distributed as dist
    from semi_pd_amd import distributed as D
    from semi_pd_amd.layers.basic import RowParallelLinear
    g = torch.Generator().manual_seed(3)
    T, I, H = 2500, 96, 64
    full_w = torch.randn(H, I, generator=g)
    x_full = torch.randn(T, I, generator=g)
    layer = RowParallelLinear(I, H, params_dtype=torch.float32)
    layer.weight.data.copy_(layer.weight.tp_shard(full_w))
    n = I // world
    x = x_full[:, rank * n:(rank + 1) * n].contiguous()
    assert D.all_reduce_overlap_chunks(T) == 4 and D.all_reduce_overlap_chunks(600) == 1
    got = layer(x)
    # blocking form on the same chunks
    step = -(-(-(-T // 4)) // 16) * 16
    want = torch.empty(T, H)
    for a in range(0, T, step):
        torch.mm(x[a:a + step], layer.weight.t(), out=want[a:a + step])
        dist.all_reduce(want[a:a + step])
    assert torch.equal(got, want)
    torch.testing.assert_close(got, x_full @ full_w.T, rtol=1e-4, atol=1e-3)
    # and the switch: one blocking call, same values up to GEMM blocking
    D.set_all_reduce_overlap(False)
    assert D.all_reduce_overlap_chunks(T) == 1
    torch.testing.assert_close(layer(x), want, rtol=1e-5, atol=1e-5)
    D.set_all_reduce_overlap(True)
    return True


def _broadcast(rank, world):
    from semi_pd_amd.distributed import barrier_cpu, broadcast_pyobj, get_tp_cpu_group
    data = [{"rids": ["a", "b"], "x": list(range(1000))}] if rank == 0 else []
    got = broadcast_pyobj(data, rank, get_tp_cpu_group(), src=0)
    barrier_cpu()
    return got[0]["rids"] == ["a", "b"] and len(got[0]["x"]) == 1000


def _sched_proc(role, rank, port, r2t, kv, names, q, paths):
    """One scheduler process of the 4-process topology (P0, P1, D0, D1): two gloo worlds (one per
    role, like the separate NCCL worlds of scheduler.py:249-259), AF_UNIX PUSH/PULL sockets on rank 0,
    req_to_token / fake-KV tensors shared between P_r and D_r (standing in for hipIpcMemHandle)."""
    try:
        for p in paths:
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(1)
        import test_semi_pd_protocol_cpu as T
        from semi_pd_amd import distributed as D
        from semi_pd_amd.managers.semi_pd_decode_scheduler import SemiPDDecodeScheduler
        from semi_pd_amd.managers.semi_pd_prefill_scheduler import SemiPDPrefillScheduler
        from semi_pd_amd.managers.transport import PullSocket, PushSocket
        from semi_pd_amd.mem_cache.memory_pool import ReqToTokenPool, TokenToKVPoolAllocator
        from types import SimpleNamespace
        D.init_distributed_environment(2, rank, f"tcp://127.0.0.1:{port}", backend="gloo")
        pool = ReqToTokenPool(r2t.shape[0], r2t.shape[1], "cpu", bypass_create_buffers=True)
        pool.req_to_token = r2t
        runner = SimpleNamespace(device=torch.device("cpu"), req_to_token_pool=pool,
                                 token_to_kv_pool_allocator=TokenToKVPoolAllocator(4000, torch.bfloat16, "cpu", None),
                                 max_total_num_tokens=4000)
        sa = T.args(tp_size=2, watchdog_timeout=30.0)
        r0 = rank == 0
        if role == "D":
            s = SemiPDDecodeScheduler(sa, runner, rank, PullSocket(names["d_in"]) if r0 else None,
                                      PushSocket(names["tok"]) if r0 else None,
                                      PushSocket(names["bridge"]) if r0 else None,
                                      PushSocket(names["p_in"]) if r0 else None)
        else:
            s = SemiPDPrefillScheduler(sa, runner, rank, PullSocket(names["p_in"]) if r0 else None,
                                       PushSocket(names["d_in"]) if r0 else None,
                                       PullSocket(names["bridge"]) if r0 else None)
        s.tp_worker = T.FakeWorker(runner, kv)
        q.put((role, rank, "ready", None))
        s.event_loop_normal()
        q.put((role, rank, "done", runner.token_to_kv_pool_allocator.available_size()))
        D.destroy_distributed_environment()
    except Exception:
        q.put((role, rank, "error", traceback.format_exc()))


def _share_agreement(rank, world):
    """model_executor/cu_share.py: one stream choice per unit of work for all TP ranks of an instance -- rank 0's, broadcast on
    the gloo group -- whatever each rank's own share board says at that instant (round-4 verdict item 4c)."""
    from semi_pd_amd import distributed as D
    from semi_pd_amd.model_executor.cu_share import FULL, SHARE, CuShare
    share = CuShare.__new__(CuShare)             # the decision logic alone: no streams, no GPU
    share.cus = {SHARE: 192, FULL: 256}
    share.tp = (rank, D.get_tp_cpu_group())
    share.board = None
    share.role = None
    mine = [FULL, SHARE, SHARE, FULL][rank::2]   # rank 0 sees FULL then SHARE, rank 1 the opposite
    seen = []
    for m in mine:
        share.choose = lambda m=m: m
        seen.append(share.decide())
    # the caller's own preference (the prefill backlog rule) is rank 0's as well
    seen.append(share.decide(FULL if rank == 0 else SHARE))
    seen.append(share.decide(SHARE if rank == 0 else None))
    assert seen == [FULL, SHARE, FULL, SHARE], (rank, seen)
    # one CU count for both streams (decode at 100 %): nothing to agree on, nothing is sent
    share.cus = {SHARE: 256, FULL: 256}
    share.choose = lambda: SHARE
    assert share.decide() == SHARE and share.decide(FULL) == FULL
    return True


# ------------------------------------------------------------------------------ tests
def test_tp2_share_choice_is_rank0s_for_every_rank():
    assert all(_spawn("_share_agreement").values())


def test_tp2_layers_match_unsharded():
    assert all(_spawn("_tp_layers").values())


def test_tp2_row_parallel_all_reduce_overlapped_with_gemm():
    assert all(_spawn("_row_parallel_overlap").values())


def test_tp2_broadcast_pyobj():
    assert all(_spawn("_broadcast").values())


def test_tp2_semi_pd_protocol_four_processes(tmp_path):
    import time
    import test_semi_pd_protocol_cpu as T
    from conftest import PKG, ROOT
    from semi_pd_amd.managers.io_struct import SamplingParams, ShutdownReq, TokenizedGenerateReqInput
    from semi_pd_amd.managers.transport import NOTHING, PullSocket, PushSocket
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    names = {k: str(tmp_path / k) for k in ("d_in", "p_in", "bridge", "tok")}
    ports = {"P": _free_port(), "D": _free_port()}
    paths = [ROOT, PKG, os.path.join(ROOT, "tests")]
    shared = {r: (torch.zeros(33, 516, dtype=torch.int32).share_memory_(),
                  torch.zeros(4001, dtype=torch.int64).share_memory_()) for r in (0, 1)}
    tok = PullSocket(names["tok"])
    procs = [ctx.Process(target=_sched_proc, args=(role, r, ports[role], shared[r][0], shared[r][1], names, q, paths))
             for role in ("D", "P") for r in (0, 1)]
    for p in procs:
        p.start()
    try:
        for _ in range(4):
            role, rank, status, info = q.get(timeout=120)
            assert status == "ready", f"{role}{rank}:\n{info}"
        d_in, p_in = PushSocket(names["d_in"]), PushSocket(names["p_in"])
        prompts = T.prompts_of([5, 40, 150, 12, 33, 90], seed=9)
        for i, pr in enumerate(prompts):
            req = TokenizedGenerateReqInput(f"r{i}", None, list(pr), SamplingParams(max_new_tokens=7, ignore_eos=True))
            d_in.send_pyobj(req)  # D first, then P
            p_in.send_pyobj(req)
        got = {}
        deadline = time.time() + 60
        while time.time() < deadline and not (len(got) == len(prompts) and all(len(v) >= 7 for v in got.values())):
            o = tok.recv_pyobj_nowait()
            if o is NOTHING:
                time.sleep(0.002)
                continue
            for rid, toks in zip(o.rids, o.output_ids):
                got.setdefault(rid, []).extend(toks)
        assert [got.get(f"r{i}") for i in range(len(prompts))] == [T.expected(pr, 7) for pr in prompts]
        d_in.send_pyobj(ShutdownReq())
        p_in.send_pyobj(ShutdownReq())
        done = {}
        for _ in range(4):
            role, rank, status, info = q.get(timeout=60)
            assert status == "done", f"{role}{rank}:\n{info}"
            done[(role, rank)] = info
        # both TP ranks of the decode instance mirrored every allocation and freed everything
        assert done[("D", 0)] == 4000 and done[("D", 1)] == 4000
        assert done[("P", 0)] == 4000 and done[("P", 1)] == 4000  # the prefill instance never allocates
    finally:
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.terminate()
        tok.close()


def _just_rendezvous(rank, world):
    import torch.distributed as dist
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    return float(t)


def test_rendezvous_of_the_tp_groups_under_a_torchrun_environment(monkeypatch):
    """`python -m torch.distributed.run ... bench.py --gpus N` hands TORCHELASTIC_USE_AGENT_STORE=True down to every
    scheduler process; with it torch makes no rank the server of a tcp:// rendezvous and the TP groups of the prefill and
    decode instances (their own ports) never form.  init_distributed_environment must serve the address itself."""
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    res = _spawn("_just_rendezvous", timeout=60)
    assert res == {0: 3.0, 1: 3.0}


# ------------------------------------------------------------------------------ rank-divergent host state
def _divergent_host_state(rank, world):
    """Every rank gets host state the other does not have -- rank 1's hipGraph capture fails, rank 0's kernel-timing sample list
    is full and never drained -- and then both run the SAME 24 decode steps through ModelRunner.forward.  Returns the path
    ("graph" / "eager") each step took."""
    from types import SimpleNamespace
    from semi_pd_amd.model_executor import hip_graph_runner as HG
    from semi_pd_amd.model_executor import model_runner as MR
    from semi_pd_amd.model_executor.forward_batch_info import ForwardMode
    from semi_pd_amd.model_executor.kernel_timing import KernelTiming

    class Runner:
        def __init__(self, mr):
            if rank == 1:
                raise RuntimeError("this rank's collective refused stream capture")

        def can_run(self, fb):
            return True

        def replay(self, fb):
            return "graph"

    HG.HipGraphRunner = Runner
    HG.recover_after_failed_capture = lambda device: None
    mr = MR.ModelRunner.__new__(MR.ModelRunner)
    mr.disable_cuda_graph, mr.cu_share, mr.tp_size, mr.tp_rank, mr.device = False, None, world, rank, "cpu"
    mr.graph_runner = None
    mr.init_cuda_graphs()
    mr.forward_decode = lambda fb: "eager"
    mr.kernel_timing = KernelTiming(sample_every=4, max_pending=8)
    if rank == 0:
        mr.kernel_timing._pending = [None] * 8          # nobody collects this rank's samples
    fb = SimpleNamespace(forward_mode=ForwardMode.DECODE, batch_size=3)
    return {"graphs": mr.graph_runner is not None, "paths": [mr.forward(fb) for _ in range(24)]}


def _all_captures_succeed(rank, world):
    from types import SimpleNamespace
    from semi_pd_amd.model_executor import hip_graph_runner as HG
    from semi_pd_amd.model_executor import model_runner as MR
    from semi_pd_amd.model_executor.forward_batch_info import ForwardMode

    class Runner:
        def __init__(self, mr):
            pass

        def can_run(self, fb):
            return True

        def replay(self, fb):
            return "graph"

    HG.HipGraphRunner = Runner
    mr = MR.ModelRunner.__new__(MR.ModelRunner)
    mr.disable_cuda_graph, mr.cu_share, mr.tp_size, mr.tp_rank, mr.device = False, None, world, rank, "cpu"
    mr.graph_runner = None
    mr.init_cuda_graphs()
    mr.forward_decode = lambda fb: "eager"
    from semi_pd_amd.model_executor.kernel_timing import KernelTiming
    mr.kernel_timing = KernelTiming(sample_every=4, max_pending=8)
    if rank == 0:
        mr.kernel_timing._pending = [None] * 8          # a full sample list on ONE rank: it samples the same steps anyway
    fb = SimpleNamespace(forward_mode=ForwardMode.DECODE, batch_size=3)
    return {"graphs": mr.graph_runner is not None, "paths": [mr.forward(fb) for _ in range(24)]}


def test_ranks_with_divergent_host_state_take_the_same_path():
    """The class of bug commit 19e5420 fixed (an N = 2 run hung after ~8 k steps: one rank sampled a step eagerly while the
    other replayed its graph, and the logits all-gather of the two paths has different block counts): whatever only ONE rank
    knows -- a failed capture, a sample list nobody drains -- must not choose between launch sequences.  Reference: the
    ranks of a TP group share every scheduling input by broadcast (managers/scheduler.py:645-659); here the start-up state
    is agreed on the CPU group as well (distributed.all_ranks_agree)."""
    res = _spawn("_divergent_host_state")
    assert res[0]["paths"] == res[1]["paths"], (res[0]["paths"], res[1]["paths"])
    assert not res[0]["graphs"] and not res[1]["graphs"]             # one failed capture: nobody replays
    assert set(res[0]["paths"]) == {"eager"}
    ok = _spawn("_all_captures_succeed")
    want = ["eager" if (i + 1) % 4 == 0 else "graph" for i in range(24)]    # every 4th step is a sampled (eager) one
    assert ok[0]["graphs"] and ok[1]["graphs"] and ok[0]["paths"] == ok[1]["paths"] == want


# ------------------------------------------------------------------------------ confined all-reduce: no fall-through
def _confined_reduce(rank, world):
    """distributed._confined_all_reduce with a stand-in for the peer-memory communicator (it sums over gloo and, like the real
    one, takes only contiguous 16-byte multiples up to max_size): odd shapes go through the staging buffer, big ones in pieces,
    the backend's own all-reduce is never called."""
    import torch.distributed as dist
    from semi_pd_amd import distributed as D
    backend_all_reduce = dist.all_reduce

    class FakeAR:
        max_size, disabled = 256, False
        calls = []

        def should_custom_ar(self, t):
            size = t.numel() * t.element_size()
            return t.dtype in (torch.float32, torch.bfloat16, torch.float16) and size > 0 and size % 16 == 0 \
                and t.is_contiguous() and size <= self.max_size

        def too_big_only(self, t):
            size = t.numel() * t.element_size()
            return t.dtype in (torch.float32, torch.bfloat16, torch.float16) and size > self.max_size and size % 16 == 0 \
                and t.is_contiguous()

        def all_reduce(self, t, out=None):
            assert self.should_custom_ar(t), "the kernel was handed a tensor it does not take"
            self.calls.append(t.numel())
            backend_all_reduce(t, group=D._DEVICE_GROUP)
            return t

        def all_reduce_in_pieces(self, t):
            flat = t.view(-1)
            step = self.max_size // t.element_size()
            for a in range(0, flat.numel(), step):
                self.all_reduce(flat[a:a + step])
            return t

        def should_custom_ag(self, t):
            size = t.numel() * t.element_size()
            return t.is_contiguous() and size > 0 and size % 16 == 0 and size <= self.max_size

        def all_gather(self, t):
            assert self.should_custom_ag(t), "the kernel was handed a tensor it does not take"
            out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype)
            backend_all_gather(list(out.unbind(0)), t.contiguous(), group=D._DEVICE_GROUP)
            return out

    backend_all_gather, backend_gather_into = dist.all_gather, dist.all_gather_into_tensor
    D._CUSTOM_AR = FakeAR()
    D._CONFINED["on"] = True

    def refuse(*a, **k):
        raise AssertionError("fell through to the backend's all-reduce on a confined instance")
    dist.all_reduce = refuse
    dist.all_gather = dist.all_gather_into_tensor = refuse
    try:
        g = torch.Generator().manual_seed(rank)
        out = {}
        odd = torch.randn(7, 5, generator=g)                    # 140 bytes: not a multiple of 16
        strided = torch.randn(6, 16, generator=g)[:, ::2]       # not contiguous
        big_odd = torch.randn(33, 5, generator=g)               # 660 bytes: staged AND in pieces
        big = torch.randn(32, 8, generator=g)                   # 1024 bytes: in pieces
        for name, t in (("odd", odd), ("strided", strided), ("big_odd", big_odd), ("big", big)):
            want = t.clone()
            r = D.tensor_model_parallel_all_reduce(t)
            assert r is t
            out[name] = (want, t.clone())
        pending = D.tensor_model_parallel_all_reduce_async(odd) if torch.cuda.is_available() else None
        del pending
        try:
            D.tensor_model_parallel_all_reduce(torch.ones(3, dtype=torch.int64))
            out["int_refused"] = False
        except RuntimeError as e:
            out["int_refused"] = "CU-confined" in str(e)
        # the logits all-gather: a payload above the kernels' limit and one that is no multiple of 16 bytes
        for name, t in (("gather_big", torch.randn(3, 100, generator=g)), ("gather_odd", torch.randn(2, 3, generator=g))):
            out[name] = (t.clone(), D.tensor_model_parallel_all_gather(t))
        out["stats"] = dict(D.CONFINED_STATS)
        return out
    finally:
        dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor = backend_all_reduce, backend_all_gather, backend_gather_into
        D._CUSTOM_AR, D._CONFINED["on"] = None, False


def test_a_confined_instance_never_falls_through_to_the_backends_all_reduce():
    """Round-5 verdict, multi-GPU item: in dynamic mode the RCCL fallback (non-contiguous / odd sizes) ran on RCCL's own
    unmasked stream.  Now every payload of a confined instance goes through the peer-memory kernels -- staged when its shape
    is odd -- and a dtype they cannot sum is refused.  parallel_state.py:376-436 is the reference's dispatch (custom kernel
    first, the backend for the rest)."""
    res = _spawn("_confined_reduce")
    for name in ("odd", "strided", "big_odd", "big"):
        total = res[0][name][0] + res[1][name][0]
        for r in (0, 1):
            assert torch.equal(res[r][name][1], total), name
    assert res[0]["int_refused"] and res[1]["int_refused"]
    for name in ("gather_big", "gather_odd"):
        want = torch.cat([res[0][name][0], res[1][name][0]], dim=-1)
        for r in (0, 1):
            assert torch.equal(res[r][name][1], want), name
    assert res[0]["stats"] == {"in_pieces": 1, "staged": 3, "gathered_in_pieces": 2}

    from semi_pd_amd.server_args import ServerArgs
    with pytest.raises(ValueError, match="disable-custom-all-reduce"):
        ServerArgs(enable_semi_pd=True, tp_size=2, disable_custom_all_reduce=True)
    ServerArgs(enable_semi_pd=True, tp_size=2, disable_custom_all_reduce=True, cu_mask_mode="env")
