"""One rank of the peer-memory all-reduce test (tests/test_gpu_all_reduce.py spawns `world` of these on
one GPU, each confined to its own CUs by HSA_CU_MASK so that all ranks are resident at the same time).

Checks against the oracle (oracle/ops.py: all_reduce_sum): bit-exact, for sizes on both sides of the
one-stage / two-stage switch, back-to-back calls without host synchronisation, in-place calls and
hipGraph replays.  Prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-pd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    max_size = int(sys.argv[4]) if len(sys.argv) > 4 else 8 << 20
    torch.set_num_threads(4)  # `world` processes share the host: the default (all hardware threads each) thrashes
    from oracle.ops import all_reduce_sum
    from semi_pd_amd.custom_all_reduce import CustomAllreduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ar = CustomAllreduce(dist.group.WORLD, dev, max_size=max_size)
    if os.environ.get("SEMIPD_AR_TEST_FAIL"):
        # one rank was broken on purpose: every rank must have fallen back, nobody may hang or raise
        assert ar.disabled and ar.custom_all_reduce(torch.zeros(64, device=dev)) is None
        dist.barrier()
        print("AR_REPORT " + json.dumps({"rank": rank, "disabled": True}), flush=True)
        return
    assert not ar.disabled
    report = {"rank": rank, "cases": 0, "bad": []}
    t_start = time.perf_counter()

    def log(what):
        print(f"[rank {rank}] {time.perf_counter() - t_start:7.2f}s {what}", flush=True)

    log("communicator ready")
    if os.environ.get("AR_PROBE"):
        # diagnostic: latency of single synchronised calls (tools/ar_concurrency_probe.py)
        x = torch.ones(4096, device=dev, dtype=torch.bfloat16)
        lat = []
        for _ in range(int(os.environ["AR_PROBE"])):
            if os.environ.get("AR_PROBE_COPY"):
                x = x.cpu().to(dev)  # a D2H and an H2D copy between the calls
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ar.all_reduce(x)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat_sorted = sorted(lat)
        log(f"probe latencies ms: median {lat_sorted[len(lat) // 2]:.2f} max {lat_sorted[-1]:.2f} "
            f"over 100 ms: {sum(v > 100 for v in lat)} of {len(lat)}")
        dist.barrier()
        print("AR_REPORT " + json.dumps({"rank": rank, "probe_ms": lat}), flush=True)
        os._exit(0)

    def inputs_of(numel, dtype, seed):
        g = torch.Generator().manual_seed(seed)
        return [(torch.randn(numel, generator=g) * 3).to(dtype) for _ in range(world)]

    def check(name, got, want):
        report["cases"] += 1
        if not torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8)):
            diff = (got.cpu().float() - want.float()).abs().max().item()
            report["bad"].append(f"{name}: max abs diff {diff}")

    # sizes in bytes: the 16-byte minimum, odd vector counts, around the 256-byte chunk granule, both
    # sides of the one-stage limit (512 KB for <= 4 ranks, 256 KB for 8), the capacity
    sizes = [16, 48, 240, 256, 272, 4096, 65536 + 16, 256 * 1024 - 16, 256 * 1024, 512 * 1024 - 32, 512 * 1024,
             1 << 20, (3 << 20) + 4096 + 16, max_size]
    seed = 0
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for nbytes in sizes:
            numel = nbytes // dtype.itemsize
            seed += 1
            xs = inputs_of(numel, dtype, seed)
            x = xs[rank].to(dev)
            assert ar.should_custom_ar(x)
            t0 = time.perf_counter()
            out = ar.custom_all_reduce(x)
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > 0.2:
                log(f"slow call: {dtype} {nbytes} B took {time.perf_counter() - t0:.2f} s")
            check(f"{dtype} {nbytes}B", out, all_reduce_sum(xs, dtype))
            assert torch.equal(x.cpu(), xs[rank])  # out of place: the input is untouched
    log("sizes x dtypes done")
    # does not qualify: 8 bytes, above capacity, wrong dtype
    assert ar.custom_all_reduce(torch.zeros(4, dtype=torch.bfloat16, device=dev)) is None
    assert ar.custom_all_reduce(torch.zeros(max_size // 2 + 8, dtype=torch.bfloat16, device=dev)) is None
    assert ar.custom_all_reduce(torch.zeros(64, dtype=torch.int32, device=dev)) is None

    # back to back, sizes mixed, no host synchronisation in between: double buffering under stress
    mix = [4096, 1 << 20, 272, 512 * 1024, 65536 + 16, 2 << 20, 16, 768 * 1024] * 6
    ins, outs = [], []
    for i, nbytes in enumerate(mix):
        xs = inputs_of(nbytes // 2, torch.bfloat16, 1000 + i)
        ins.append(xs)
        outs.append((xs[rank].to(dev), None))
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    res = [ar.all_reduce(x) for x, _ in outs]
    torch.cuda.synchronize()
    report["mixed_48_calls_ms"] = (time.perf_counter() - t0) * 1e3
    for i, r in enumerate(res):
        check(f"mixed[{i}] {mix[i]}B", r, all_reduce_sum(ins[i], torch.bfloat16))

    # all-gather through the same regions (vocab-parallel logits), interleaved with reductions
    for i, (numel, dtype) in enumerate([(4, torch.float32), (1000 * 8, torch.bfloat16), (33 * 16032, torch.float32),
                                        (max_size // 2, torch.bfloat16)]):
        xs = inputs_of(numel, dtype, 5000 + i)
        x = xs[rank].to(dev)
        assert ar.should_custom_ag(x)
        got = ar.all_gather(x)
        red = ar.all_reduce(x) if ar.should_custom_ar(x) else None
        check(f"all_gather {numel} x {dtype}", got, torch.stack(xs))
        if red is not None:
            check(f"all_reduce after all_gather {numel} x {dtype}", red, all_reduce_sum(xs, dtype))
    log("mixed back-to-back done")
    # in place + hipGraph: h <- allreduce(h) * 0.5 three times per replay, five replays
    xs = inputs_of(8192 * 4, torch.bfloat16, 7)
    h = xs[rank].to(dev).clone()
    static = h.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ar.all_reduce(static, out=static)  # warm-up outside the capture (every rank does it once)
        static.copy_(h)
        torch.cuda.synchronize()
        dist.barrier()
        with ar.capture():
            with torch.cuda.graph(g, stream=s):
                for _ in range(3):
                    ar.all_reduce(static, out=static)
                    static.mul_(0.5)
    torch.cuda.current_stream().wait_stream(s)
    want = [x.clone() for x in xs]
    for _ in range(5):
        g.replay()
        for _ in range(3):
            red = all_reduce_sum(want, torch.bfloat16)
            want = [(red.float() * 0.5).to(torch.bfloat16) for _ in range(world)]
    torch.cuda.synchronize()
    check("graph replay", static, want[0])

    # bounded waits (what the start-up self-test relies on): rank 0 reduces alone, its waits give up after
    # 200 ms and are counted; the others catch up afterwards and the next call is correct again
    import ctypes as C
    from semi_pd_amd import _lib
    lib = _lib.load()
    x = torch.ones(2048, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        lib.semipd_ar_set_timeout_ms(ar._comm, 200)
        t0 = time.perf_counter()
        ar.all_reduce(x)
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
        n = C.c_uint32()
        _lib.check(lib.semipd_ar_timed_out(ar._comm, C.addressof(n)), "ar_timed_out")
        lib.semipd_ar_set_timeout_ms(ar._comm, 0)
        report["cases"] += 1
        # one wait per peer and barrier: 1 barrier in the one-stage kernel, 2 in the two-stage kernel
        if n.value not in (world - 1, 2 * (world - 1)) or not 0.15 < took < 5.0:
            report["bad"].append(f"lonely call: {n.value} waits gave up in {took:.2f} s")
    dist.barrier()
    if rank != 0:
        ar.all_reduce(x)
        torch.cuda.synchronize()
    dist.barrier()
    check("after the lonely call", ar.all_reduce(x), torch.full((2048,), float(world), dtype=torch.bfloat16))
    log("graph replays and timeout done")
    # timing of the decode-sized payload (32 tokens x 8192 hidden bf16): events around 200 calls
    x = torch.randn(32 * 8192, device=dev).to(torch.bfloat16)
    o = torch.empty_like(x)
    for _ in range(20):
        ar.all_reduce(x, out=o)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ar.all_reduce(x, out=o)
    e1.record()
    torch.cuda.synchronize()
    report["us_per_call_512KB"] = e0.elapsed_time(e1) * 1e3 / 200
    dist.barrier()
    ar.close()
    dist.destroy_process_group()
    print("AR_REPORT " + json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
