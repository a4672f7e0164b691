"""One rank of tests/test_gpu_comm_confined.py: two tensor-parallel ranks as processes on the one GPU (as in ar_worker.py),
each "on its CU share": compute and communication stream carry one CU mask (rank 0: logical CUs 0-95, rank 1: 128-223;
disjoint so that both ranks' spinning kernels are resident at once).  Every collective goes through semi_pd_amd.distributed
the way the layers call it -- blocking, overlapped (communication stream), above the peer-memory limit (pieces), logits
all-gather -- with the communicator's CU trace on; the placement probe on the same masked stream says which hardware CU
slots the mask allows.  Prints one JSON line: results right?  every traced slot inside the mask?"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-pd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def slots(t):
    return {int(i) for i in torch.nonzero(t.cpu()).flatten().tolist()}


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.set_num_threads(4)
    import ctypes as C
    from semi_pd_amd import _lib, distributed as D
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    D.init_distributed_environment(world, rank, f"tcp://127.0.0.1:{port}", backend="gloo", device=dev)
    ar = D.get_custom_all_reduce()
    assert ar is not None and not ar.disabled
    lib = _lib.load()
    lo = rank * 128
    words = (C.c_uint32 * 8)()
    for i in range(lo, lo + 96):
        words[i >> 5] |= 1 << (i & 31)

    def masked_stream():
        s = C.c_void_p(0)
        _lib.check(lib.semipd_stream_create_cu_mask(0, C.addressof(words), 8, C.addressof(s)), "stream_create_cu_mask")
        return torch.cuda.ExternalStream(int(s.value), device=dev)

    compute, comm = masked_stream(), masked_stream()
    report = {"rank": rank, "bad": [], "cases": 0}
    # which hardware slots does the mask allow?  (4096 workgroups x ~50 us on the masked stream; ranks one after the other)
    probe = torch.full((4096, 2), -1, dtype=torch.int32, device=dev)
    for r in range(world):
        if r == rank:
            with torch.cuda.stream(compute):
                _lib.check(lib.semipd_probe_cu_placement(probe.data_ptr(), 4096, 100000, compute.cuda_stream), "probe")
            torch.cuda.synchronize()
        dist.barrier()
    allowed = {int(x) * 256 + int(c) for x, c in probe.cpu().tolist()}
    report["allowed_slots"] = len(allowed)

    trace = torch.zeros(2048, dtype=torch.int32, device=dev)
    ar.set_cu_trace(trace)
    torch.cuda.set_stream(compute)
    D.set_comm_stream(0, comm, confined=True)

    def check(name, got, want):
        report["cases"] += 1
        if not torch.equal(got.cpu(), want):
            report["bad"].append(name)

    def vec(n, r, dtype):
        return ((torch.arange(n) * 7 + r * 13) % 31).to(dtype)

    def want_sum(n, dtype):
        return sum(((torch.arange(n) * 7 + r * 13) % 31) for r in range(world)).to(dtype)

    # blocking, peer-memory kernel (1 MB)
    n = 512 * 1024
    x = vec(n, rank, torch.bfloat16).to(dev)
    check("blocking 1 MB", D.tensor_model_parallel_all_reduce(x), want_sum(n, torch.bfloat16))
    # above the kernels' limit: 40 MB in pieces, still on the caller's stream
    n = 20 * 1024 * 1024
    assert n * 2 > ar.max_size
    x = vec(n, rank, torch.bfloat16).view(2560, 8192).to(dev)
    check("blocking 40 MB in pieces", D.tensor_model_parallel_all_reduce(x).view(-1), want_sum(n, torch.bfloat16))
    # overlapped with the "GEMM": communication stream
    n = 1024 * 4096
    x = vec(n, rank, torch.bfloat16).view(1024, 4096).to(dev)
    h = D.tensor_model_parallel_all_reduce_async(x)
    h.wait()
    check("overlapped 8 MB", x.view(-1), want_sum(n, torch.bfloat16))
    x = vec(20 * 1024 * 1024, rank, torch.bfloat16).view(2560, 8192).to(dev)
    h = D.tensor_model_parallel_all_reduce_async(x)
    h.wait()
    check("overlapped 40 MB in pieces", x.view(-1), want_sum(20 * 1024 * 1024, torch.bfloat16))
    # logits all-gather
    g = D.tensor_model_parallel_all_gather(vec(8 * 4096, rank, torch.float32).view(8, 4096).to(dev))
    want = torch.cat([vec(8 * 4096, r, torch.float32).view(8, 4096) for r in range(world)], dim=-1)
    check("all-gather", g, want)
    torch.cuda.synchronize()
    seen = slots(trace)
    report["traced_slots"] = len(seen)
    report["outside_the_mask"] = sorted(seen - allowed)[:8]
    report["overlap_stats"] = dict(D.OVERLAP_STATS)
    ar.set_cu_trace(None)
    dist.barrier()
    print("AR_REPORT " + json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
