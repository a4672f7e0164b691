"""Is a CU-masked STREAM in an unmasked process the same neighbour as a process under HSA_CU_MASK?

A "prefill-like" child loops a library GEMM (1024 x 4096 x 28672, bf16) confined to the lowest 81 % of the CUs either by
HSA_CU_MASK (process mask) or by a hipExtStreamCreateWithCUMask stream (the dynamic shares); the parent, unmasked, times a
"decode-like" chain of 64 dependent weight-streaming GEMMs (32 rows) and reports the distribution of the chain time.
Run on the box:  python tools/mask_equiv_probe.py
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]


def prefill_like(mode, stop, ready, pct):
    import torch
    from semi_pd_amd.semi_pd.utils import cu_masked_stream
    x = torch.randn(1024, 4096, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(28672, 4096, device="cuda", dtype=torch.bfloat16)
    w2 = torch.randn(4096, 14336, device="cuda", dtype=torch.bfloat16)
    st = cu_masked_stream(0, pct, False) if mode == "stream" else torch.cuda.current_stream()
    n = 0
    with torch.cuda.stream(st):
        for _ in range(3):
            y = x @ w.t()
        torch.cuda.synchronize()
        ready.set()
        t0 = time.time()
        while not stop.is_set():
            for _ in range(20):
                y = x @ w.t()
                z = y[:, :14336] @ w2.t()
            torch.cuda.synchronize()
            n += 20
        dt = time.time() - t0
    print(f"  prefill-like child ({mode}): {dt / max(n, 1) * 1e3:.3f} ms per GEMM pair", flush=True)


def main():
    import numpy as np
    import torch
    from semi_pd_amd import ops
    from semi_pd_amd.semi_pd.utils import cu_mask_env, get_device_sm_count
    ncu = get_device_sm_count(0)
    pct = int(os.environ.get("PCT", "81"))
    x = torch.randn(32, 4096, device="cuda", dtype=torch.bfloat16)
    ws = [torch.randn(6144, 4096, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(8)]

    def chain():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h = x
        for i in range(64):
            y = ops.stream_linear(h, ws[i % 8])
            h = y[:, :4096]
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    def measure(label, n=300):
        for _ in range(10):
            chain()
        t = np.array([chain() for _ in range(n)])
        print(f"{label}: chain of 64 streaming GEMMs p50 {np.percentile(t, 50):.3f} ms  p90 {np.percentile(t, 90):.3f}  "
              f"p99 {np.percentile(t, 99):.3f}  max {t.max():.3f}", flush=True)

    measure("alone")
    ctx = mp.get_context("spawn")
    for mode in ("env", "stream", "none"):
        stop, ready = ctx.Event(), ctx.Event()
        env = cu_mask_env(0, ncu, pct, False) if mode == "env" else {}
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        p = ctx.Process(target=prefill_like, args=(mode, stop, ready, pct))
        p.start()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ready.wait(120)
        measure(f"next to a GEMM loop confined to {pct} % by {mode:6s}")
        stop.set()
        p.join(60)


if __name__ == "__main__":
    main()
