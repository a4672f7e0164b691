"""Kernel micro-benchmarks at the SURVEY §8d shapes (run on the GPU box).  Prints one line per case with
the achieved fraction of the HBM / MFMA roofline.  Usage: python tools/kbench.py [decode|extend|norm|moe|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]
import torch  # noqa: E402

from semi_pd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
HBM, MFMA = 8000.0, 2500.0


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3  # seconds


def bench_decode():
    print("# decode attention: B, ctx, Hq, Hkv, D, splits -> us, GB/s, frac of 8 TB/s")
    for (Hq, Hkv, D) in [(32, 8, 128), (8, 1, 128), (12, 12, 64)]:
        for B in (1, 32, 128, 256):
            for ctx in (1024, 8192):
                if B * ctx > 2_200_000:
                    continue
                N = B * ctx + 1
                kb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
                vb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
                q = torch.randn(B, Hq, D, device=dev, dtype=torch.bfloat16)
                o = torch.empty_like(q)
                indptr = (torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx)
                idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
                best = None
                for splits in (1, 2, 4, 8, 16, 32):
                    if ctx // splits < 64:
                        continue
                    lg = torch.empty(B, Hq, splits, D + 1, device=dev, dtype=torch.float32)
                    t = timeit(lambda: ops.decode_attention_fwd(q, kb, vb, o, indptr, idx, lg, splits, D ** -0.5))
                    if best is None or t < best[0]:
                        best = (t, splits)
                t, splits = best
                nbytes = B * ctx * Hkv * 2 * D * 2 + 2 * B * Hq * D * 2
                print(f"decode B={B:4d} ctx={ctx:5d} Hq={Hq} Hkv={Hkv} D={D} splits={splits:2d}: {t * 1e6:8.1f} us "
                      f"{nbytes / t / 1e9:7.0f} GB/s  {nbytes / t / 1e9 / HBM:.3f}")
                del kb, vb


def bench_decode_small():
    """Serving-run regime: small decode batches, ctx ~ 1.1 k, every split factor (pick with HSA_CU_MASK)."""
    print("# decode attention, small batches: B, ctx, splits -> us (stage1+stage2), GB/s   HSA_CU_MASK=%s"
          % os.environ.get("HSA_CU_MASK", "-"))
    Hq, Hkv, D = 32, 8, 128
    for B in (4, 8, 16, 32, 64, 256):
        ctx = 1100
        N = B * ctx + 1
        kb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
        vb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
        q = torch.randn(B, Hq, D, device=dev, dtype=torch.bfloat16)
        o = torch.empty_like(q)
        indptr = (torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx)
        idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
        nbytes = B * ctx * Hkv * 2 * D * 2 + 2 * B * Hq * D * 2
        kvd = os.environ.get("KBENCH_KV", "")
        if kvd:  # fp8 pool rows: half the bytes
            fd = torch.float8_e5m2 if kvd == "fp8_e5m2" else torch.float8_e4m3fn
            kb, vb = kb.to(fd), vb.to(fd)
            nbytes = B * ctx * Hkv * 2 * D * 1 + 2 * B * Hq * D * 2
        row = []
        for splits in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            lg = torch.empty(B, Hq, splits, D + 1, device=dev, dtype=torch.float32)
            t = timeit(lambda: ops.decode_attention_fwd(q, kb, vb, o, indptr, idx, lg, splits, D ** -0.5), iters=50)
            row.append(f"{splits}:{t * 1e6:.1f}us/{nbytes / t / 1e9:.0f}")
        print(f"B={B:3d} ctx={ctx} kv={kvd or 'bf16'}: " + "  ".join(row))


def bench_mla():
    kvd = {"": torch.bfloat16, "fp8_e5m2": torch.float8_e5m2, "fp8_e4m3": torch.float8_e4m3fn}[os.environ.get("KBENCH_KV", "")]
    print(f"# MLA decode attention (latent 576/512, rows stored as {kvd}): B, ctx, H, splits -> us, GB/s, frac of 8 TB/s")
    for H in (16, 128):
        for B in (1, 32, 128):
            for ctx in (1024, 8192):
                N = B * ctx + 1
                kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16).to(kvd)
                q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
                o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
                indptr = (torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx)
                idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
                best = None
                for splits in (1, 2, 4, 8, 16, 32, 64):
                    if ctx // splits < 64:
                        continue
                    lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
                    t = timeit(lambda: ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1))
                    if best is None or t < best[0]:
                        best = (t, splits)
                t, splits = best
                nbytes = B * ctx * 576 * kvd.itemsize + B * H * (576 + 512) * 2
                print(f"mla B={B:4d} ctx={ctx:5d} H={H:3d} splits={splits:2d}: {t * 1e6:8.1f} us "
                      f"{nbytes / t / 1e9:7.0f} GB/s  {nbytes / t / 1e9 / HBM:.3f}")


def bench_mla_small():
    """Serving-run regime for MLA decode: small batches, ctx ~ 1.1 k, H = 16, every split factor."""
    print("# MLA decode, small batches: B, ctx, splits -> us (stage1+stage2) / GB/s   HSA_CU_MASK=%s"
          % os.environ.get("HSA_CU_MASK", "-"))
    H, ctx = 16, 1100
    for B in (4, 8, 16, 32, 64, 128):
        N = B * ctx + 1
        kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
        q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
        o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
        indptr = (torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx)
        idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
        nbytes = B * ctx * 576 * 2 + B * H * (576 + 512) * 2
        row = []
        for splits in (1, 2, 3, 4, 6, 8, 12, 16):
            lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
            t = timeit(lambda: ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1), iters=50)
            row.append(f"{splits}:{t * 1e6:.1f}us/{nbytes / t / 1e9:.0f}")
        print(f"B={B:3d} ctx={ctx}: " + "  ".join(row))


def bench_extend():
    print("# extend attention: B, ext, prefix, Hq, Hkv, D -> us, TFLOP/s, frac of 2.5 PF")
    for (Hq, Hkv, D) in [(32, 8, 128), (12, 12, 64)]:
        for B, ext, pre in [(1, 1024, 0), (8, 1024, 0), (1, 8192, 0), (4, 2048, 0), (8, 128, 1024), (32, 128, 0), (2, 1024, 1024)]:
            T = B * ext
            N = B * pre + 8
            q = torch.randn(T, Hq, D, device=dev, dtype=torch.bfloat16)
            k = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
            v = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
            o = torch.empty_like(q)
            kb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
            vb = torch.randn(N, Hkv, D, device=dev, dtype=torch.bfloat16)
            qo = torch.arange(B + 1, device=dev, dtype=torch.int32) * ext
            kvp = torch.arange(B + 1, device=dev, dtype=torch.int32) * pre
            idx = (torch.randperm(N - 1, device=dev)[: max(B * pre, 1)] + 1).to(torch.int32)
            t = timeit(lambda: ops.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvp, idx, None, None, ext), iters=10)
            flops = 4.0 * Hq * D * B * ext * (pre + (ext + 1) / 2)
            print(f"extend B={B:3d} ext={ext:5d} pre={pre:5d} Hq={Hq} Hkv={Hkv} D={D}: {t * 1e6:9.1f} us "
                  f"{flops / t / 1e12:7.1f} TF/s  {flops / t / 1e12 / MFMA:.3f}")


def bench_norm():
    print("# fused_add_rmsnorm / rmsnorm / silu_and_mul / rope_kv_store: T, H -> us, GB/s")
    for H in (4096, 8192):
        for T in (32, 256, 8192):
            x = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
            r = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
            w = torch.ones(H, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: ops.fused_add_rmsnorm(x, r, w, 1e-5))
            print(f"fused_add_rmsnorm T={T:5d} H={H}: {t * 1e6:7.1f} us {4 * T * H * 2 / t / 1e9:7.0f} GB/s")
    for T in (32, 8192):
        x = torch.randn(T, 2 * 14336, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.silu_and_mul(x))
        print(f"silu_and_mul T={T:5d} d=14336: {t * 1e6:7.1f} us {3 * T * 14336 * 2 / t / 1e9:7.0f} GB/s")


def bench_moe():
    print("# fused MoE (align + GEMM1 + silu*mul + GEMM2 + sum): T, E, k, K, N -> us, TFLOP/s, weight GB/s")
    from semi_pd_amd.layers.moe import fused_experts
    E, k, K, N = 64, 6, 2048, 1408
    w1 = torch.randn(E, 2 * N, K, device=dev, dtype=torch.bfloat16) * 0.02
    w2 = torch.randn(E, K, N, device=dev, dtype=torch.bfloat16) * 0.02
    for T in [int(v) for v in os.environ.get("KBENCH_MOE_TS", "1,32,256,512,1024,2048,4096,8192").split(",")]:
        x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
        tw, ti = ops.topk_softmax(torch.randn(T, E, device=dev), k, True)
        t = timeit(lambda: fused_experts(x, w1, w2, tw, ti), iters=10)
        flops = 2.0 * T * k * 3 * N * K
        wbytes = min(E, T * k) * 3 * N * K * 2
        print(f"moe T={T:5d} E={E} k={k} K={K} N={N}: {t * 1e6:9.1f} us {flops / t / 1e12:7.1f} TF/s "
              f"{wbytes / t / 1e9:7.0f} GB/s(weights)")


def bench_lm_head():
    print("# lm_head_argmax: B, H, V -> us, GB/s of weight stream")
    for B in (1, 32, 256):
        H, V = 4096, 128256
        h = torch.randn(B, H, device=dev, dtype=torch.bfloat16)
        w = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02
        t = timeit(lambda: ops.lm_head_argmax(h, w), iters=5)
        t2 = timeit(lambda: torch.matmul(h, w.T).float().argmax(-1), iters=5)
        print(f"lm_head_argmax B={B:3d}: {t * 1e6:8.1f} us {V * H * 2 / t / 1e9:6.0f} GB/s   (torch matmul+argmax {t2 * 1e6:8.1f} us)")


def bench_linear():
    """Decode-sized dense layers of Llama-3-8B: hipBLASLt (F.linear) vs the weight-streaming split-K kernel
    (ops.linear; KBENCH_NUM_CUS = compute units assumed by its split heuristic)."""
    print("# decode linear M x [N, K]: hipBLASLt us / GB/s   ops.linear us / GB/s   HSA_CU_MASK=%s"
          % os.environ.get("HSA_CU_MASK", "-"))
    import torch.nn.functional as F
    for M in (1, 16, 32, 64):
        for (N, K) in ((28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)):
            # rotate over several weight copies so that the 4 MB L2 / 256 MB MALL do not serve re-reads
            copies = max(2, int(1.2e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            it = [0]

            def f1():
                it[0] += 1
                return F.linear(x, ws[it[0] % copies])

            ncu = int(os.environ.get("KBENCH_NUM_CUS", "0"))

            def f2():
                it[0] += 1
                return ops.linear(x, ws[it[0] % copies], num_cus=ncu)
            t1 = timeit(f1, iters=3 * copies)
            t2 = timeit(f2, iters=3 * copies)
            by = N * K * 2
            print(f"linear M={M:3d} N={N:6d} K={K:6d}: blaslt {t1 * 1e6:7.1f} us {by / t1 / 1e9:6.0f} GB/s   "
                  f"skinny {t2 * 1e6:7.1f} us {by / t2 / 1e9:6.0f} GB/s")
            del ws


def bench_stream_linear():
    """Decode-sized dense layers of Llama-3-8B: hipBLASLt (F.linear [+ silu_and_mul]) vs ops.stream_linear with its
    number of K slices swept through SEMIPD_SL_KS; KBENCH_NUM_CUS = CUs assumed."""
    import torch.nn.functional as F
    ncu = int(os.environ.get("KBENCH_NUM_CUS", "0"))
    print("# stream_linear M x [N, K]: hipBLASLt us | default us GB/s | best knob us GB/s   HSA_CU_MASK=%s num_cus=%d"
          % (os.environ.get("HSA_CU_MASK", "-"), ncu))
    full = os.environ.get("KBENCH_SL_SWEEP", "1") == "1"
    if ncu:
        from semi_pd_amd import _lib
        _lib.load().semipd_stream_linear_set_cus(ncu)   # the K split fills whole rounds of this share
    for M in [int(v) for v in os.environ.get("KBENCH_MS", "8,16,32,64").split(",")]:
        for (N, K, silu) in ((28672, 4096, True), (28672, 4096, False), (4096, 14336, False), (6144, 4096, False),
                             (4096, 4096, False)):
            copies = max(2, int(1.2e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            it = [0]

            def f1():
                it[0] += 1
                y = F.linear(x, ws[it[0] % copies])
                return ops.silu_and_mul(y) if silu else y

            def f2():
                it[0] += 1
                return ops.stream_linear(x, ws[it[0] % copies], fuse_silu_mul=silu)
            t1 = timeit(f1, iters=3 * copies)
            os.environ.pop("SEMIPD_SL_KS", None)
            t0 = timeit(f2, iters=3 * copies)
            res = {}
            if full:
                for ksp in (1, 2, 3, 4, 6, 8, 16):
                    os.environ["SEMIPD_SL_KS"] = str(ksp)
                    res[ksp] = timeit(f2, iters=2 * copies)
                os.environ.pop("SEMIPD_SL_KS", None)
            by = N * K * 2
            line = (f"M={M:3d} N={N:6d} K={K:6d} silu={int(silu)}: blaslt {t1 * 1e6:6.1f} | default {t0 * 1e6:6.1f} us "
                    f"{by / t0 / 1e9:5.0f} GB/s")
            if res:
                best = min(res, key=res.get)
                top = sorted(res.items(), key=lambda kv: kv[1])[:4]
                line += (f" | best KS={best} {res[best] * 1e6:6.1f} us {by / res[best] / 1e9:5.0f} GB/s | "
                         + " ".join(f"{k}:{v * 1e6:.0f}" for k, v in sorted(res.items())))
            print(line, flush=True)
            del ws


def bench_stream_planes():
    """The decode GEMM kernel ALONE (ops.stream_linear_planes: no reduction launch -- in a decode step the planes are summed by
    the consumer), Llama-3-8B and 70B-rank shapes, K slices swept through SEMIPD_SL_KS; KBENCH_NUM_CUS = CUs declared.
    Weights cycle through > 1 GB so that neither L2 nor the 256 MB MALL holds them."""
    ncu = int(os.environ.get("KBENCH_NUM_CUS", "0"))
    print("# stream_linear_planes M x [N, K]: default us GB/s | per KS us   HSA_CU_MASK=%s num_cus=%d SEMIPD_SL_RING=%s"
          % (os.environ.get("HSA_CU_MASK", "-"), ncu, os.environ.get("SEMIPD_SL_RING", "3")))
    if ncu:
        from semi_pd_amd import _lib
        _lib.load().semipd_stream_linear_set_cus(ncu)
    shapes = ((4096, 4096), (6144, 4096), (4096, 14336), (28672, 4096), (1280, 8192), (8192, 1024))
    for M in [int(v) for v in os.environ.get("KBENCH_MS", "32").split(",")]:
        for (N, K) in shapes:
            copies = max(2, int(1.2e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            it = [0]

            def f():
                it[0] += 1
                return ops.stream_linear_planes(x, ws[it[0] % copies])
            os.environ.pop("SEMIPD_SL_KS", None)
            t0 = timeit(f, iters=3 * copies)
            ks0 = f().ksplit
            res = {}
            for ksp in (1, 2, 4, 8, 16):
                os.environ["SEMIPD_SL_KS"] = str(ksp)
                res[ksp] = timeit(f, iters=2 * copies)
            os.environ.pop("SEMIPD_SL_KS", None)
            by = N * K * 2
            print(f"M={M:3d} N={N:6d} K={K:6d}: default KS={ks0} {t0 * 1e6:6.1f} us {by / t0 / 1e9:5.0f} GB/s | "
                  + " ".join(f"{k}:{v * 1e6:.1f}" for k, v in sorted(res.items())), flush=True)
            del ws


def graph_time(fn, launches, replays=5):
    """Seconds per launch of `fn` inside a hipGraph of `launches` calls (device time: no host launch cost between them)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=side):
        for _ in range(launches):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / (replays * launches)


def bench_stream_planes_graph():
    """bench_stream_planes with the launches inside a hipGraph: what a decode step pays per GEMM kernel (the eager loop of
    bench_stream_planes is bound by ~11 us of host work per call for the small shapes).  `+norm`: GEMM -> fused add + norm on
    the planes, the pair a decode layer runs for o_proj / down_proj."""
    ncu = int(os.environ.get("KBENCH_NUM_CUS", "0"))
    print("# stream_linear_planes in a hipGraph, M x [N, K]: KS -> us per launch (GB/s of weights) | with the consumer   HSA_CU_MASK=%s "
          "num_cus=%d SEMIPD_SL_RING=%s" % (os.environ.get("HSA_CU_MASK", "-"), ncu, os.environ.get("SEMIPD_SL_RING", "3")))
    if ncu:
        from semi_pd_amd import _lib
        _lib.load().semipd_stream_linear_set_cus(ncu)
    shapes = ((4096, 4096), (6144, 4096), (4096, 14336), (28672, 4096), (1280, 8192), (8192, 1024))
    for M in [int(v) for v in os.environ.get("KBENCH_MS", "32").split(",")]:
        for (N, K) in shapes:
            copies = max(4, int(1.2e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            it = [0]

            def f():
                it[0] += 1
                return ops.stream_linear_planes(x, ws[it[0] % copies])
            res = {}
            for ksp in (0, 1, 2, 4, 8, 16):
                if ksp:
                    os.environ["SEMIPD_SL_KS"] = str(ksp)
                else:
                    os.environ.pop("SEMIPD_SL_KS", None)
                res[ksp if ksp else "default=%d" % f().ksplit] = graph_time(f, copies)
            os.environ.pop("SEMIPD_SL_KS", None)
            by = N * K * 2
            line = f"M={M:3d} N={N:6d} K={K:6d}: " + " ".join(f"{k}:{v * 1e6:.1f}" for k, v in res.items())
            best = min(res.values())
            line += f" | best {best * 1e6:.1f} us {by / best / 1e9:5.0f} GB/s"
            if N == 4096:
                res_n = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
                wn = torch.ones(N, device=dev, dtype=torch.bfloat16)

                def fn():
                    it[0] += 1
                    return ops.fused_add_rmsnorm_planes(ops.stream_linear_planes(x, ws[it[0] % copies]), res_n, wn, 1e-5)
                line += f" | +norm {graph_time(fn, copies) * 1e6:.1f} us per pair"
            print(line, flush=True)
            del ws


def bench_gemm_tall():
    """ops.gemm_tall (csrc/gemm8p.hip) vs hipBLASLt at Llama-3-8B layer shapes and the lm_head: tall decode batches
    (weight-stream bound: GB/s of weights) up to prefill-sized ones (TFLOP/s)."""
    import torch.nn.functional as F
    ncu = int(os.environ.get("KBENCH_NUM_CUS", "0"))
    if ncu:
        from semi_pd_amd import _lib
        _lib.load().semipd_gemm_tall_set_cus(ncu)
    print("# gemm_tall M x [N, K]: hipBLASLt us | gemm_tall us  TF/s  GB/s(weights)   HSA_CU_MASK=%s num_cus=%d"
          % (os.environ.get("HSA_CU_MASK", "-"), ncu))
    for M in [int(v) for v in os.environ.get("KBENCH_MS", "96,128,192,256,1024,4096").split(",")]:
        for (N, K, silu) in ((28672, 4096, True), (28672, 4096, False), (4096, 14336, False), (6144, 4096, False),
                             (4096, 4096, False), (128256, 4096, False)):
            copies = max(2, int(1.2e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            it = [0]

            def f1():
                it[0] += 1
                y = F.linear(x, ws[it[0] % copies])
                return ops.silu_and_mul(y) if silu else y

            def f2():
                it[0] += 1
                return ops.gemm_tall(x, ws[it[0] % copies], fuse_silu_mul=silu)
            t1 = timeit(f1, iters=3 * copies)
            t2 = timeit(f2, iters=3 * copies)
            fl, by = 2.0 * M * N * K, N * K * 2
            print(f"M={M:5d} N={N:6d} K={K:6d} silu={int(silu)}: blaslt {t1 * 1e6:8.1f} us {fl / t1 / 1e12:7.1f} TF | "
                  f"gemm_tall {t2 * 1e6:8.1f} us {fl / t2 / 1e12:7.1f} TF {by / t2 / 1e9:6.0f} GB/s", flush=True)
            del ws


def bench_linear_prefill():
    """Prefill-sized dense layers of Llama-3-8B through hipBLASLt under a CU mask (HSA_CU_MASK) with / without
    TENSILE_STREAMK_MAX_CUS: does the library's stream-K grid follow the share?"""
    import torch.nn.functional as F
    if os.environ.get("KBENCH_BLAS") == "rocblas":
        torch.backends.cuda.preferred_blas_library("cublas")   # = rocBLAS on ROCm
    print("# prefill linear M x [N, K] (%s): us, TFLOP/s   HSA_CU_MASK=%s  %s"
          % (os.environ.get("KBENCH_BLAS", "hipBLASLt"), os.environ.get("HSA_CU_MASK", "-"),
             " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TENSILE_"))))
    for M in [int(x) for x in os.environ.get("KBENCH_MS", "16,1024,4096,8192").split(",")]:
        for (N, K) in ((28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)):
            w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: F.linear(x, w), iters=10)
            print(f"linear M={M:5d} N={N:6d} K={K:6d}: {t * 1e6:8.1f} us {2.0 * M * N * K / t / 1e12:7.1f} TF/s")
            del w


def bench_linear_sweep():
    """ops.linear with every (NG, ksplit) forced through SEMIPD_LINEAR_NG / SEMIPD_LINEAR_KSPLIT vs hipBLASLt."""
    import torch.nn.functional as F
    print("# decode linear sweep: M, N, K: hipBLASLt us | best (NG, ksplit) us | all   HSA_CU_MASK=%s"
          % os.environ.get("HSA_CU_MASK", "-"))
    ncu = int(os.environ.get("KBENCH_NUM_CUS", "0"))
    for M in (16, 48):
        for (N, K) in ((28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)):
            copies = max(2, int(1.2e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            it = [0]

            def f1():
                it[0] += 1
                return F.linear(x, ws[it[0] % copies])

            def f2():
                it[0] += 1
                return ops.linear(x, ws[it[0] % copies], num_cus=ncu)
            t1 = timeit(f1, iters=3 * copies)
            res = {}
            for ng in (1, 2, 4):
                for ks in (1, 2, 3, 4, 6, 8):
                    if ks > 1 and (K // 256) // ks < 2:
                        continue
                    os.environ["SEMIPD_LINEAR_NG"], os.environ["SEMIPD_LINEAR_KSPLIT"] = str(ng), str(ks)
                    res[(ng, ks)] = timeit(f2, iters=2 * copies) * 1e6
            os.environ.pop("SEMIPD_LINEAR_NG"), os.environ.pop("SEMIPD_LINEAR_KSPLIT")
            t_auto = timeit(f2, iters=3 * copies) * 1e6
            best = min(res, key=res.get)
            row = " ".join(f"{ng}/{ks}:{t:.0f}" for (ng, ks), t in sorted(res.items()))
            print(f"M={M:3d} N={N:6d} K={K:6d}: blaslt {t1 * 1e6:6.1f} | best NG={best[0]} ks={best[1]} {res[best]:6.1f} "
                  f"| auto {t_auto:6.1f} | {row}")
            del ws


def bench_fp8():
    """Block-scaled fp8 at DeepSeek-V3 shapes (test_block_fp8.py:229-240): quantisation, the dense
    matmul against the bf16 library GEMM on the same shape, and the fused MoE against the bf16 fused MoE."""
    import torch.nn.functional as F
    from semi_pd_amd.layers.moe import fused_experts, fused_experts_fp8
    F8 = torch.float8_e4m3fn
    print("# per_token_group_quant_fp8 (group 128): T x H -> us, GB/s (bf16 in, fp8 + scales out)")
    for T, H in ((32, 7168), (256, 7168), (4096, 7168), (8192 * 6, 2048)):
        x = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.per_token_group_quant_fp8(x, 128), iters=20)
        print(f"quant T={T:6d} H={H}: {t * 1e6:8.1f} us {T * H * (3 + 4 / 128) / t / 1e9:6.0f} GB/s")
    print("# w8a8_block_fp8_matmul M x [N, K]: fp8 us, weight GB/s, TFLOP/s | bf16 F.linear us (weights twice the bytes)")
    for M in (1, 32, 64, 128, 512, 4096):
        for (N, K) in ((24576, 7168), (7168, 2048), (4608, 7168), (1536, 7168), (7168, 18432 // 8)):
            copies = max(2, int(0.8e9 // (N * K)))
            wq = [(torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(F8) for _ in range(copies)]
            ws = torch.rand(-(-N // 128), -(-K // 128), device=dev) * 1e-2
            wb = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(min(copies, 3))]
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            xq, xs = ops.per_token_group_quant_fp8(x, 128)
            it = [0]

            def f8():
                it[0] += 1
                return ops.w8a8_block_fp8_matmul(xq, wq[it[0] % copies], xs, ws, [128, 128], torch.bfloat16)

            def b16():
                it[0] += 1
                return F.linear(x, wb[it[0] % len(wb)])
            t1 = timeit(f8, iters=2 * copies)
            t2 = timeit(b16, iters=6)
            print(f"fp8mm M={M:4d} N={N:6d} K={K:6d}: {t1 * 1e6:8.1f} us {N * K / t1 / 1e9:6.0f} GB/s "
                  f"{2.0 * M * N * K / t1 / 1e12:7.1f} TF/s | bf16 {t2 * 1e6:8.1f} us")
            del wq, wb
    print("# fused MoE fp8 (quant + GEMM1 + silu*mul + quant + GEMM2 + sum) vs bf16: DeepSeek-V3 expert shapes / 8 experts-per-GPU subset")
    E, k, K, N = 32, 8, 7168, 2048
    w1 = (torch.randn(E, 2 * N, K, device=dev) * 100).clamp(-448, 448).to(F8)
    w2 = (torch.randn(E, K, N, device=dev) * 100).clamp(-448, 448).to(F8)
    w1s = torch.rand(E, 2 * N // 128, K // 128, device=dev) * 1e-2
    w2s = torch.rand(E, K // 128, N // 128, device=dev) * 1e-2
    w1b = torch.randn(E, 2 * N, K, device=dev, dtype=torch.bfloat16) * 0.02
    w2b = torch.randn(E, K, N, device=dev, dtype=torch.bfloat16) * 0.02
    for T in (1, 32, 256, 1024, 4096):
        x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
        tw, ti = ops.topk_softmax(torch.randn(T, E, device=dev), k, True)
        t1 = timeit(lambda: fused_experts_fp8(x, w1, w2, w1s, w2s, tw, ti, [128, 128]), iters=5)
        t2 = timeit(lambda: fused_experts(x, w1b, w2b, tw, ti), iters=5)
        flops = 2.0 * T * k * 3 * N * K
        wbytes = min(E, T * k) * 3 * N * K
        print(f"moe_fp8 T={T:5d} E={E} k={k} K={K} N={N}: {t1 * 1e6:9.1f} us {flops / t1 / 1e12:7.1f} TF/s "
              f"{wbytes / t1 / 1e9:6.0f} GB/s(fp8 weights) | bf16 {t2 * 1e6:9.1f} us")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "fp8":
        bench_fp8()
    if which == "linear_sweep":
        bench_linear_sweep()
    if which == "linear":
        bench_linear()
    if which == "stream_linear":
        bench_stream_linear()
    if which == "stream_planes":
        bench_stream_planes()
    if which == "stream_planes_graph":
        bench_stream_planes_graph()
    if which == "gemm_tall":
        bench_gemm_tall()
    if which == "linear_prefill":
        bench_linear_prefill()
    if which == "decode_small":
        bench_decode_small()
    if which == "mla_small":
        bench_mla_small()
    if which in ("decode", "all"):
        bench_decode()
    if which in ("mla", "all"):
        bench_mla()
    if which in ("extend", "all"):
        bench_extend()
    if which in ("norm", "all"):
        bench_norm()
    if which in ("moe", "all"):
        bench_moe()
    if which in ("lm_head", "all"):
        bench_lm_head()
