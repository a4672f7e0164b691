"""Decode batches of 65 .. 128 requests, the four dense layers of Llama-3-8B: the tiled GEMM (ops.gemm_tall, csrc/gemm8p.hip) against
the weight-streaming kernel's wide form (ops.stream_linear, csrc/stream_linear.hip: 2 x 16 weight rows per wave, activation ring of
two blocks under weight rings of three), each in a hipGraph of 4 launches, alternating; 64 rows for scale.
KBENCH_KS=a,b,c also times the wide form with a forced K split (SEMIPD_SL_KS is read per call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
dev = torch.device("cuda:0")
REP = 4


def graph_time(fn, iters=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(REP):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) / iters / REP * 1e3


cus = int(os.environ.get("KBENCH_CUS", "256"))
ops._lib.load().semipd_stream_linear_set_cus(cus)
ops._lib.load().semipd_gemm_tall_set_cus(cus)
rows = [int(r) for r in os.environ.get("KBENCH_ROWS", "64,80,96,128").split(",")]
forced = [int(k) for k in os.environ.get("KBENCH_KS", "").split(",") if k]
print(f"# us per call (hipGraph of {REP}); streaming kernel | tiled GEMM" + "".join(f" | streaming, {k} K slices" for k in forced)
      + f"   [TB/s of weights, streaming kernel]   HSA_CU_MASK={os.environ.get('HSA_CU_MASK', '-')} declared CUs={cus}")
tot = {}
for name, N, K, silu in (("qkv", 6144, 4096, False), ("o_proj", 4096, 4096, False), ("gate_up+silu", 28672, 4096, True),
                         ("down", 4096, 14336, False)):
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.01
    for M in rows:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        os.environ.pop("SEMIPD_SL_KS", None)
        t_s = min(graph_time(lambda: ops.stream_linear(x, w, fuse_silu_mul=silu)) for _ in range(2))
        t_t = min(graph_time(lambda: ops.gemm_tall(x, w, fuse_silu_mul=silu)) for _ in range(2)) if M > 64 else float("nan")
        extra = []
        for k in forced:
            os.environ["SEMIPD_SL_KS"] = str(k)
            extra.append(min(graph_time(lambda: ops.stream_linear(x, w, fuse_silu_mul=silu)) for _ in range(2)))
        os.environ.pop("SEMIPD_SL_KS", None)
        tot.setdefault(M, [0.0, 0.0])
        tot[M][0] += t_s
        tot[M][1] += t_t
        print(f"{name:13s} M={M:4d}: {t_s:7.1f} | {t_t:7.1f}" + "".join(f" | {e:7.1f}" for e in extra)
              + f"   [{N * K * 2 / (t_s * 1e-6) / 1e12:.2f}]", flush=True)
    del w
for M in rows:
    print(f"layer total   M={M:4d}: {tot[M][0]:7.1f} | {tot[M][1]:7.1f}   [{436.2e6 / (tot[M][0] * 1e-6) / 1e12:.2f}]")
