"""down_proj / o_proj of a prefill batch followed by the fused add + RMSNorm, Llama-3-8B shapes: the tiled GEMM + its
reduction launch + fused_add_rmsnorm against gemm_tall_planes + fused_add_rmsnorm_planes (the norm sums the K slices).
Both forms in a hipGraph of 8 pairs (no host time between launches), alternating, on whatever CU mask the process has."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
dev = torch.device("cuda:0")
REP = 8


def graph_time(fn, iters=30):
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


cus = int(os.environ.get("KBENCH_CUS", "0"))
if cus:
    ops._lib.load().semipd_gemm_tall_set_cus(cus)
print(f"# tiled GEMM -> fused add + norm, us per pair (hipGraph of {REP} pairs): reducing form | planes form   "
      f"HSA_CU_MASK={os.environ.get('HSA_CU_MASK', '-')} declared CUs={cus or 256}")
for (M, N, K) in ((1024, 4096, 14336), (1357, 4096, 14336), (1536, 4096, 14336), (1024, 4096, 4096), (512, 4096, 14336)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.01
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(N, device=dev, dtype=torch.bfloat16)

    def reducing():
        y = ops.gemm_tall(x, w)
        ops.fused_add_rmsnorm(y, res, nw, 1e-5)

    ks = [0]

    def planes():
        p = ops.gemm_tall_planes(x, w)
        if isinstance(p, ops.SplitKPlanes):
            ks[0] = p.ksplit
            ops.fused_add_rmsnorm_planes(p, res, nw, 1e-5)
        else:
            ks[0] = 1
            ops.fused_add_rmsnorm(p, res, nw, 1e-5)

    row = []
    for _ in range(2):
        row.append((graph_time(reducing), graph_time(planes)))
    print(f"M={M:5d} N={N} K={K:5d} ksplit={ks[0]}: " + "  ".join(f"{a:7.1f} | {b:7.1f}" for a, b in row), flush=True)
