"""Fused MoE of DeepSeek-V2-Lite (64 experts top-6, K 2048, N 1408) at prefill sizes with the tiled grouped GEMM in its 8-wave and
4-wave forms (csrc/gemm8p.hip), whole pipeline (align + GEMM1 with SiLU * mul + GEMM2 + sum) through layers.moe.fused_experts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
from semi_pd_amd.layers.moe import fused_experts
dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


E, k, K, N = 64, 6, 2048, 1408
w1 = torch.randn(E, 2 * N, K, device=dev, dtype=torch.bfloat16) * 0.02
w2 = torch.randn(E, K, N, device=dev, dtype=torch.bfloat16) * 0.02
print("# fused MoE us (PFLOP/s): form 8 | form 4 | form 0 (by epilogue), twice")
for T in (2048, 4096, 8192, 16384):
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    tw, ti = ops.topk_softmax(torch.randn(T, E, device=dev), k, True)
    flops = 2.0 * T * k * 3 * N * K
    row = []
    for _ in range(2):
        for form in (8, 4, 0):
            ops.gemm_tall_set_form(form)
            t = timeit(lambda: fused_experts(x, w1, w2, tw, ti))
            row.append(f"{t:7.0f} ({flops / t / 1e9:.2f})")
    ops.gemm_tall_set_form(0)
    print(f"T={T:5d}: " + " | ".join(row), flush=True)
