"""Per-stage timing of the fused-MoE pipeline at prefill sizes (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
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
for T in (2048, 4096, 8192):
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    tw, ti = ops.topk_softmax(torch.randn(T, E, device=dev), k, True)
    numel, bm = T * k, 128
    max_sorted = numel + E * (bm - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=dev)
    expert_ids = torch.empty((max_sorted + bm - 1) // bm, dtype=torch.int32, device=dev)
    npp = torch.empty(1, dtype=torch.int32, device=dev)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=dev)
    c1 = torch.empty(numel, 2 * N, dtype=torch.bfloat16, device=dev)
    c3 = torch.empty(numel, K, dtype=torch.bfloat16, device=dev)
    t_al = timeit(lambda: ops.moe_align_block_size(ti, E, bm, sorted_ids, expert_ids, npp, None, cumsum))
    t_g1 = timeit(lambda: ops.moe_grouped_gemm(x, w1, c1, None, sorted_ids, expert_ids, npp, numel, k, False, bm))
    c2 = ops.silu_and_mul(c1)
    t_si = timeit(lambda: ops.silu_and_mul(c1))
    t_g2 = timeit(lambda: ops.moe_grouped_gemm(c2, w2, c3, tw.reshape(-1), sorted_ids, expert_ids, npp, numel, 1, True, bm))
    t_su = timeit(lambda: ops.moe_sum(c3.view(T, k, K)))
    f1, f2 = 2.0 * numel * 2 * N * K, 2.0 * numel * N * K
    print(f"T={T}: align {t_al:.0f} us | GEMM1 {t_g1:.0f} us {f1 / t_g1 / 1e6:.0f} TF/s | silu {t_si:.0f} us | "
          f"GEMM2 {t_g2:.0f} us {f2 / t_g2 / 1e6:.0f} TF/s | sum {t_su:.0f} us | total {t_al + t_g1 + t_si + t_g2 + t_su:.0f} us", flush=True)
