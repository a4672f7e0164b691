"""Isolated timings of the small prefill-side kernels at ~1 k tokens, Llama-3-8B shapes (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
dev = torch.device("cuda:0")

def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for T in (1024, 1184, 4096):
    Hq, Hk, D = 32, 8, 128
    qkv = torch.randn(T, (Hq + 2 * Hk) * D, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv.split([Hq * D, Hk * D, Hk * D], dim=-1)
    pos = torch.arange(T, device=dev, dtype=torch.int64)
    cache = torch.randn(8192, D, device=dev, dtype=torch.float32)
    kb = torch.empty(T + 8, Hk, D, device=dev, dtype=torch.bfloat16)
    vb = torch.empty(T + 8, Hk, D, device=dev, dtype=torch.bfloat16)
    loc = torch.randperm(T, device=dev).to(torch.int64)
    t_rope = timeit(lambda: ops.rope_and_store_kv(pos, q, k, v, D, cache, True, kb, vb, loc))
    x = torch.randn(T, 4096, device=dev, dtype=torch.bfloat16); r = torch.randn_like(x); w = torch.ones(4096, device=dev, dtype=torch.bfloat16)
    t_norm = timeit(lambda: ops.fused_add_rmsnorm(x, r, w, 1e-5))
    g = torch.randn(T, 2 * 14336, device=dev, dtype=torch.bfloat16)
    t_silu = timeit(lambda: ops.silu_and_mul(g))
    print(f"T={T}: rope+kv store {t_rope:.1f} us ({T * 26 * 1024 / t_rope / 1e6:.2f} TB/s) | fused_add_rmsnorm {t_norm:.1f} us ({4 * T * 4096 * 2 / t_norm / 1e6:.2f} TB/s) | silu_and_mul {t_silu:.1f} us ({3 * T * 14336 * 2 / t_silu / 1e6:.2f} TB/s)", flush=True)
