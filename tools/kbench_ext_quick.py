"""Quick extend-attention timing at two shapes (dev tool for ablations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
dev = torch.device("cuda:0")

def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

Hq, Hkv, D = 32, 8, 128
for B, ext, pre in [(1, 1024, 0), (8, 1024, 0), (1, 8192, 0)]:
    T = B * ext
    q = torch.randn(T, Hq, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    kb = torch.randn(8, Hkv, D, device=dev, dtype=torch.bfloat16)
    qo = torch.arange(B + 1, device=dev, dtype=torch.int32) * ext
    kvp = torch.zeros(B + 1, device=dev, dtype=torch.int32)
    idx = torch.zeros(1, device=dev, dtype=torch.int32)
    t = timeit(lambda: ops.extend_attention_fwd(q, k, v, o, kb, kb, qo, kvp, idx, None, None, ext))
    flops = 4.0 * Hq * D * B * ext * (pre + (ext + 1) / 2)
    print(f"ABL={os.environ.get('SEMIPD_SKV_ABL', '0')} B={B} ext={ext}: {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s", flush=True)
