"""MLA decode with 16 heads per rank (DeepSeek-V2-Lite; DeepSeek-V3 at TP = 8): decode_attention_fwd (stage 1 + stage 2) per
split count at serving shapes; GB/s of latent rows against the 8 TB/s roofline (0.21 -> 0.4)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semi-pd_amd"))
from semi_pd_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


H = 16
print("# MLA decode, 16 heads, bf16 rows: B ctx splits -> us (stage 1 + stage 2), GB/s of rows, fraction of 8 TB/s")
for B, ctx in ((32, 1100), (32, 4096), (64, 1100), (8, 1100), (128, 1100)):
    N = B * ctx + 1
    kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
    o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    for splits in (2, 4, 8, 16, 32):
        if ctx // splits < 32:
            continue
        lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
        t = timeit(lambda: ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1))
        nbytes = B * ctx * 1152
        print(f"mla16 B={B:4d} ctx={ctx:5d} splits={splits:2d}: {t * 1e6:7.1f} us {nbytes / t / 1e9:6.0f} GB/s  {nbytes / t / 8e12:5.2f}", flush=True)
