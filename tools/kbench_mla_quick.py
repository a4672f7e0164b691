"""MLA decode at 64 / 128 heads: the shared-tile kernel against the wide one (SEMIPD_MLA_SHARED=0 in a second run)."""
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


print("# MLA decode, bf16 rows: B ctx H splits -> us, GB/s of rows (+ q, o), TFLOP/s")
for H in (128, 64):
    for B, ctx in ((1, 8192), (8, 8192), (32, 1024), (32, 8192), (128, 1024), (128, 8192), (256, 2048)):
        N = B * ctx + 1
        kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
        q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
        o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
        indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
        idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
        best = None
        for splits in (1, 2, 4, 8, 16, 32, 64):
            if ctx // splits < 64 or B * (H // 64) * splits > 8192:
                continue
            lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
            t = timeit(lambda: ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1))
            if best is None or t < best[0]:
                best = (t, splits)
        t, splits = best
        nbytes = B * ctx * 1152 + B * H * (576 + 512) * 2
        flop = 2.0 * B * ctx * H * (576 + 512)
        print(f"mla H={H:3d} B={B:4d} ctx={ctx:5d} splits={splits:2d}: {t * 1e6:8.1f} us {nbytes / t / 1e9:7.0f} GB/s "
              f"{flop / t / 1e12:6.0f} TFLOP/s", flush=True)
