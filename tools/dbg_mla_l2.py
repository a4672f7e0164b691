"""MLA shared-tile kernel: the same shape with every row index inside a 1 MiB window (L2 hits) against random rows."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semi-pd_amd"))
from semi_pd_amd import ops
dev = torch.device("cuda:0")
B, ctx, H, splits = 128, 8192, 128, 2
N = B * ctx + 1
kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
for name, idx in (("random rows", (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)),
                  ("1 MiB window", torch.randint(1, 900, (B * ctx,), device=dev, dtype=torch.int32)),
                  ("sequential rows", torch.arange(1, B * ctx + 1, device=dev, dtype=torch.int32))):
    for _ in range(3):
        ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(10):
        ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
    e.record(); torch.cuda.synchronize()
    print(f"{name:16s} {s.elapsed_time(e) / 10 * 1e3:8.1f} us (stage 1 + stage 2)")
