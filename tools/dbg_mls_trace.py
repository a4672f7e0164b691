"""Per-phase cycle sums of the MLA shared-tile kernel's tile loop (library built with -DMLS_TRACE)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semi-pd_amd"))
from semi_pd_amd import ops, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
B, ctx, H, splits = 128, 8192, 128, 2
N = B * ctx + 1
kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
for _ in range(3):
    ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.semipd_debug_mls_trace(None, 1)
ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)
torch.cuda.synchronize()
lib.semipd_debug_mls_trace(buf, 0)
tiles = ctx // splits // 32
names = ["loop edge", "vm wait", "barrier", "QK + dma", "softmax", "PV", "idx wait + rotate", "-"]
tot = sum(buf[:7])
print(f"tiles {tiles}; cycles per tile (s_memtime ticks, 100 MHz?) total {tot / tiles:.1f}")
for n, v in zip(names, buf):
    print(f"  {n:20s} {v / tiles:9.1f}")
