"""MLA decode at serving-sized contexts (1-2 k) and batches: shared-tile kernel (forced) vs wide kernel, per split count."""
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
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


H = 128
print("# us per call (stage 1 + stage 2): rows = B ctx, columns = splits; S = shared-tile kernel forced, W = wide kernel")
for B, ctx in ((8, 1100), (32, 1100), (64, 1100), (96, 1100), (32, 2200), (64, 2200), (16, 4400), (32, 4400)):
    N = B * ctx + 1
    kv = torch.randn(N, 1, 576, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B, H, 576, device=dev, dtype=torch.bfloat16)
    o = torch.empty(B, H, 512, device=dev, dtype=torch.bfloat16)
    indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * ctx
    idx = (torch.randperm(N - 1, device=dev)[: B * ctx] + 1).to(torch.int32)
    for mode, tag in (("2", "S"), ("0", "W")):
        os.environ["SEMIPD_MLA_SHARED"] = mode
        row = []
        for splits in (1, 2, 4, 8, 16):
            lg = torch.empty(B, H, splits, 513, device=dev, dtype=torch.float32)
            row.append(timeit(lambda: ops.decode_attention_fwd(q, kv, kv[..., :512], o, indptr, idx, lg, splits, 0.1)))
        print(f"B={B:3d} ctx={ctx:5d} {tag}: " + "  ".join(f"{t:7.1f}" for t in row), flush=True)
os.environ.pop("SEMIPD_MLA_SHARED", None)
