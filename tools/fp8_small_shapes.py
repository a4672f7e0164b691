import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/semi-pd_amd"]
import torch
from semi_pd_amd import ops
sys.path.insert(0, "/root/repo/tools")
dev = torch.device("cuda:0")
def timeit(fn, iters=200, warmup=10):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters // 20): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters // 20 * 20) * 1e3
F8 = torch.float8_e4m3fn
for M in (8, 43, 64):
    for (N, K) in ((3072, 2048), (576, 2048), (2048, 2048), (5632, 2048), (2048, 2816), (1536, 7168), (7168, 2048)):
        wq = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(F8)
        ws = torch.rand(-(-N // 128), -(-K // 128), device=dev) * 1e-2
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        xq, xs = ops.per_token_group_quant_fp8(x, 128)
        t = timeit(lambda: ops.w8a8_block_fp8_matmul(xq, wq, xs, ws, [128, 128], torch.bfloat16))
        print(f"M={M:3d} N={N:5d} K={K:5d} ksplit={os.environ.get('SEMIPD_FP8_KSPLIT','auto'):>4s}: {t:6.1f} us", flush=True)
