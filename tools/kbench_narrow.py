"""The streaming GEMM's workgroup width at decode batch sizes, small output widths: eight waves (128 weight rows per workgroup) against
four (64 rows, two workgroups per CU), by K split; planes form (no reduction launch), each in a hipGraph of 4 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
from semi_pd_amd import ops
dev = torch.device("cuda:0")
REP = 4


def graph_time(fn, iters=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(REP):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) / iters / REP * 1e3


cus = int(os.environ.get("KBENCH_CUS", "256"))
ops._lib.load().semipd_stream_linear_set_cus(cus)
KS = [0, 2, 3, 4, 6, 8]
print(f"# us per call (planes form, hipGraph of {REP}); columns: K slices " + " ".join(f"{k or 'auto':>6}" for k in KS)
      + f"   HSA_CU_MASK={os.environ.get('HSA_CU_MASK', '-')} declared CUs={cus}")
for name, N, K in (("qkv", 6144, 4096), ("o_proj", 4096, 4096), ("down", 4096, 14336), ("qkv 70b/tp8", 1280, 8192),
                   ("o 70b/tp8", 8192, 1024), ("down 70b/tp8", 8192, 3584)):
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.01 for _ in range(3)]   # rotate: not out of L2 / MALL
    for M in [int(m) for m in os.environ.get("KBENCH_ROWS", "16,32").split(",")]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        for nw in [int(v) for v in os.environ.get("KBENCH_NW", "8,4").split(",")]:
            os.environ["SEMIPD_SL_NW"] = str(nw)
            row = []
            for ks in KS:
                if ks:
                    os.environ["SEMIPD_SL_KS"] = str(ks)
                else:
                    os.environ.pop("SEMIPD_SL_KS", None)
                i = [0]

                def fn():
                    i[0] += 1
                    return ops.stream_linear_planes(x, ws[i[0] % 3])
                row.append(min(graph_time(fn) for _ in range(2)))
            os.environ.pop("SEMIPD_SL_KS", None)
            best = min(row)
            print(f"{name:13s} M={M:3d} waves={nw}: " + " ".join(f"{t:6.1f}" for t in row)
                  + f"   best {best:5.1f} us = {N * K * 2 / (best * 1e-6) / 1e12:.2f} TB/s", flush=True)
    del ws
os.environ.pop("SEMIPD_SL_NW", None)
