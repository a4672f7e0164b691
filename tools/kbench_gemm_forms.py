"""Prefill-sized dense layers of Llama-3-8B: the library GEMM (F.linear; + silu_and_mul for gate_up) against the tiled GEMM in its
8-wave and 4-wave forms (csrc/gemm8p.hip), each in a hipGraph of 4 launches (no host time between them), alternating, on whatever
CU mask the process has (HSA_CU_MASK; KBENCH_CUS declares the share to the K-split rule)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]
import torch
import torch.nn.functional as F
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


cus = int(os.environ.get("KBENCH_CUS", "0"))
if cus:
    ops._lib.load().semipd_gemm_tall_set_cus(cus)
rows = [int(r) for r in os.environ.get("KBENCH_ROWS", "1024,1411,2048,4096,8192").split(",")]
print(f"# us per call (hipGraph of {REP}); library | tiled 8 waves | tiled 4 waves   [PFLOP/s of the 4-wave form]   "
      f"HSA_CU_MASK={os.environ.get('HSA_CU_MASK', '-')} declared CUs={cus or 256}")
for name, N, K, silu in (("gate_up+silu", 28672, 4096, True), ("down", 4096, 14336, False), ("qkv", 6144, 4096, False),
                         ("o_proj", 4096, 4096, False), ("gate_up", 28672, 4096, False)):
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.01
    for M in rows:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)

        def lib():
            y = F.linear(x, w)
            return ops.silu_and_mul(y) if silu else y

        def tall():
            return ops.gemm_tall(x, w, fuse_silu_mul=silu)

        res = []
        for _ in range(2):
            t_lib = graph_time(lib)
            ops.gemm_tall_set_form(8)
            t8 = graph_time(tall)
            ops.gemm_tall_set_form(4)
            t4 = graph_time(tall)
            res.append((t_lib, t8, t4))
        ops.gemm_tall_set_form(0)
        pf = 2.0 * M * N * K / (min(r[2] for r in res) * 1e-6) / 1e15
        print(f"{name:13s} M={M:5d}: " + "   ".join(f"{a:7.1f} | {b:7.1f} | {c:7.1f}" for a, b, c in res) + f"   [{pf:.2f}]", flush=True)
        del x
    del w
