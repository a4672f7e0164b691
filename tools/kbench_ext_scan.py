"""Extend attention, ONE request of growing length: fixed cost vs per-tile cost (dev tool)."""
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

Hq, Hkv, D = 32, 8, 128
g = torch.cuda.CUDAGraph()
for ext in (32, 128, 256, 512, 768, 1024, 1536, 2048):
    q = torch.randn(ext, Hq, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(ext, Hkv, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(ext, Hkv, D, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    kb = torch.randn(8, Hkv, D, device=dev, dtype=torch.bfloat16)
    qo = torch.tensor([0, ext], device=dev, dtype=torch.int32)
    kvp = torch.zeros(2, device=dev, dtype=torch.int32)
    idx = torch.zeros(1, device=dev, dtype=torch.int32)
    fn = lambda: ops.extend_attention_fwd(q, k, v, o, kb, kb, qo, kvp, idx, None, None, ext)
    t = timeit(fn)
    # the same launch replayed from a graph of 20: no host launch gaps
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(20):
            fn()
    tg = timeit(gr.replay, iters=10) / 20
    print(f"ext={ext:5d}: eager {t:6.1f} us | back to back in a graph {tg:6.1f} us", flush=True)
