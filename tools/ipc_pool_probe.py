"""Probe: MHATokenToKVPool slab (fp8 / bf16) exported per layer, imported in a child process."""
import os
import sys
import time
import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semi-pd_amd")]


def child(q_in, q_out):
    import torch
    from semi_pd_amd.semi_pd.utils import convert_ipc_handle_to_tensor
    torch.cuda.set_device(0)
    while True:
        item = q_in.get()
        if item is None:
            return
        handles, numel, dtype = item
        t0 = time.time()
        s = 0
        for h in handles:
            t = convert_ipc_handle_to_tensor(h, numel, dtype, "cuda:0")
            s += int(t.view(torch.uint8)[:4].sum().item())
        import semi_pd_ipc
        q_out.put((time.time() - t0, s, semi_pd_ipc.num_open_mappings()))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    import torch
    from semi_pd_amd.mem_cache.memory_pool import MHATokenToKVPool
    from semi_pd_amd.semi_pd.utils import get_ipc_handle
    q_in, q_out = mp.Queue(), mp.Queue()
    p = mp.Process(target=child, args=(q_in, q_out))
    p.start()
    junk = [torch.randn(4096, 4096, device="cuda:0", dtype=torch.bfloat16) for _ in range(64)]
    for dtype, tokens in ((torch.float8_e5m2, 20000), (torch.bfloat16, 100000), (torch.uint8, 40000), (torch.float8_e5m2, 40000)):
        pool = MHATokenToKVPool(tokens, 1, dtype, 8, 128, 32, "cuda:0")
        pool.slab.view(torch.uint8).view(-1)[:4] = 1
        torch.cuda.synchronize()
        handles = [get_ipc_handle(pool.k_buffer[i]) for i in range(32)] + [get_ipc_handle(pool.v_buffer[i]) for i in range(32)]
        print("distinct handles:", len({tuple(h[0]) for h in handles}), flush=True)
        print(dtype, tokens, "slab bytes", pool.slab.numel() * pool.slab.element_size(), "offsets", handles[0][1], handles[1][1], flush=True)
        q_in.put((handles, pool.k_buffer[0].numel(), dtype))
        try:
            dt, s, nopen = q_out.get(timeout=40)
            print(f"   imported 64 views in {dt * 1e3:.1f} ms, checksum {s}, open mappings in child {nopen}", flush=True)
        except Exception as e:
            print(f"   TIMEOUT / {e!r}", flush=True)
            break
        del pool
        torch.cuda.empty_cache()
    q_in.put(None)
    p.join(5)
    if p.is_alive():
        p.kill()
