"""One rank of the op-set test: drives semi_pd_amd/sgl_kernel_allreduce.py in the order the reference's CustomAllreduce
does on ROCm (custom_all_reduce.py:283-302 constructor, :417-426 registration, :490-501 calls, :554-562 close) and checks
the sums against the oracle bit for bit, eagerly and from a replayed hipGraph.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-pd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def gather_ipc_meta(shard, rank, world):
    """custom_all_reduce.py:393-415: broadcast_object_list from every rank on the CPU group."""
    all_data = [[None] for _ in range(world)]
    all_data[rank][0] = shard
    for r in range(world):
        dist.broadcast_object_list(all_data[r], src=r, device="cpu")
    return [d[0][0] for d in all_data], [d[0][1] for d in all_data]


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.set_num_threads(4)
    from oracle.ops import all_reduce_sum
    from semi_pd_amd import sgl_kernel_allreduce as ops
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    max_size = 2 << 20
    meta = ops.allocate_meta_buffer(ops.meta_size() + max_size)
    buffer = torch.empty(max_size, dtype=torch.uint8, device=dev)
    handle = ops.get_meta_buffer_ipc_handle(meta)
    assert handle.dtype == torch.uint8 and handle.numel() == 64 and not handle.is_cuda
    handles, offsets = gather_ipc_meta((bytes(handle.numpy().tobytes()), 0), rank, world)
    rank_data = torch.empty(8 * 1024 * 1024, dtype=torch.uint8, device=dev)
    fa = ops.init_custom_ar(meta, rank_data, handles, offsets, rank, True)
    bh, bo = gather_ipc_meta((bytes(ops.get_meta_buffer_ipc_handle(meta).numpy().tobytes()), 0), rank, world)
    ops.register_buffer(fa, buffer, bh, bo)
    dist.barrier()
    report = {"rank": rank, "cases": 0, "bad": []}

    def inputs_of(numel, dtype, seed):
        g = torch.Generator().manual_seed(seed)
        return [(torch.randn(numel, generator=g) * 3).to(dtype) for _ in range(world)]

    def check(name, got, want):
        report["cases"] += 1
        if not torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8)):
            report["bad"].append(f"{name}: max abs diff {(got.cpu().float() - want.float()).abs().max().item()}")

    seed = 0
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for nbytes in (16, 4096, 256 * 1024, max_size):
            numel = nbytes // torch.empty((), dtype=dtype).element_size()
            xs = inputs_of(numel, dtype, seed)
            seed += 1
            x = xs[rank].to(dev)
            out = torch.empty_like(x)
            ops.all_reduce_reg(fa, x, out)
            check(f"reg {dtype} {nbytes}", out, all_reduce_sum(xs, dtype))
            out2 = torch.empty_like(x)
            ops.all_reduce_unreg(fa, x, buffer, out2)
            check(f"unreg {dtype} {nbytes}", out2, all_reduce_sum(xs, dtype))
    # a captured call, registered afterwards the way custom_all_reduce.py:421-426 does, replayed on new data
    xs = inputs_of(8192, torch.bfloat16, 99)
    x = xs[rank].to(dev)
    out = torch.empty_like(x)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        ops.all_reduce_reg(fa, x, out)
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        ops.all_reduce_reg(fa, x, out)
    gh, go = ops.get_graph_buffer_ipc_meta(fa)
    assert gh.numel() == 0 and go == []
    hs, offs = gather_ipc_meta((bytes(gh.numpy().tobytes()), go), rank, world)
    ops.register_graph_buffers(fa, hs, offs)
    for s in (100, 101):
        xs = inputs_of(8192, torch.bfloat16, s)
        x.copy_(xs[rank])
        torch.cuda.synchronize()
        dist.barrier()
        g.replay()
        torch.cuda.synchronize()
        check(f"graph replay {s}", out, all_reduce_sum(xs, torch.bfloat16))
    # error behaviour of the reference's checks
    try:
        ops.all_reduce_unreg(fa, torch.zeros(64, device=dev), torch.empty(16, dtype=torch.uint8, device=dev), torch.zeros(64, device=dev))
        report["bad"].append("a too small registered buffer was accepted")
    except RuntimeError as e:
        assert "too small" in str(e)
    dist.barrier()
    ops.dispose(fa)
    ops.free_meta_buffer(meta)
    print("AR_REPORT " + json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
