"""GPU tests of the IPC seam (semi_pd_ipc drop-in) and CU-mask isolation.

The reference has no test for semi-pd-ipc; the known-answer test is the round trip:
export in process A, import in process B, compare bytes, write back, observe in A."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _child_roundtrip(conn, pkg_paths):
    try:
        for p in pkg_paths:
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch
        import semi_pd_ipc
        torch.cuda.set_device(0)
        items = conn.recv()
        report = {}
        tensors = []
        for name, (handle, offset), numel, dtype_str, shape, checksum in items:
            t = semi_pd_ipc.convert_ipc_handle_to_tensor((handle, offset), numel, dtype_str, torch.device("cuda:0"))
            t = t.view(shape)
            got = float(t.double().sum().item())
            report[name] = (tuple(t.shape), str(t.dtype), got, abs(got - checksum) < 1e-3 * max(1.0, abs(checksum)))
            tensors.append(t)
        report["mappings"] = semi_pd_ipc.num_open_mappings()
        # write through the mapping: parent must see it
        tensors[0].fill_(7)
        torch.cuda.synchronize()
        conn.send(report)
        conn.recv()  # wait until the parent has checked
        for t in tensors:
            semi_pd_ipc.close_ipc_tensor(t)
        conn.send({"mappings_after_close": semi_pd_ipc.num_open_mappings()})
    except Exception as e:  # pragma: no cover
        import traceback
        conn.send({"error": traceback.format_exc()})


def test_ipc_roundtrip_between_processes(device):
    import semi_pd_ipc
    from conftest import PKG, ROOT
    ctx = mp.get_context("spawn")
    parent, child = ctx.Pipe()
    p = ctx.Process(target=_child_roundtrip, args=(child, [ROOT, PKG]))
    p.start()
    torch.manual_seed(0)
    # several tensors; the small ones share one caching-allocator segment -> same handle, different offsets
    a = torch.arange(1000, dtype=torch.float32, device=device)
    b = torch.randn(33, 7, device=device).to(torch.bfloat16)
    c = torch.randint(0, 100, (5, 11), dtype=torch.int32, device=device)
    big = torch.randn(64 << 20 >> 1, device=device).to(torch.bfloat16)  # 64 MiB: its own allocation
    torch.cuda.synchronize()
    items = []
    handles = set()
    for name, t in (("a", a), ("b", b), ("c", c), ("big", big)):
        h, off = semi_pd_ipc.get_ipc_handle_and_offset(t)
        assert len(h) == 64 and all(0 <= x < 256 for x in h)
        assert semi_pd_ipc.get_ipc_handle(t) == h
        handles.add(tuple(h))
        dtype_str = {torch.float32: "at::kFloat", torch.bfloat16: "at::kBFloat16", torch.int32: "at::kInt"}[t.dtype]
        items.append((name, (h, off), t.numel(), dtype_str, tuple(t.shape), float(t.double().sum().item())))
    parent.send(items)
    report = parent.recv()
    assert "error" not in report, report.get("error")
    for name in ("a", "b", "c", "big"):
        assert report[name][3], f"{name}: child saw different contents {report[name]}"
    assert report["mappings"] == len(handles)  # one mapping per allocation, not per tensor
    torch.cuda.synchronize()
    assert float(a.sum().item()) == 7000.0  # child's write is visible here
    parent.send("ok")
    after = parent.recv()
    assert "error" not in after, after.get("error")
    assert after["mappings_after_close"] == 0
    p.join(30)
    assert p.exitcode == 0


def test_convert_rejects_bad_input(device):
    import semi_pd_ipc
    with pytest.raises(ValueError):
        semi_pd_ipc.convert_ipc_handle_to_tensor(([0] * 64, 0), 4, "at::kNoSuchType", device)
    with pytest.raises(ValueError):
        semi_pd_ipc.convert_ipc_handle_to_tensor(([0] * 10, 0), 4, "at::kFloat", device)
    with pytest.raises(RuntimeError):  # garbage handle: hipIpcOpenMemHandle fails, we raise (reference ignores)
        semi_pd_ipc.convert_ipc_handle_to_tensor(([1] * 64, 0), 4, "at::kFloat", device)


def test_device_cu_count(device):
    import semi_pd_ipc
    n = semi_pd_ipc.get_device_sm_count(0)
    assert n == torch.cuda.get_device_properties(0).multi_processor_count
    assert n >= 64


def _placement(stream_ptr, nwg=4096, spin=20000):
    from semi_pd_amd import _lib
    lib = _lib.load()
    out = torch.full((nwg, 2), -1, dtype=torch.int32, device="cuda:0")
    _lib.check(lib.semipd_probe_cu_placement(out.data_ptr(), nwg, spin, stream_ptr), "probe")
    torch.cuda.synchronize()
    o = out.cpu()
    return {(int(x), int(c)) for x, c in o.tolist()}


def test_cu_masked_stream_confines_workgroups(device):
    """A stream created with half the CU bits runs on half the (XCD, CU) slots, spread over all XCDs,
    and the two halves of a prefill/decode split are disjoint."""
    from semi_pd_amd import _lib
    import semi_pd_ipc
    lib = _lib.load()
    ncu = semi_pd_ipc.get_device_sm_count(0)
    words = (ncu + 31) // 32
    full = _placement(None)
    assert len(full) >= ncu * 0.9, f"unmasked launch only touched {len(full)} CU slots"
    seen = {}
    for from_top in (0, 1):
        mask = (C.c_uint32 * words)()
        n = lib.semipd_cu_mask_fill(ncu, 50, from_top, C.addressof(mask), words)
        assert n == ncu // 2
        s = C.c_void_p(0)
        _lib.check(lib.semipd_stream_create_cu_mask(0, C.addressof(mask), words, C.addressof(s)), "create")
        back = (C.c_uint32 * words)()
        _lib.check(lib.semipd_stream_get_cu_mask(s, C.addressof(back), words), "get mask")
        assert list(back) == list(mask)
        slots = _placement(s)
        seen[from_top] = slots
        assert len(slots) <= n + 2, f"masked stream ran on {len(slots)} CU slots, mask has {n}"
        assert len({x for x, _ in slots}) == len({x for x, _ in full}), "mask is not spread over all XCDs"
        _lib.check(lib.semipd_stream_destroy(s), "destroy")
    assert not (seen[0] & seen[1]), "prefill and decode CU sets overlap"


def test_export_refuses_allocation_sizes_the_importer_cannot_map(device):
    """ROCm 7.2 dmabuf IPC: hipIpcOpenMemHandle never returns when the allocation size modulo 4 GiB is
    >= 2 GiB (tools/ipc_big_probe.py).  The exporter must fail loudly instead of hanging the prefill
    instance at start-up, and the pools must size their slabs around it."""
    import semi_pd_ipc
    from semi_pd_amd.mem_cache.memory_pool import ipc_safe_zeros
    bad = torch.empty(int(2.5 * (1 << 30)), dtype=torch.uint8, device=device)
    with pytest.raises(RuntimeError, match="cannot be imported"):
        semi_pd_ipc.get_ipc_handle_and_offset(bad)
    del bad
    torch.cuda.empty_cache()
    # the same number of payload bytes through the pool allocator: padded to 4 GiB, exportable
    t = ipc_safe_zeros((32, 2, 40001, 8, 128), torch.float8_e5m2, device)
    assert t.numel() == 32 * 2 * 40001 * 8 * 128 and t.dtype == torch.float8_e5m2
    handle, off = semi_pd_ipc.get_ipc_handle_and_offset(t[3, 1])
    assert len(handle) == 64 and off == t[3, 1].data_ptr() - t.data_ptr()
    h2, _ = semi_pd_ipc.get_ipc_handle_and_offset(t[5, 0])
    assert h2 == handle  # one allocation, one handle: the importer's mapping cache depends on it
