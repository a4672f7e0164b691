"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/semipd.h declares; argument errors come back as RuntimeError; no CPU fallback exists."""
import os
import re

import pytest
import torch

from conftest import PKG, ROOT
from semi_pd_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "semipd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(semipd_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.semipd_version() >= 100


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 30
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/semipd.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes table and header disagree"


def test_argument_errors_are_reported_not_swallowed():
    lib = _lib.load()
    rc = lib.semipd_rmsnorm(None, None, None, 4, 0, 0, 0, 1e-6, _lib.BF16, None)  # hidden == 0
    assert rc == -1
    assert "rmsnorm" in _lib.last_error()
    rc = lib.semipd_decode_attention(None, None, None, None, None, None, None, 1, 6, 4, 64, 64, 0, 0, 0, 0, 1,
                                     1.0, 0.0, _lib.BF16, _lib.BF16, None)  # Hq % Hkv != 0
    assert rc == -3
    with pytest.raises(RuntimeError):
        _lib.check(rc, "decode_attention")


def test_no_cpu_fallback():
    from semi_pd_amd import ops
    x = torch.randn(2, 64)
    with pytest.raises(RuntimeError):
        ops.rmsnorm(x, torch.ones(64))
    with pytest.raises(RuntimeError):
        _lib.dtype_code(torch.float64)


def test_cu_mask_fill_is_balanced():
    import ctypes as C
    lib = _lib.load()
    words = 8
    lo = (C.c_uint32 * words)()
    hi = (C.c_uint32 * words)()
    n_lo = lib.semipd_cu_mask_fill(256, 50, 0, C.addressof(lo), words)
    n_hi = lib.semipd_cu_mask_fill(256, 50, 1, C.addressof(hi), words)
    assert n_lo == 128 and n_hi == 128
    bits_lo = [i for i in range(256) if lo[i >> 5] >> (i & 31) & 1]
    bits_hi = [i for i in range(256) if hi[i >> 5] >> (i & 31) & 1]
    assert not set(bits_lo) & set(bits_hi) and len(set(bits_lo) | set(bits_hi)) == 256
    for xcd in range(8):  # logical CU i lands on XCD i % 8
        assert sum(1 for i in bits_lo if i % 8 == xcd) == 16
        assert sum(1 for i in bits_hi if i % 8 == xcd) == 16
    n80 = lib.semipd_cu_mask_fill(256, 80, 0, C.addressof(lo), words)
    assert n80 == 192 and n80 % 32 == 0    # whole groups of 32: one CU per shader engine of every XCD (csrc/ipc.hip)


def test_integration_appendix_names_every_entry_point():
    """INTEGRATION.md's appendix (tools/abi_table.py --write) is generated from the header: it must be current and
    name every declared symbol with the reference interface its declaration cites."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("abi_table", os.path.join(root, "tools", "abi_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    assert mod.BEGIN in text and mod.END in text
    block = text[text.index(mod.BEGIN) + len(mod.BEGIN):text.index(mod.END)].strip()
    assert block == mod.table(), "run `python tools/abi_table.py --write`"
    names = [n for n, _, _ in mod.entries()]
    assert len(names) == len(set(names)) >= 58
    for n in names:
        assert f"| `{n}` |" in block


def test_ipc_open_watchdog_ends_a_stuck_importer_with_a_reason():
    """hipIpcOpenMemHandle can hang for allocation sizes the measured rule does not cover (csrc/ipc.hip): the importer
    bounds every open; a process stuck past the limit prints what it was doing and exits with code 71."""
    import subprocess
    import sys
    code = ("import sys, time; sys.path[:0] = %r; import semi_pd_ipc; "
            "t = semi_pd_ipc._open_watchdog([7] * 64, 0.3); t.start(); time.sleep(5); print('not reached')"
            % ([PKG],))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 71 and "did not return within" in r.stderr and "not reached" not in r.stdout


def test_reference_all_reduce_op_names_are_exported():
    """The ROCm custom all-reduce op set of the reference (sgl-kernel/csrc/torch_extension_rocm.cc:25-55, wrappers in
    sgl-kernel/python/sgl_kernel/allreduce.py:5-52): every name, with the wrapper's parameter names in order."""
    import inspect
    from semi_pd_amd import sgl_kernel_allreduce as ops
    want = {
        "init_custom_ar": ["meta", "rank_data", "handles", "offsets", "rank", "full_nvlink"],
        "all_reduce_reg": ["fa", "inp", "out"],
        "all_reduce_unreg": ["fa", "inp", "reg_buffer", "out"],
        "dispose": ["fa"],
        "meta_size": [],
        "register_buffer": ["fa", "t", "handles", "offsets"],
        "get_graph_buffer_ipc_meta": ["fa"],
        "register_graph_buffers": ["fa", "handles", "offsets"],
        "allocate_meta_buffer": ["size"],
        "get_meta_buffer_ipc_handle": ["inp"],
    }
    for name, params in want.items():
        assert list(inspect.signature(getattr(ops, name)).parameters) == params, name
    assert ops.meta_size() > 0 and ops.meta_size() % 256 == 0
