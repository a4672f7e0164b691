"""torch.ops.sgl_kernel.<op> under the reference's schemas (sgl-kernel/csrc/torch_extension.cc:49-175): every registered op,
called the way the reference's Python wrappers call it (sgl-kernel/python/sgl_kernel/elementwise.py, moe.py, gemm.py,
sampling.py), against the same op through semi_pd_amd.ops (which the other GPU tests hold to the oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from semi_pd_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def sgl(device):
    from semi_pd_amd import sgl_kernel_torch_ops
    names = sgl_kernel_torch_ops.register()
    assert len(names) == 21 and all(hasattr(torch.ops.sgl_kernel, n) for n in names)
    return torch.ops.sgl_kernel


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_elementwise_ops(sgl, ops, device):
    torch.manual_seed(0)
    x = torch.randn(37, 4096, device=device).to(torch.bfloat16)
    w = torch.randn(4096, device=device).to(torch.bfloat16)
    out = torch.empty_like(x)
    sgl.rmsnorm(out, x, w, 1e-6, _stream())                      # elementwise.py:9-18
    assert torch.equal(out, ops.rmsnorm(x, w, 1e-6))
    side = torch.cuda.Stream()                                    # an explicit stream is honoured
    side.wait_stream(torch.cuda.current_stream())
    out2 = torch.empty_like(x)
    sgl.rmsnorm(out2, x, w, 1e-6, side.cuda_stream)
    side.synchronize()
    assert torch.equal(out2, out)
    a, r = x.clone(), torch.randn_like(x)
    a2, r2 = a.clone(), r.clone()
    sgl.fused_add_rmsnorm(a, r, w, 1e-6)                          # elementwise.py:21-24
    ops.fused_add_rmsnorm(a2, r2, w, 1e-6)
    assert torch.equal(a, a2) and torch.equal(r, r2)
    g = torch.randn(37, 2 * 1408, device=device).to(torch.bfloat16)
    o = torch.empty(37, 1408, dtype=torch.bfloat16, device=device)
    sgl.silu_and_mul(o, g, _stream())                             # elementwise.py:61-75
    assert torch.equal(o, ops.silu_and_mul(g))


@pytest.mark.parametrize("neox", [True, False])
def test_rope_op(sgl, ops, device, neox):
    torch.manual_seed(1)
    nnz, Hq, Hk, D = 19, 8, 2, 128
    q = torch.randn(nnz, Hq * D, device=device).to(torch.bfloat16)
    k = torch.randn(nnz, Hk * D, device=device).to(torch.bfloat16)
    pos = torch.randint(0, 2000, (nnz,), device=device, dtype=torch.int64)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j -> ij", torch.arange(2048, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(device)
    q1, k1 = q.clone(), k.clone()
    ops.apply_rope_with_cos_sin_cache_inplace(pos, q1, k1, D, cache, is_neox=neox)
    q2, k2 = q.clone(), k.clone()
    # elementwise.py:142-151: in place, q / k as [nnz, heads, head_size] views, interleave = not is_neox
    sgl.apply_rope_pos_ids_cos_sin_cache(q2.view(nnz, -1, D), k2.view(nnz, -1, D), q2.view(nnz, -1, D), k2.view(nnz, -1, D),
                                         cache, pos.long(), not neox, _stream())
    assert torch.equal(q1, q2) and torch.equal(k1, k2)


def test_moe_align_and_bmm_fp8_ops(sgl, ops, device):
    torch.manual_seed(2)
    T, k, E, block = 100, 6, 64, 64
    ids = torch.randint(0, E, (T, k), device=device, dtype=torch.int32)
    n = T * k + E * (block - 1)

    def bufs():
        return (torch.empty(n, dtype=torch.int32, device=device), torch.empty((n + block - 1) // block, dtype=torch.int32, device=device),
                torch.empty(1, dtype=torch.int32, device=device), torch.zeros((E + 1) * E, dtype=torch.int32, device=device),
                torch.zeros(E + 1, dtype=torch.int32, device=device))

    s1, e1, p1, c1, cs1 = bufs()
    sgl.moe_align_block_size(ids, E, block, s1, e1, p1, c1, cs1)     # moe.py:4-23
    s2, e2, p2, _, cs2 = bufs()
    ops.moe_align_block_size(ids, E, block, s2, e2, p2, None, cs2)
    npp = int(p1)
    assert npp == int(p2) and torch.equal(e1[:npp // block], e2[:npp // block])
    for b in range(npp // block):   # the order inside an expert's rows is not defined (atomics): compare as sets
        assert sorted(s1[b * block:(b + 1) * block].tolist()) == sorted(s2[b * block:(b + 1) * block].tolist())
    a = torch.randn(4, 9, 128, device=device).to(torch.bfloat16)
    b = torch.randn(4, 512, 128, device=device).to(torch.bfloat16)
    a8, a_s = ops.input_to_float8(a)
    b8, b_s = ops.input_to_float8(b)
    d = torch.empty(4, 9, 512, dtype=torch.bfloat16, device=device)
    ws = torch.empty(1024, dtype=torch.uint8, device=device)
    sgl.bmm_fp8(a8, b8.transpose(1, 2), d, a_s, b_s, ws, 0, _stream())   # gemm.py:49-63
    assert torch.equal(d, ops.bmm_fp8(a8, b8.transpose(1, 2), a_s, b_s, torch.bfloat16))


def test_sampling_ops(sgl, ops, device):
    torch.manual_seed(3)
    B, V = 11, 5000
    probs = torch.softmax(torch.randn(B, V, device=device) * 3, dim=-1)
    u = torch.rand(32, B, device=device)
    samples = torch.empty(B, dtype=torch.int32, device=device)
    success = torch.empty(B, dtype=torch.bool, device=device)
    top_k = torch.randint(1, 50, (B,), device=device, dtype=torch.int32)
    top_p = torch.rand(B, device=device) * 0.5 + 0.4
    sgl.top_k_top_p_sampling_from_probs(probs, u, samples, success, top_k, 0, top_p, 0.0, True, _stream())   # sampling.py:139-165
    want, ok = ops.top_k_top_p_sampling_from_probs(probs, u, top_k, top_p)
    assert torch.equal(samples, want) and torch.equal(success, ok) and bool(success.all())
    sgl.top_p_sampling_from_probs(probs, u, samples, success, None, 0.8, True, _stream())                     # sampling.py:100-136
    want, ok = ops.top_k_top_p_sampling_from_probs(probs, u, V, 0.8)
    assert torch.equal(samples, want)
    sgl.min_p_sampling_from_probs(probs, u, samples, None, 0.05, True, _stream())                            # sampling.py:194-210
    assert torch.equal(samples, ops.min_p_sampling_from_probs(probs, u, 0.05))
    rp = torch.empty_like(probs)
    sgl.top_k_renorm_probs_wrapper(probs, rp, top_k, 0, _stream())                                            # sampling.py:25-32
    assert torch.equal(rp, ops.top_k_renorm_prob(probs, top_k))
    sgl.top_p_renorm_probs(probs, rp, None, 0.7, _stream())                                                   # sampling.py:53-60
    assert torch.equal(rp, ops.top_p_renorm_prob(probs, 0.7))


def test_all_reduce_ops_are_registered(sgl, device):
    """The ten ROCm all-reduce ops dispatch (the reductions themselves run in tests/test_gpu_all_reduce.py)."""
    assert sgl.meta_size() > 0
    meta = sgl.allocate_meta_buffer(sgl.meta_size() + (1 << 20))
    try:
        h = sgl.get_meta_buffer_ipc_handle(meta)
        assert h.dtype == torch.uint8 and h.numel() == 64 and not h.is_cuda and int(np.count_nonzero(h.numpy())) > 0
    finally:
        from semi_pd_amd import sgl_kernel_allreduce
        sgl_kernel_allreduce.free_meta_buffer(meta)
