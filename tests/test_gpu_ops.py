"""GPU parity: the HIP kernels (through the C-ABI, via semi_pd_amd.ops) against the CPU oracle on the
same seeded inputs, and against the golden vectors produced by the reference.  Shapes and
tolerances follow the reference's own kernel tests (cited per test)."""
import numpy as np
import pytest
import torch

from conftest import DTYPES, from_bits, load_golden
from oracle import ops as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: dict(rtol=1e-5, atol=1e-5), torch.float16: dict(rtol=1e-3, atol=1e-3),
       torch.bfloat16: dict(rtol=1.6e-2, atol=1e-2)}


@pytest.fixture(scope="module")
def ops():
    from semi_pd_amd import ops as _ops
    return _ops


def _close(got, want, dtype, **kw):
    tol = dict(TOL[dtype])
    tol.update(kw)
    torch.testing.assert_close(got.detach().cpu().float(), want.detach().cpu().float(), **tol)


# ----------------------------------------------------------------------------- RMSNorm
# sgl-kernel/tests/test_norm.py:52-129 (batch 1/19/99/989, hidden 111..16384, tol 1e-3 fp16)
@pytest.mark.parametrize("batch", [1, 19, 99, 989])
@pytest.mark.parametrize("hidden", [111, 500, 1024, 3072, 3584, 4096, 8192, 16384])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rmsnorm(ops, device, batch, hidden, dtype):
    torch.manual_seed(batch * 7 + hidden)
    x = torch.randn(batch, hidden).to(dtype)
    w = torch.randn(hidden).to(dtype)
    y = ops.rmsnorm(x.to(device), w.to(device), 1e-6)
    _close(y, O.rms_norm(x, w, 1e-6), dtype)
    out = torch.empty_like(x, device=device)
    ops.rmsnorm(x.to(device), w.to(device), 1e-6, out=out)
    _close(out, O.rms_norm(x, w, 1e-6), dtype)


@pytest.mark.parametrize("batch", [1, 19, 99, 989])
@pytest.mark.parametrize("hidden", [111, 500, 1024, 3072, 4096, 7168, 8192, 16384])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_fused_add_rmsnorm(ops, device, batch, hidden, dtype):
    torch.manual_seed(batch * 3 + hidden)
    x = torch.randn(batch, hidden).to(dtype)
    r = torch.randn(batch, hidden).to(dtype)
    w = torch.randn(hidden).to(dtype)
    xd, rd = x.to(device), r.to(device)
    ops.fused_add_rmsnorm(xd, rd, w.to(device), 1e-6)
    y, r2 = O.fused_add_rms_norm(x, r, w, 1e-6)
    _close(xd, y, dtype)
    _close(rd, r2, dtype)


def test_rmsnorm_golden_bitwise(ops, device):
    """Golden vectors come from the reference's RMSNorm.forward_native (layers/layernorm.py:59-76);
    the kernel reproduces them bit for bit on these cases."""
    g = load_golden("rmsnorm")
    eps = float(g["eps"])
    for ci in range(int(g["n"])):
        tag = f"c{ci}_"
        dt = DTYPES[str(g[tag + "dtype"])]
        x, r, w = (from_bits(g[tag + k], dt) for k in ("x", "r", "w"))
        y = ops.rmsnorm(x.to(device), w.to(device), eps).cpu()
        _close(y, from_bits(g[tag + "y"], dt), dt)
        if dt != torch.float32:  # 16-bit outputs: the fp32 reduction order is invisible after rounding
            mism = (y.view(torch.int16) != from_bits(g[tag + "y"], dt).view(torch.int16)).float().mean()
            assert mism < 0.02, f"case {ci}: {mism:.3f} of elements differ by an ulp"
        xd, rd = x.to(device), r.to(device)
        ops.fused_add_rmsnorm(xd, rd, w.to(device), eps)
        assert torch.equal(rd.cpu(), from_bits(g[tag + "r_fused"], dt))  # the sum is exactly rounded
        _close(xd, from_bits(g[tag + "y_fused"], dt), dt)


def test_rmsnorm_strided_rows(ops, device):
    x = torch.randn(16, 3 * 256).to(torch.bfloat16)
    w = torch.randn(256).to(torch.bfloat16)
    xs = x.to(device)[:, 256:512]  # row stride 768
    y = ops.rmsnorm(xs, w.to(device), 1e-5)
    _close(y, O.rms_norm(x[:, 256:512], w, 1e-5), torch.bfloat16)


# ----------------------------------------------------------------------------- SiLU*mul
# sgl-kernel/tests/test_activation.py:8-15 (dim 128..16384, batch 1..16, seq 1..512, tol 1e-3)
@pytest.mark.parametrize("dim", [128, 256, 1408, 14336])
@pytest.mark.parametrize("rows", [1, 7, 512])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_silu_and_mul(ops, device, dim, rows, dtype):
    torch.manual_seed(dim + rows)
    x = torch.randn(rows, 2 * dim).to(dtype)
    y = ops.silu_and_mul(x.to(device))
    _close(y, O.silu_and_mul(x), dtype)
    x3 = torch.randn(2, rows, 2 * dim).to(dtype)
    _close(ops.silu_and_mul(x3.to(device)), O.silu_and_mul(x3), dtype)


# ----------------------------------------------------------------------------- RoPE
# sgl-kernel/tests/test_rotary_embedding.py:143-196: (head, rot, max_pos, base, neox, dtype, batch, seq, Hq, Hkv)
ROPE_CASES = [
    (64, 64, 32, 8000, True, torch.bfloat16, 32, 32, 1, 1),
    (256, 128, 4096, 10000, True, torch.bfloat16, 2, 512, 4, 2),
    (512, 128, 311, 10000, True, torch.bfloat16, 3, 39, 4, 2),
    (128, 128, 2048, 10000, False, torch.bfloat16, 2, 512, 32, 8),
    (128, 128, 2048, 10000, False, torch.bfloat16, 2, 512, 16, 4),
    (512, 128, 311, 10000, False, torch.bfloat16, 3, 39, 4, 2),
    (64, 20, 311, 10000, True, torch.float16, 3, 5, 4, 2),  # rot not a multiple of 16: scalar path
]


@pytest.mark.parametrize("head,rot,max_pos,base,neox,dtype,batch,seq,Hq,Hk", ROPE_CASES)
def test_rope(ops, device, head, rot, max_pos, base, neox, dtype, batch, seq, Hq, Hk):
    torch.manual_seed(head + rot + seq)
    cache = O.cos_sin_cache_from_inv_freq(O.rope_inv_freq(rot, base), max_pos)
    T = batch * seq
    pos = torch.randint(0, max_pos, (T,))
    q = torch.randn(T, Hq * head).to(dtype)
    k = torch.randn(T, Hk * head).to(dtype)
    qd, kd = q.to(device), k.to(device)
    ops.apply_rope_with_cos_sin_cache_inplace(pos.to(device), qd, kd, head, cache.to(device), neox)
    qo, ko = O.apply_rope(pos, q, k, head, cache, neox)
    _close(qd, qo, dtype, rtol=1e-2, atol=1e-2)
    _close(kd, ko, dtype, rtol=1e-2, atol=1e-2)


def test_rope_golden(ops, device):
    g = load_golden("rope")
    for name in g["names"]:
        name = str(name)
        head, neox = int(g[name + "_head"]), bool(g[name + "_neox"])
        cache = torch.from_numpy(g[name + "_cache"])
        pos = torch.from_numpy(g[name + "_pos"])
        q, k = torch.from_numpy(g[name + "_q"]).to(device), torch.from_numpy(g[name + "_k"]).to(device)
        ops.apply_rope_with_cos_sin_cache_inplace(pos.to(device), q, k, head, cache.to(device), neox)
        _close(q, torch.from_numpy(g[name + "_qo"]), torch.float32, rtol=1e-5, atol=1e-5)
        _close(k, torch.from_numpy(g[name + "_ko"]), torch.float32, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("Hq,Hk,head,rot", [(32, 8, 128, 128), (12, 12, 64, 64), (4, 2, 64, 32)])
def test_rope_kv_store_fused(ops, device, neox, Hq, Hk, head, rot):
    torch.manual_seed(Hq + head)
    dtype = torch.bfloat16
    T, N = 37, 101
    cache = O.cos_sin_cache_from_inv_freq(O.rope_inv_freq(rot, 10000), 512)
    pos = torch.randint(0, 512, (T,))
    loc = (torch.randperm(N - 1)[:T] + 1).long()
    q = torch.randn(T, Hq * head).to(dtype)
    k = torch.randn(T, Hk * head).to(dtype)
    v = torch.randn(T, Hk * head).to(dtype)
    kbuf = torch.zeros(N, Hk, head, dtype=dtype, device=device)
    vbuf = torch.zeros(N, Hk, head, dtype=dtype, device=device)
    qd, kd = q.to(device), k.to(device)
    ops.rope_and_store_kv(pos.to(device), qd, kd, v.to(device), head, cache.to(device), neox, kbuf, vbuf,
                          loc.to(device))
    qo, ko = O.apply_rope(pos, q, k, head, cache, neox)
    _close(qd, qo, dtype, rtol=1e-2, atol=1e-2)
    _close(kd, ko, dtype, rtol=1e-2, atol=1e-2)
    assert torch.equal(kbuf.cpu()[loc].reshape(T, -1), kd.cpu())  # pool holds exactly the rotated keys
    assert torch.equal(vbuf.cpu()[loc].reshape(T, -1), v)
    untouched = torch.ones(N, dtype=torch.bool)
    untouched[loc] = False
    assert kbuf.cpu()[untouched].abs().sum() == 0 and vbuf.cpu()[untouched].abs().sum() == 0


def test_store_kv_rows_and_gather(ops, device):
    dtype = torch.bfloat16
    src = torch.randn(50, 8, 128).to(dtype)
    loc = (torch.randperm(199)[:50] + 1).long()
    buf = torch.zeros(200, 8, 128, dtype=dtype, device=device)
    ops.store_kv_rows(buf, loc.to(device), src.to(device))
    ref = torch.zeros(200, 8, 128, dtype=dtype)
    ref[loc] = src
    assert torch.equal(buf.cpu(), ref)
    # MLA latent rows: 576 elements, 1152 B rows
    lat = torch.randn(9, 1, 576).to(dtype)
    buf2 = torch.zeros(32, 1, 576, dtype=dtype, device=device)
    ops.store_kv_rows(buf2, torch.arange(1, 10).to(device), lat.to(device))
    assert torch.equal(buf2.cpu()[1:10], lat)
    hidden = torch.randn(100, 768).to(dtype)
    idx = torch.tensor([3, 99, 0, 50])
    assert torch.equal(ops.gather_rows(hidden.to(device), idx.to(device)).cpu(), hidden[idx])


# ----------------------------------------------------------------------------- kv indices / positions
def test_kv_indices_golden_and_random(ops, device):
    g = load_golden("kv_indices")
    r2t = torch.from_numpy(g["req_to_token"]).to(device)
    rpi = torch.from_numpy(g["req_pool_indices"]).to(device)
    for tag, start in (("nostart", None), ("start", torch.from_numpy(g["start"]).to(device))):
        lens = torch.from_numpy(g[tag + "_lens"]).to(device)
        indptr = torch.full((5,), -1, dtype=torch.int32, device=device)
        ind = torch.full((int(g[tag + "_indptr"][-1]),), -1, dtype=torch.int32, device=device)
        ops.create_flashinfer_kv_indices(r2t, rpi, lens, indptr, start, ind)
        assert np.array_equal(indptr.cpu().numpy(), g[tag + "_indptr"])
        assert np.array_equal(ind.cpu().numpy(), g[tag + "_indices"])
    # test/srt/test_create_kvindices.py style: random batch against the python loop
    torch.manual_seed(0)
    B, C = 257, 700
    r2t = torch.randint(0, 10000, (300, C), dtype=torch.int32)
    rpi = torch.randperm(300)[:B].long()
    lens = torch.randint(1, C, (B,), dtype=torch.int32)
    want_ptr, want = O.create_kv_indices(r2t, rpi, lens)
    indptr = torch.empty(B + 1, dtype=torch.int32, device=device)
    ind = torch.empty(int(want_ptr[-1]), dtype=torch.int32, device=device)
    ops.create_flashinfer_kv_indices(r2t.to(device), rpi.to(device), lens.to(device), indptr, None, ind)
    assert torch.equal(indptr.cpu(), want_ptr) and torch.equal(ind.cpu(), want)


def test_compute_position(ops, device):
    pre = torch.tensor([0, 5, 100, 7], dtype=torch.int32)
    ext = torch.tensor([3, 1, 300, 64], dtype=torch.int32)
    pos, start = ops.compute_position(pre.to(device), ext.to(device), int(ext.sum()))
    wp, ws = O.compute_position(pre, ext)
    assert torch.equal(pos.cpu(), wp) and torch.equal(start.cpu(), ws)


# ----------------------------------------------------------------------------- decode attention
def _paged(B, lens, Hkv, Dk, Dv, dtype, seed, extra=11):
    g = torch.Generator().manual_seed(seed)
    total = int(sum(lens))
    N = total + extra
    perm = torch.randperm(N - 1, generator=g)[:total] + 1
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    k_buf = torch.randn(N, Hkv, Dk, generator=g).to(dtype)
    v_buf = torch.randn(N, Hkv, Dv, generator=g).to(dtype)
    return k_buf, v_buf, kv_indptr, perm.to(torch.int32)


DECODE_CASES = [
    # B, lens, Hq, Hkv, Dk, Dv, splits, cap
    (3, [5, 33, 700], 32, 8, 128, 128, 8, 0.0),      # Llama-3-8B heads
    (2, [1, 257], 8, 1, 128, 128, 16, 0.0),          # 70B/TP8: g = 8
    (4, [64, 65, 127, 1], 12, 12, 64, 64, 4, 0.0),   # OPT-125m: MHA D=64
    (2, [300, 17], 16, 2, 128, 128, 1, 0.0),         # g = 8, single split writes o directly
    (2, [90, 31], 6, 2, 96, 96, 3, 30.0),            # g = 3, D = 96, logit cap
    (2, [40, 9], 4, 4, 80, 80, 2, 0.0),              # D = 80
    (3, [50, 3, 128], 32, 2, 64, 64, 5, 0.0),        # g = 16: two head tiles per kv head
    (2, [33, 70], 16, 1, 576, 512, 4, 0.0),          # MLA latent (generic path)
    (3, [300, 1, 65], 40, 1, 576, 512, 1, 0.0),      # MLA, 3 head tiles (last partial), single split
    (2, [129, 1000], 128, 1, 576, 512, 8, 20.0),     # MLA, DeepSeek-V3 head count, logit cap
    (20, [129, 1000, 7, 64, 65, 300, 31, 32, 33, 512, 1, 96, 257, 40, 700, 2, 95, 128, 160, 511], 128, 1, 576, 512, 8, 0.0),   # MLA, 128 heads: enough
                                                     # workgroups for the shared-tile kernel (mla_decode_shared.hip)
    (2, [12, 30], 3, 1, 13, 13, 2, 0.0),             # odd head dim (generic path)
    (1, [2048], 32, 8, 128, 128, 16, 0.0),
]


@pytest.mark.parametrize("B,lens,Hq,Hkv,Dk,Dv,splits,cap", DECODE_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_attention(ops, device, B, lens, Hq, Hkv, Dk, Dv, splits, cap, dtype):
    k_buf, v_buf, indptr, indices = _paged(B, lens, Hkv, Dk, Dv, dtype, seed=Hq * 31 + Dk)
    if Dk == 576:  # MLA: V is the first 512 columns of the latent K rows (memory_pool.py:430-437)
        v_buf = k_buf[..., :Dv]
    torch.manual_seed(B + Hq)
    q = torch.randn(B, Hq, Dk).to(dtype)
    sm_scale = 1.0 / (Dk ** 0.5)
    o = torch.empty(B, Hq, Dv, dtype=dtype, device=device)
    logits = torch.empty(B, Hq, splits, Dv + 1, dtype=torch.float32, device=device)
    kd = k_buf.to(device)
    vd = kd[..., :Dv] if Dk == 576 else v_buf.to(device)
    ops.decode_attention_fwd(q.to(device), kd, vd, o, indptr.to(device), indices.to(device), logits, splits,
                             sm_scale, cap)
    want = O.decode_attention(q, k_buf, v_buf, indptr, indices, sm_scale, cap)
    # reference bar: cos-sim > 0.99, atol 3e-2 (test_triton_attention_kernels.py:342-347); ours is tighter
    _close(o, want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)


MLA_SHARED_CASES = [
    # B, lens, Hq, splits  -- forced onto mla_decode_shared.hip whatever the workgroup count
    (2, [129, 1000], 128, 8),
    (3, [1, 31, 97], 128, 1),             # single split, rows short of one tile
    (4, [5, 2048, 32, 33], 128, 16),      # more splits than rows (empty splits), tile boundaries
    (2, [64, 333], 64, 3),                # 64 heads: two waves per workgroup
    (1, [700], 256, 4),                   # two head groups
    (2, [3000, 95], 128, 2),              # ~47 tiles per split: the ring wraps many times
    (33, [40 + 7 * i for i in range(33)], 128, 3),   # 4224 (request, head) pairs: the one-wave-per-head stage 2
]


@pytest.mark.parametrize("B,lens,Hq,splits", MLA_SHARED_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mla_decode_shared_tile_kernel(ops, device, monkeypatch, B, lens, Hq, splits, dtype):
    """MLA decode with 64 / 128 / 256 heads on one latent row: the kernel that shares a latent tile between 128 heads
    (mla_decode_shared.hip; SEMIPD_MLA_SHARED=2 takes it whatever the size) against the oracle, and against the wide
    kernel it replaces (SEMIPD_MLA_SHARED=0) on the same inputs."""
    k_buf, _, indptr, indices = _paged(B, lens, 1, 576, 512, dtype, seed=Hq * 7 + splits)
    torch.manual_seed(B * 13 + Hq)
    q = torch.randn(B, Hq, 576).to(dtype)
    sm_scale = 576 ** -0.5
    kd = k_buf.to(device)
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("SEMIPD_MLA_SHARED", mode)
        o = torch.full((B, Hq, 512), float("nan"), dtype=dtype, device=device)
        logits = torch.empty(B, Hq, splits, 513, dtype=torch.float32, device=device)
        ops.decode_attention_fwd(q.to(device), kd, kd[..., :512], o, indptr.to(device), indices.to(device), logits, splits,
                                 sm_scale, 0.0)
        outs[mode] = o.float().cpu()
    want = O.decode_attention(q, k_buf, k_buf[..., :512], indptr, indices, sm_scale, 0.0)
    _close(outs["2"].to(dtype), want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)
    assert torch.isfinite(outs["2"]).all()
    assert (outs["2"] - outs["0"]).abs().max() < (4e-3 if dtype == torch.float16 else 2e-2)


@pytest.mark.parametrize("seed", range(12))
def test_mla_decode_shared_tile_kernel_random_batches(ops, device, monkeypatch, seed):
    """Random batches (1-6 requests of 1-1500 rows, 1-16 splits, 64 / 128 heads): shared-tile kernel == oracle and ~= the wide
    kernel; empty splits, one-row requests and splits shorter than a tile come up by themselves."""
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 7))
    lens = [int(x) for x in rng.integers(1, 1500, size=B)]
    if seed % 3 == 0:
        lens[0] = int(rng.integers(1, 40))
    Hq = 128 if seed % 2 == 0 else 64
    splits = int(rng.integers(1, 17))
    dtype = torch.bfloat16
    k_buf, _, indptr, indices = _paged(B, lens, 1, 576, 512, dtype, seed=seed)
    torch.manual_seed(seed)
    q = (torch.randn(B, Hq, 576) * float(rng.uniform(0.3, 2.0))).to(dtype)
    sm_scale = 576 ** -0.5
    kd = k_buf.to(device)
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("SEMIPD_MLA_SHARED", mode)
        o = torch.full((B, Hq, 512), float("nan"), dtype=dtype, device=device)
        logits = torch.empty(B, Hq, splits, 513, dtype=torch.float32, device=device)
        ops.decode_attention_fwd(q.to(device), kd, kd[..., :512], o, indptr.to(device), indices.to(device), logits, splits,
                                 sm_scale, 0.0)
        outs[mode] = o.float().cpu()
    want = O.decode_attention(q, k_buf, k_buf[..., :512], indptr, indices, sm_scale, 0.0)
    _close(outs["2"].to(dtype), want, dtype, rtol=2e-2, atol=1.5e-2)
    assert (outs["2"] - outs["0"]).abs().max() < 2e-2


@pytest.mark.parametrize("fixture,mla_shared", [("decode_attention", ""), ("decode_attention_8c", ""),
                                                ("decode_attention_8c", "2")])
def test_decode_attention_golden(ops, device, monkeypatch, fixture, mla_shared):
    """fp32 golden vectors from the reference Triton kernel, evaluated here in bf16 (8c: the SURVEY 8(c) shapes,
    MLA 576 / 512 with 16 and 128 heads, D 80 / 13, group 16, 16 splits).  mla_shared = "2": the 128-head MLA vector
    (one request: too few workgroups for the size rule) through mla_decode_shared.hip as well."""
    if mla_shared:
        monkeypatch.setenv("SEMIPD_MLA_SHARED", mla_shared)
    g = load_golden(fixture)
    for name in g["names"]:
        name = str(name)
        q, k = (torch.from_numpy(g[f"{name}_{x}"]).to(torch.bfloat16) for x in ("q", "k"))
        v = (k[..., :int(g[name + "_meta"][3])] if name.startswith("mla_576")
             else torch.from_numpy(g[name + "_v"]).to(torch.bfloat16))
        indptr, indices = torch.from_numpy(g[name + "_indptr"]), torch.from_numpy(g[name + "_indices"])
        splits, sm_scale, cap = g[name + "_meta"][:3]
        B, Hq, _ = q.shape
        Dv = v.shape[2]
        o = torch.empty(B, Hq, Dv, dtype=torch.bfloat16, device=device)
        logits = torch.empty(B, Hq, int(splits), Dv + 1, dtype=torch.float32, device=device)
        kd = k.to(device)
        vd = kd[..., :Dv] if name.startswith("mla_576") else v.to(device)
        ops.decode_attention_fwd(q.to(device), kd, vd, o, indptr.to(device),
                                 indices.to(device), logits, int(splits), float(sm_scale), float(cap))
        _close(o, torch.from_numpy(g[name + "_o"]), torch.bfloat16, rtol=3e-2, atol=3e-2)


def test_decode_attention_ragged_batch_and_padding(ops, device):
    """B smaller than the scratch batch (CUDA-graph padding, decode_attention.py:638-640)."""
    dtype = torch.bfloat16
    B, Hq, Hkv, D, splits = 5, 8, 2, 128, 4
    lens = [1, 2, 3, 1000, 64]
    k_buf, v_buf, indptr, indices = _paged(B, lens, Hkv, D, D, dtype, seed=5)
    q = torch.randn(B, Hq, D).to(dtype)
    o = torch.empty(B, Hq, D, dtype=dtype, device=device)
    logits = torch.empty(16, Hq, splits, D + 1, dtype=torch.float32, device=device)
    indptr_pad = torch.cat([indptr, indptr[-1:].repeat(11)])
    ops.decode_attention_fwd(q.to(device), k_buf.to(device), v_buf.to(device), o, indptr_pad.to(device),
                             indices.to(device), logits, splits, 0.088, 0.0)
    _close(o, O.decode_attention(q, k_buf, v_buf, indptr, indices, 0.088), dtype, rtol=2e-2, atol=1.5e-2)


# ----------------------------------------------------------------------------- extend attention
EXTEND_CASES = [
    # prefix lens, extend lens, Hq, Hkv, Dk, Dv, cap
    ([0, 13, 40], [20, 7, 170], 8, 2, 128, 128, 0.0),
    ([0, 0], [5, 260], 12, 12, 64, 64, 0.0),
    ([300], [129], 8, 1, 128, 128, 0.0),
    ([9, 64, 0], [33, 64, 1], 4, 1, 128, 128, 25.0),
    ([6, 0], [10, 3], 2, 1, 96, 64, 0.0),
    ([17], [70], 4, 2, 80, 80, 0.0),
    ([5, 0], [40, 9], 4, 2, 192, 128, 0.0),      # DeepSeek MHA prefill: qk 128+64, v 128
    ([3], [20], 3, 3, 13, 13, 0.0),              # test_triton_attention_kernels.py D=13
    ([11, 2], [6, 14], 4, 1, 576, 512, 0.0),     # MLA absorbed with prefix (generic path)
]


@pytest.mark.parametrize("pre,ext,Hq,Hkv,Dk,Dv,cap", EXTEND_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_extend_attention(ops, device, pre, ext, Hq, Hkv, Dk, Dv, cap, dtype):
    B = len(pre)
    k_buf, v_buf, kv_indptr, kv_indices = _paged(B, pre, Hkv, Dk, Dv, dtype, seed=Hq + Dk)
    T = sum(ext)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
    torch.manual_seed(T)
    q = torch.randn(T, Hq, Dk).to(dtype)
    k = torch.randn(T, Hkv, Dk).to(dtype)
    v = torch.randn(T, Hkv, Dv).to(dtype)
    o = torch.empty(T, Hq, Dv, dtype=dtype, device=device)
    sm_scale = 1.0 / (Dk ** 0.5)
    ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, k_buf.to(device), v_buf.to(device),
                             qo_indptr.to(device), kv_indptr.to(device), kv_indices.to(device), None, None,
                             max(ext), sm_scale, cap)
    want = O.extend_attention(q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, sm_scale, cap)
    # reference bar: allclose(rtol=1e-2) bf16 vs redundant attention (test_triton_attention_kernels.py:168)
    _close(o, want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)


# Head size 128 / 128 without a cap runs the shared-KV kernel (extend_attention_shared_kv.hip): one workgroup = G q heads
# of a kv head x 4 / G blocks of 32 tokens x 2 halves of every 128-row KV tile.  Cases around every seam of it: group
# sizes that give G = 4 / 2 / 1 (and two workgroups per kv head), tile and half boundaries of the prefix and of the new
# tokens, token blocks and whole tiles past the end of short sequences next to a long one, one-token requests.
SHARED_KV_CASES = [
    # prefix lens, extend lens, Hq, Hkv
    ([0], [1], 4, 1),
    ([0, 0, 0], [31, 32, 33], 8, 2),
    ([0], [64], 8, 2), ([0], [65], 8, 2), ([0], [127], 4, 1), ([0], [128], 4, 1), ([0], [129], 4, 1),
    ([63, 64, 65], [3, 2, 1], 4, 1),
    ([127, 128, 129, 1], [40, 1, 70, 200], 8, 2),
    ([300, 0, 17], [257, 5, 130], 16, 2),          # group 8: two workgroups per kv head
    ([70, 0], [100, 300], 6, 3),                   # group 2: two token blocks per workgroup
    ([70, 0, 130], [100, 300, 2], 3, 3),           # group 1 (MHA): four token blocks per workgroup
    ([5], [97], 6, 2),                             # group 3: G = 1
    ([513], [1000], 32, 8),                        # Llama-3-8B heads, a 1000-token prompt behind a 513-token prefix
]


@pytest.mark.parametrize("pre,ext,Hq,Hkv", SHARED_KV_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_extend_attention_shared_kv(ops, device, pre, ext, Hq, Hkv, dtype):
    D = 128
    B = len(pre)
    k_buf, v_buf, kv_indptr, kv_indices = _paged(B, pre, Hkv, D, D, dtype, seed=Hq + sum(ext))
    T = sum(ext)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
    torch.manual_seed(T + Hq)
    q = torch.randn(T, Hq, D).to(dtype)
    k = torch.randn(T, Hkv, D).to(dtype)
    v = torch.randn(T, Hkv, D).to(dtype)
    # one key that dominates its row (a late rescale of the running maximum) and rows of V that would poison a sum if a
    # masked or out-of-range row leaked in with a non-zero weight
    k[T - 1] = q[T - 1, ::Hq // Hkv] * 3
    v[T // 2] = 50.0
    o = torch.full((T, Hq, D), float("nan"), dtype=dtype, device=device)
    sm_scale = 1.0 / (D ** 0.5)
    args = (q.to(device), k.to(device), v.to(device), o, k_buf.to(device), v_buf.to(device), qo_indptr.to(device),
            kv_indptr.to(device), kv_indices.to(device), None, None, max(ext), sm_scale, 0.0)
    ops.extend_attention_fwd(*args)
    want = O.extend_attention(q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, sm_scale, 0.0)
    _close(o, want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)


@pytest.mark.parametrize("seed", range(12))
def test_extend_attention_shared_kv_random_batches(ops, device, seed):
    """Seeded random batches through the shared-KV kernel (both forms by grid size): 1-6 requests, prefixes 0-700 (several
    128-row tiles, odd and even counts), 1-900 new tokens, the head layouts of the models in scope."""
    rng = np.random.RandomState(1000 + seed)
    Hq, Hkv = [(4, 1), (8, 2), (6, 3), (3, 3), (16, 2), (32, 8)][seed % 6]
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    B = int(rng.randint(1, 7))
    pre = [int(rng.choice([0, 0, rng.randint(1, 130), rng.randint(130, 700)])) for _ in range(B)]
    ext = [int(rng.choice([1, rng.randint(2, 65), rng.randint(65, 300), rng.randint(300, 900)])) for _ in range(B)]
    D = 128
    k_buf, v_buf, kv_indptr, kv_indices = _paged(B, pre, Hkv, D, D, dtype, seed=seed)
    T = sum(ext)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
    torch.manual_seed(seed)
    q = torch.randn(T, Hq, D).to(dtype)
    k = torch.randn(T, Hkv, D).to(dtype)
    v = torch.randn(T, Hkv, D).to(dtype)
    o = torch.full((T, Hq, D), float("nan"), dtype=dtype, device=device)
    sm_scale = 1.0 / (D ** 0.5)
    ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, k_buf.to(device), v_buf.to(device),
                             qo_indptr.to(device), kv_indptr.to(device), kv_indices.to(device), None, None, max(ext),
                             sm_scale, 0.0)
    want = O.extend_attention(q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, sm_scale, 0.0)
    _close(o, want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)


@pytest.mark.parametrize("fixture", ["extend_attention", "extend_attention_8c"])
def test_extend_attention_golden(ops, device, fixture):
    g = load_golden(fixture)
    for name in g["names"]:
        name = str(name)
        q, k, v = (torch.from_numpy(g[f"{name}_{x}"]).to(torch.bfloat16) for x in ("q", "k", "v"))
        kb, vb = (torch.from_numpy(g[f"{name}_{x}"]).to(torch.bfloat16) for x in ("kbuf", "vbuf"))
        sm_scale, cap = g[name + "_meta"]
        qo = torch.from_numpy(g[name + "_qo_indptr"])
        o = torch.empty(q.shape[0], q.shape[1], v.shape[2], dtype=torch.bfloat16, device=device)
        ext = (qo[1:] - qo[:-1]).max().item()
        ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, kb.to(device), vb.to(device),
                                 qo.to(device), torch.from_numpy(g[name + "_kv_indptr"]).to(device),
                                 torch.from_numpy(g[name + "_kv_indices"]).to(device), None, None, ext,
                                 float(sm_scale), float(cap))
        _close(o, torch.from_numpy(g[name + "_o"]), torch.bfloat16, rtol=3e-2, atol=3e-2)


def _tree_masks(g, pre, ext, prefix_bits):
    """Flat custom_mask + mask_indptr of a batch: sub-causal triangles with the diagonal kept, prefix columns all
    ones or random with the first column kept (the layout of test_triton_attention_kernels.py:121-139)."""
    blocks = []
    for p, e in zip(pre, ext):
        tri = torch.tril(torch.rand(e, e, generator=g) < 0.5) | torch.eye(e, dtype=torch.bool)
        pm = torch.rand(e, p, generator=g) < 0.6 if prefix_bits else torch.ones(e, p, dtype=torch.bool)
        if p:
            pm[:, 0] = True
        blocks.append(torch.cat([pm, tri], 1).flatten())
    indptr = torch.zeros(len(pre) + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.tensor([b.numel() for b in blocks]), 0)
    return torch.cat(blocks), indptr


def test_extend_attention_custom_mask_golden(ops, device):
    """HIP under a custom mask vs what the reference's Triton kernel wrote for the same inputs."""
    g = load_golden("extend_attention_mask")
    for name in g["names"]:
        name = str(name)
        q, k, v = (torch.from_numpy(g[f"{name}_{x}"]).to(torch.bfloat16) for x in ("q", "k", "v"))
        kb, vb = (torch.from_numpy(g[f"{name}_{x}"]).to(torch.bfloat16) for x in ("kbuf", "vbuf"))
        sm_scale, cap, skip = g[name + "_meta"]
        qo = torch.from_numpy(g[name + "_qo_indptr"])
        o = torch.empty(q.shape[0], q.shape[1], v.shape[2], dtype=torch.bfloat16, device=device)
        ext = (qo[1:] - qo[:-1]).max().item()
        ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, kb.to(device), vb.to(device),
                                 qo.to(device), torch.from_numpy(g[name + "_kv_indptr"]).to(device),
                                 torch.from_numpy(g[name + "_kv_indices"]).to(device),
                                 torch.from_numpy(g[name + "_mask"]).to(device).to(torch.bool),
                                 torch.from_numpy(g[name + "_mask_indptr"]).to(device), ext,
                                 float(sm_scale), float(cap), bool(skip))
        _close(o, torch.from_numpy(g[name + "_o"]), torch.bfloat16, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("skip_prefix", [True, False])
@pytest.mark.parametrize("pre,ext,Hq,Hkv,Dk,Dv,cap", [
    ([0, 13, 40], [20, 7, 70], 4, 2, 64, 64, 0.0),        # tile kernel, masked instantiation
    ([130, 0, 65], [64, 129, 300], 8, 2, 128, 128, 0.0),  # Llama heads: the masked form, not the shared-KV kernel
    ([9], [33], 4, 1, 128, 128, 25.0),                    # with a logit cap
    ([6, 0], [10, 3], 2, 1, 96, 64, 0.0),                 # 96 / 64
    ([40, 3], [17, 200], 2, 1, 192, 128, 0.0),            # MLA prefill shape
    ([5, 2], [9, 4], 2, 1, 80, 13, 0.0),                  # ragged head sizes: the one-wave kernel reads the mask
    ([7], [12], 2, 2, 256, 256, 0.0),                     # no MFMA instantiation: one-wave kernel
    ([11, 2], [6, 14], 4, 1, 576, 512, 0.0),              # MLA absorbed rows behind a chunked prefill: one-wave kernel
])
def test_extend_attention_custom_mask(ops, device, pre, ext, Hq, Hkv, Dk, Dv, cap, skip_prefix, dtype):
    g = torch.Generator().manual_seed(17)
    B, T = len(pre), sum(ext)
    k_buf, v_buf, kv_indptr, kv_indices = _paged(B, pre, Hkv, Dk, Dv, dtype, seed=Hq + Dk)
    q = (torch.randn(T, Hq, Dk, generator=g)).to(dtype)
    k = (torch.randn(T, Hkv, Dk, generator=g)).to(dtype)
    v = (torch.randn(T, Hkv, Dv, generator=g)).to(dtype)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
    mask, mask_indptr = _tree_masks(g, pre, ext, not skip_prefix)
    sm_scale = Dk ** -0.5
    o = torch.full((T, Hq, Dv), float("nan"), dtype=dtype, device=device)
    ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, k_buf.to(device), v_buf.to(device),
                             qo_indptr.to(device), kv_indptr.to(device), kv_indices.to(device), mask.to(device),
                             mask_indptr.to(device), max(ext), sm_scale, cap, skip_prefix)
    want = O.extend_attention(q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, sm_scale, cap, mask, mask_indptr,
                              skip_prefix)
    _close(o, want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)


@pytest.mark.parametrize("Hq,Hkv,Dk,Dv", [(4, 2, 128, 128), (2, 1, 80, 13)])   # the tile kernel | the one-wave kernel
def test_extend_attention_fully_masked_row_is_nan_on_every_route(ops, device, Hq, Hkv, Dk, Dv):
    """A query whose every key the custom mask removes gets NaN -- the reference's kernel computes exp(-inf - -inf) in its
    rescale (triton_ops/extend_attention.py:233-262), the oracle's softmax of an all -inf row is NaN too, and both HIP
    routes agree (the tile kernel wrote 0 where the one-wave kernel wrote NaN); the other rows are untouched."""
    dtype = torch.bfloat16
    pre, ext = [0, 3], [9, 6]
    g = torch.Generator().manual_seed(31)
    B, T = len(pre), sum(ext)
    k_buf, v_buf, kv_indptr, kv_indices = _paged(B, pre, Hkv, Dk, Dv, dtype, seed=9)
    q, k, v = (torch.randn(T, h, d, generator=g).to(dtype) for h, d in ((Hq, Dk), (Hkv, Dk), (Hkv, Dv)))
    qo = torch.zeros(B + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(torch.tensor(ext), 0)
    blocks = [torch.cat([torch.ones(e, p, dtype=torch.bool), torch.tril(torch.ones(e, e, dtype=torch.bool))], 1) for p, e in zip(pre, ext)]
    blocks[0][4, :] = False          # sequence 0 has no prefix: row 4 sees nothing at all
    blocks[1][2, :] = False          # sequence 1: row 2 loses its prefix bits too (skip_prefix_custom_mask = False)
    indptr = torch.zeros(B + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.tensor([b.numel() for b in blocks]), 0)
    mask = torch.cat([b.flatten() for b in blocks])
    o = torch.zeros(T, Hq, Dv, dtype=dtype, device=device)
    ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, k_buf.to(device), v_buf.to(device), qo.to(device),
                             kv_indptr.to(device), kv_indices.to(device), mask.to(device), indptr.to(device), max(ext),
                             Dk ** -0.5, 0.0, False)
    want = O.extend_attention(q, k, v, k_buf, v_buf, qo, kv_indptr, kv_indices, Dk ** -0.5, 0.0, mask, indptr, False)
    dead = [4, ext[0] + 2]
    assert torch.isnan(want[dead]).all() and torch.isnan(o[dead].float()).all()
    live = [i for i in range(T) if i not in dead]
    assert not torch.isnan(o[live].float()).any()
    _close(o[live], want[live], dtype, rtol=2e-2, atol=1.5e-2)


def test_extend_attention_custom_mask_properties(ops, device):
    """Size-independent properties at Llama-3-8B heads: (1) the mask that spells the default out (prefix ones + the
    causal triangle) gives what the unmasked launch gives; (2) with skip_prefix_custom_mask the prefix bits are not read
    (zeros there change nothing); (3) uint8 and bool masks are the same bytes; (4) a mask without mask_indptr is refused."""
    dtype = torch.bfloat16
    Hq, Hkv, D = 32, 8, 128
    pre, ext = [700, 0, 129], [64, 300, 5]
    g = torch.Generator().manual_seed(5)
    B, T = len(pre), sum(ext)
    k_buf, v_buf, kv_indptr, kv_indices = _paged(B, pre, Hkv, D, D, dtype, seed=3)
    q, k, v = (torch.randn(T, h, D, generator=g).to(dtype).to(device) for h in (Hq, Hkv, Hkv))
    qo = torch.zeros(B + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(torch.tensor(ext), 0)
    blocks = [torch.cat([torch.ones(e, p, dtype=torch.bool), torch.tril(torch.ones(e, e, dtype=torch.bool))], 1).flatten()
              for p, e in zip(pre, ext)]
    indptr = torch.zeros(B + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.tensor([b.numel() for b in blocks]), 0)
    causal = torch.cat(blocks)
    common = (k_buf.to(device), v_buf.to(device), qo.to(device), kv_indptr.to(device), kv_indices.to(device))

    def run(mask, mp, skip=True):
        o = torch.empty(T, Hq, D, dtype=dtype, device=device)
        ops.extend_attention_fwd(q, k, v, o, *common, None if mask is None else mask.to(device),
                                 None if mp is None else mp.to(device), max(ext), D ** -0.5, 0.0, skip)
        return o.float().cpu()

    plain = run(None, None)
    for skip in (True, False):
        torch.testing.assert_close(run(causal, indptr, skip), plain, rtol=2e-2, atol=1e-2)   # two kernels, one answer
    holes = causal.clone()
    off = 0
    for p, e in zip(pre, ext):      # zero every prefix bit
        holes[off:off + e * (p + e)].view(e, p + e)[:, :p] = False
        off += e * (p + e)
    assert torch.equal(run(holes, indptr, True), run(causal, indptr, True))
    assert torch.equal(run(causal.to(torch.uint8), indptr, False), run(causal, indptr, False))
    with pytest.raises(RuntimeError, match="mask_indptr"):
        run(causal, None)


def test_extend_equals_decode_on_last_token(ops, device):
    """Size-independent property at a BASELINE-sized shape (Llama-3-8B heads, ctx 4k): the last query
    row of an extend over [prefix | new tokens] equals decode attention over the same KV rows."""
    dtype = torch.bfloat16
    Hq, Hkv, D = 32, 8, 128
    pre, ext = 3000, 1096
    torch.manual_seed(0)
    N = pre + ext + 5
    k_all = torch.randn(N, Hkv, D, device=device).to(dtype)
    v_all = torch.randn(N, Hkv, D, device=device).to(dtype)
    perm = (torch.randperm(N - 1, device=device) + 1)[:pre + ext].to(torch.int32)
    q = torch.randn(ext, Hq, D, device=device).to(dtype)
    k_ext = k_all[perm[pre:].long()].contiguous()
    v_ext = v_all[perm[pre:].long()].contiguous()
    o = torch.empty(ext, Hq, D, dtype=dtype, device=device)
    qo = torch.tensor([0, ext], dtype=torch.int32, device=device)
    kvp = torch.tensor([0, pre], dtype=torch.int32, device=device)
    ops.extend_attention_fwd(q, k_ext, v_ext, o, k_all, v_all, qo, kvp, perm[:pre].contiguous(), None, None, ext)
    od = torch.empty(1, Hq, D, dtype=dtype, device=device)
    logits = torch.empty(1, Hq, 16, D + 1, dtype=torch.float32, device=device)
    full = torch.tensor([0, pre + ext], dtype=torch.int32, device=device)
    ops.decode_attention_fwd(q[-1:].contiguous(), k_all, v_all, od, full, perm, logits, 16, 1.0 / D ** 0.5)
    _close(o[-1:], od, dtype, rtol=2e-2, atol=8e-3)
    # and an independent torch check of a middle row
    row = 517
    kk = torch.cat([k_all[perm[:pre].long()], k_ext[:row + 1]]).float()
    vv = torch.cat([v_all[perm[:pre].long()], v_ext[:row + 1]]).float()
    for h in (0, 13, 31):
        s = (kk[:, h // 4] @ q[row, h].float()) / D ** 0.5
        ref = torch.softmax(s, 0) @ vv[:, h // 4]
        _close(o[row, h], ref, dtype, rtol=2e-2, atol=8e-3)


# ----------------------------------------------------------------------------- sampling
@pytest.mark.parametrize("B,V", [(1, 50272), (32, 128256), (7, 1000), (3, 129280)])
def test_greedy_argmax(ops, device, B, V):
    torch.manual_seed(V)
    x = torch.randn(B, V)
    x[0, V - 1] = 100.0
    if B > 1:
        x[1, 5] = x[1, 77] = 50.0  # tie -> lowest index
    got = ops.greedy_argmax(x.to(device))
    assert got.dtype == torch.int32 and got.cpu().tolist() == O.greedy_argmax(x).tolist()
    got64 = ops.greedy_argmax(x.to(device).to(torch.bfloat16), torch.int64)
    assert got64.cpu().tolist() == O.greedy_argmax(x.to(torch.bfloat16)).tolist()


@pytest.mark.parametrize("B,H,V", [(1, 768, 50272), (5, 4096, 32000), (70, 2048, 1000)])
def test_lm_head_argmax(ops, device, B, H, V):
    torch.manual_seed(B + H)
    h = torch.randn(B, H).to(torch.bfloat16)
    w = (torch.randn(V, H) * 0.05).to(torch.bfloat16)
    logits, ids = ops.lm_head_argmax(h.to(device), w.to(device))
    want = O.logits_last_token(h.float(), None, w.float())
    _close(logits, want, torch.float32, rtol=1e-3, atol=2e-3)
    top2 = want.topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-2  # rows whose winner is not a near-tie
    assert torch.equal(ids.cpu()[safe].long(), want.argmax(-1)[safe])


# ----------------------------------------------------------------------------- MoE
def _same_topk(w, ids, rw, rids, rtol, atol):
    for t in range(ids.shape[0]):
        a = sorted(zip(ids[t].tolist(), w[t].tolist()))
        b = sorted(zip(rids[t].tolist(), rw[t].tolist()))
        assert [x[0] for x in a] == [x[0] for x in b], (t, a, b)
        np.testing.assert_allclose([x[1] for x in a], [x[1] for x in b], rtol=rtol, atol=atol)


def test_moe_routing_golden(ops, device):
    g = load_golden("moe_topk")
    for name in ("native_e8", "native_e64"):
        k, ren = g[name + "_meta"]
        w, ids = ops.topk_softmax(torch.from_numpy(g[name + "_gate"]).to(device), int(k), bool(ren))
        _same_topk(w.cpu().numpy(), ids.cpu().numpy(), g[name + "_w"], g[name + "_ids"], 1e-5, 1e-6)
    for name, dt, scoring in (("grouped_f32", torch.float32, "softmax"), ("grouped_bf16", torch.bfloat16, "softmax"),
                              ("grouped_sigmoid", torch.float32, "sigmoid")):
        k, ren, ng, tg = g[name + "_meta"]
        w, ids = ops.grouped_topk(from_bits(g[name + "_gate"], dt).to(device), int(k), bool(ren), int(ng), int(tg),
                                  None, scoring)
        tol = (1e-5, 1e-6) if dt == torch.float32 else (8e-3, 1e-3)
        _same_topk(w.cpu().numpy(), ids.cpu().numpy(), g[name + "_w"], g[name + "_ids"], *tol)
    for name, dt in (("biased_f32", torch.float32), ("biased_bf16", torch.bfloat16)):
        k, ren, ng, tg = g[name + "_meta"]
        w, ids = ops.grouped_topk(from_bits(g[name + "_gate"], dt).to(device), int(k), bool(ren), int(ng), int(tg),
                                  torch.from_numpy(g[name + "_bias"]).to(device))
        tol = (1e-5, 1e-6) if dt == torch.float32 else (8e-3, 1e-3)
        _same_topk(w.cpu().numpy(), ids.cpu().numpy(), g[name + "_w"], g[name + "_ids"], *tol)


@pytest.mark.parametrize("T,E,k", [(1, 64, 6), (33, 64, 6), (222, 8, 2), (4096, 256, 8)])
def test_topk_softmax_random(ops, device, T, E, k):
    torch.manual_seed(T + E)
    gate = torch.randn(T, E)
    w, ids = ops.topk_softmax(gate.to(device), k, True)
    rw, rids = O.fused_topk_native(gate, k, True)
    _same_topk(w.cpu().numpy(), ids.cpu().numpy(), rw.numpy(), rids.numpy(), 1e-5, 1e-6)


@pytest.mark.parametrize("numel_tokens,topk,E,block", [(1, 6, 64, 64), (33, 6, 64, 64), (4096, 8, 256, 64), (5, 2, 8, 64)])
def test_moe_align_block_size(ops, device, numel_tokens, topk, E, block):
    """Same comparison as sgl-kernel/tests/test_moe_align.py:151-222 (expert_ids, num_tokens_post_pad),
    plus the permutation property of sorted_token_ids."""
    torch.manual_seed(numel_tokens + E)
    ids = torch.stack([torch.randperm(E)[:topk] for _ in range(numel_tokens)]).to(torch.int32)
    numel = ids.numel()
    max_sorted = numel + E * (block - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
    expert_ids = torch.full(((max_sorted + block - 1) // block,), -1, dtype=torch.int32, device=device)
    npp = torch.empty(1, dtype=torch.int32, device=device)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=device)
    ops.moe_align_block_size(ids.to(device), E, block, sorted_ids, expert_ids, npp, None, cumsum)
    rs, re_, rn = O.moe_align_block_size(ids, block, E)
    n = int(npp.item())
    assert n == int(rn)
    assert torch.equal(expert_ids.cpu()[: n // block], re_[: n // block])
    got = sorted_ids.cpu()
    flat = ids.flatten()
    # every expert owns the same padded range and the same multiset of tokens as in the stable oracle
    # (order inside an expert is unspecified in the reference too: atomicAdd scatter, moe_align_kernel.cu:77-95)
    eids = re_[: n // block].tolist()
    start = 0
    while start < len(eids):
        end = start
        while end < len(eids) and eids[end] == eids[start]:
            end += 1
        lo, hi = start * block, end * block
        assert sorted(got[lo:hi].tolist()) == sorted(rs[lo:hi].tolist()), f"expert {eids[start]}"
        real = [i for i in got[lo:hi].tolist() if i < numel]
        assert all(int(flat[i]) == eids[start] for i in real)
        start = end
    assert (got[n:] == numel).all()


def test_silu_and_mul_golden(ops, device):
    """SiluAndMul.forward_native of the reference (golden/silu_and_mul.npz): the kernel rounds once (like the
    sgl-kernel op), forward_native twice, so outputs agree within one unit in the last place."""
    g = load_golden("silu_and_mul")
    for ci in range(int(g["n"])):
        dt = DTYPES[str(g[f"c{ci}_dtype"])]
        if dt == torch.float32:
            continue
        x = from_bits(g[f"c{ci}_x"], dt)
        y = ops.silu_and_mul(x.to(device)).cpu()
        assert torch.equal(y, O.silu_and_mul(x)) or (y.float() - O.silu_and_mul(x).float()).abs().max() < 1e-2
        want = from_bits(g[f"c{ci}_y"], dt).float()
        ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
        assert torch.all((y.float() - want).abs() <= 1.01 * ulp * want.abs() + 1e-6), ci


def test_moe_align_golden(ops, device):
    """Against the Triton implementation of sgl-kernel/tests/test_moe_align.py run under the interpreter
    (golden/moe_align.npz): expert_ids, num_tokens_post_pad and, per expert, the token set."""
    g = load_golden("moe_align")
    for ci in range(int(g["n"])):
        bs, T, k, E = (int(x) for x in g[f"c{ci}_meta"])
        ids = torch.from_numpy(g[f"c{ci}_topk_ids"])
        numel = ids.numel()
        max_sorted = numel + E * (bs - 1)
        sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
        expert_ids = torch.full(((max_sorted + bs - 1) // bs,), -1, dtype=torch.int32, device=device)
        npp = torch.empty(1, dtype=torch.int32, device=device)
        cumsum = torch.empty(E + 1, dtype=torch.int32, device=device)
        ops.moe_align_block_size(ids.to(device), E, bs, sorted_ids, expert_ids, npp, None, cumsum)
        n = int(g[f"c{ci}_n_post"][0])
        assert int(npp.item()) == n
        nb = n // bs
        assert np.array_equal(expert_ids.cpu().numpy()[:nb], g[f"c{ci}_expert_ids"][:nb])
        got, ref = sorted_ids.cpu().numpy(), g[f"c{ci}_sorted"]
        eids = g[f"c{ci}_expert_ids"][:nb]
        for e in np.unique(eids):
            blks = np.nonzero(eids == e)[0]
            seg = slice(blks[0] * bs, (blks[-1] + 1) * bs)
            assert sorted(got[seg].tolist()) == sorted(ref[seg].tolist()), (ci, e)


def test_fused_experts_golden(ops, device):
    """fused_moe_native.py / torch_naive_moe outputs of the reference (golden/fused_moe.npz, fp32 inputs)
    against the fused-experts layer in bf16, at the reference test's bf16 bar (test_fused_moe.py:31-44)."""
    from semi_pd_amd.layers.moe import fused_experts
    g = load_golden("fused_moe")
    for ci in range(int(g["n"])):
        m, n, k, e, topk = (int(x) for x in g[f"c{ci}_meta"])
        a, w1, w2, score = (torch.from_numpy(g[f"c{ci}_{x}"]) for x in ("a", "w1", "w2", "score"))
        for renorm, key in ((False, "out_native"), (True, "out_renorm")):
            tw, ti = ops.topk_softmax(score.to(device), topk, renorm)
            out = fused_experts(a.to(device, torch.bfloat16), w1.to(device, torch.bfloat16),
                                w2.to(device, torch.bfloat16), tw, ti)
            torch.testing.assert_close(out.float().cpu(), torch.from_numpy(g[f"c{ci}_{key}"]), rtol=1e-1, atol=1e-2)


@pytest.mark.parametrize("T,N,K,E,topk", [(1, 128, 128, 8, 2), (33, 1024, 511 + 1, 8, 2), (64, 1408, 2048, 64, 6),
                                           (222, 128, 1024, 64, 6)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block", [64, 128])
def test_fused_moe_pipeline(ops, device, T, N, K, E, topk, dtype, block):
    """align -> grouped GEMM1 -> silu_and_mul -> grouped GEMM2 (x routed weight) -> moe_sum against the
    reference's naive MoE (test/srt/test_fused_moe.py:46-63, tolerances :31-44)."""
    torch.manual_seed(T + N)
    a = (torch.randn(T, K) / 10).to(dtype)
    w1 = (torch.randn(E, 2 * N, K) / 10).to(dtype)
    w2 = (torch.randn(E, K, N) / 10).to(dtype)
    gate = torch.randn(T, E)
    tw, tid = ops.topk_softmax(gate.to(device), topk, True)
    numel = T * topk
    max_sorted = numel + E * (block - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
    expert_ids = torch.empty((max_sorted + block - 1) // block, dtype=torch.int32, device=device)
    npp = torch.empty(1, dtype=torch.int32, device=device)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=device)
    ops.moe_align_block_size(tid, E, block, sorted_ids, expert_ids, npp, None, cumsum)
    c1 = torch.empty(numel, 2 * N, dtype=dtype, device=device)
    ops.moe_grouped_gemm(a.to(device), w1.to(device), c1, None, sorted_ids, expert_ids, npp, numel, topk, False, block)
    c2 = ops.silu_and_mul(c1)
    c3 = torch.empty(numel, K, dtype=dtype, device=device)
    ops.moe_grouped_gemm(c2, w2.to(device), c3, tw.flatten().contiguous(), sorted_ids, expert_ids, npp, numel, 1, True, block)
    out = ops.moe_sum(c3.view(T, topk, K))
    want = O.fused_moe(a, w1, w2, tw.cpu(), tid.cpu())
    _close(out, want, dtype, rtol=1e-1, atol=1e-2)


@pytest.mark.parametrize("T,topk,H", [(1, 6, 2048), (37, 6, 2048), (64, 2, 136), (300, 8, 7168)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scale,with_addend", [(1.0, True), (16.0, True), (2.5, False), (0.3, True)])
def test_moe_sum_scale_add_has_the_bits_of_the_three_launches(ops, device, T, topk, H, dtype, scale, with_addend):
    """DeepseekV2MoE.forward (models/deepseek_v2.py:139-160): moe_sum, `* routed_scaling_factor`, `+ shared_output`;
    the one-launch form rounds to the storage type after each of the three like the separate kernels do."""
    torch.manual_seed(T * topk + H)
    x = torch.randn(T, topk, H, device=device).to(dtype)
    addend = torch.randn(T, H, device=device).to(dtype) if with_addend else None
    want = ops.moe_sum(x)
    if scale != 1.0:
        want = want * scale
    if addend is not None:
        want = want + addend
    got = ops.moe_sum_scale_add(x, scale, addend)
    assert torch.equal(got, want)
    if addend is not None:
        with pytest.raises(RuntimeError):
            ops.moe_sum_scale_add(x, scale, addend[:, : H // 2])


@pytest.mark.parametrize("T,topk,E,N,K", [(512, 6, 64, 2816, 2048), (700, 4, 8, 352, 192), (400, 8, 16, 2048, 1408),
                                          (2100, 1, 3, 96, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("routed", [False, True])
def test_moe_grouped_gemm_prefill_sized_rows(ops, device, T, topk, E, N, K, dtype, routed):
    """The tiled grouped GEMM (moe_tiled_gemm.hip: block_m 128 and at least 2048 routed rows) row by row against an fp32
    product with the routed expert's weights: tiles past N (352, 96), k-blocks that do not fill the DMA ring (K = 64, 192),
    experts without rows, ragged last blocks, both epilogues (GEMM1: plain, rows through sorted id / topk; GEMM2: times the
    routed weight, rows through the sorted id)."""
    torch.manual_seed(T + N + K)
    numel = T * topk
    div = 1 if routed else topk
    a = (torch.randn(numel // div, K) / 4).to(dtype)
    w = (torch.randn(E, N, K) / 4).to(dtype)
    gate = torch.randn(T, E)
    if E > 4:
        gate[:, 1] -= 100.0   # an expert nobody is routed to
    tw, tid = ops.topk_softmax(gate.to(device), topk, True)
    max_sorted = numel + E * 127
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
    expert_ids = torch.empty((max_sorted + 127) // 128, dtype=torch.int32, device=device)
    npp = torch.empty(1, dtype=torch.int32, device=device)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=device)
    ops.moe_align_block_size(tid, E, 128, sorted_ids, expert_ids, npp, None, cumsum)
    c = torch.full((numel, N), float("nan"), dtype=dtype, device=device)
    ops.moe_grouped_gemm(a.to(device), w.to(device), c, tw.flatten().contiguous() if routed else None, sorted_ids,
                         expert_ids, npp, numel, div, routed, 128)
    # fp32 product per expert (on the device: gathering one weight matrix per routed row on the host was 70 GB of copies)
    flat_e = tid.flatten().long()
    rows = torch.arange(numel, device=device) // div
    a32, w32 = a.to(device).float(), w.to(device).float()
    want = torch.zeros(numel, N, device=device)
    for e in flat_e.unique().tolist():
        sel = (flat_e == e).nonzero().flatten()
        want[sel] = a32[rows[sel]] @ w32[e].t()
    if routed:
        want = want * tw.flatten().float()[:, None]
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    torch.testing.assert_close(c.float(), want, rtol=tol, atol=tol * float(want.abs().max()))


@pytest.mark.parametrize("T,topk,E,Nh,K", [(512, 6, 64, 1408, 2048), (700, 4, 8, 352, 192), (2100, 1, 3, 96, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_moe_gemm1_with_silu_epilogue_has_the_bits_of_the_two_calls(ops, device, T, topk, E, Nh, K, dtype):
    """semipd_moe_grouped_gemm_silu (prefill-sized calls) against semipd_moe_grouped_gemm + semipd_silu_and_mul, bit for
    bit: partial last column tile (352, 96), short K, an expert without rows, ragged last blocks."""
    torch.manual_seed(T + Nh)
    numel = T * topk
    a = (torch.randn(T, K) / 4).to(dtype).to(device)
    w = (torch.randn(E, 2 * Nh, K) / 4).to(dtype).to(device)
    gate = torch.randn(T, E)
    if E > 4:
        gate[:, 1] -= 100.0
    tw, tid = ops.topk_softmax(gate.to(device), topk, True)
    max_sorted = numel + E * 127
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
    expert_ids = torch.empty((max_sorted + 127) // 128, dtype=torch.int32, device=device)
    npp = torch.empty(1, dtype=torch.int32, device=device)
    cumsum = torch.empty(E + 1, dtype=torch.int32, device=device)
    ops.moe_align_block_size(tid, E, 128, sorted_ids, expert_ids, npp, None, cumsum)
    fused = ops.moe_grouped_gemm_silu(a, w, sorted_ids, expert_ids, npp, numel, topk, 128)
    assert fused is not None and fused.shape == (numel, Nh)
    c1 = torch.empty(numel, 2 * Nh, dtype=dtype, device=device)
    ops.moe_grouped_gemm(a, w, c1, None, sorted_ids, expert_ids, npp, numel, topk, False, 128)
    want = ops.silu_and_mul(c1)
    assert torch.equal(fused.view(torch.int16), want.view(torch.int16))
    assert ops.moe_grouped_gemm_silu(a[:100], w, sorted_ids, expert_ids, npp, 100 * topk, topk, 128) is None   # decode-sized


def test_fused_experts_layer_prefill_sized(ops, device):
    """layers.moe.fused_experts above the decode threshold (T * topk > 2048 -> 128-row blocks): many rows
    per expert, several blocks per expert, ragged last blocks."""
    from semi_pd_amd.layers.moe import fused_experts
    torch.manual_seed(3)
    T, N, K, E, topk = 700, 256, 512, 8, 4
    a = (torch.randn(T, K) / 10).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * N, K) / 10).to(torch.bfloat16)
    w2 = (torch.randn(E, K, N) / 10).to(torch.bfloat16)
    gate = torch.randn(T, E)
    gate[:, 0] += 2.0  # skewed routing: expert 0 gets most tokens
    tw, tid = ops.topk_softmax(gate.to(device), topk, True)
    out = fused_experts(a.to(device), w1.to(device), w2.to(device), tw, tid)
    want = O.fused_moe(a, w1, w2, tw.cpu(), tid.cpu())
    _close(out, want, torch.bfloat16, rtol=1e-1, atol=1e-2)


# ----------------------------------------------------------------------------- decode-sized dense layers
# UnquantizedLinearMethod.apply = F.linear (layers/linear.py:165-172); bf16 tolerance of the fused-MoE
# GEMM test (test/srt/test_fused_moe.py:31-44)
@pytest.mark.parametrize("M", [1, 7, 16, 33, 64, 130])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (1000, 512), (64, 96)])
@pytest.mark.parametrize("num_cus", [0, 128])
def test_linear_split_k(ops, device, M, N, K, num_cus):
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(torch.bfloat16)
    w = (torch.randn(N, K) * 0.05).to(torch.bfloat16)
    want = x.float() @ w.float().T
    got = ops.linear(x.to(device), w.to(device), num_cus=num_cus)
    torch.testing.assert_close(got.cpu().float(), want, rtol=2e-2, atol=2e-2 * float(want.abs().max()))
    # deterministic (fixed-order split-K reduction) and the counters are left clean for the next call
    again = ops.linear(x.to(device), w.to(device), num_cus=num_cus)
    assert torch.equal(got, again)


def test_linear_strided_rows_and_f16(ops, device):
    torch.manual_seed(5)
    big = torch.randn(9, 2 * 512).to(torch.float16)
    w = (torch.randn(256, 512) * 0.05).to(torch.float16)
    x = big.to(device)[:, 512:]  # row stride 1024, 16-byte aligned start
    got = ops.linear(x, w.to(device))
    torch.testing.assert_close(got.cpu().float(), big[:, 512:].float() @ w.float().T, rtol=2e-3, atol=2e-2)
    with pytest.raises(RuntimeError):
        ops.linear(torch.randn(300, 512, device=device, dtype=torch.float16), w.to(device))


# ----------------------------------------------------------------------------- decode batches: LDS-DMA streaming linear
@pytest.mark.parametrize("M", [1, 7, 16, 17, 33, 48, 64, 65, 80, 96, 97, 128])   # 65+: the wide form (rings of 2 / 3 slots)
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (1008, 512), (16, 1024),
                                 # Llama-3-8B per rank at TP = 8 / TP = 4: qkv, o_proj, down_proj
                                 (768, 4096), (4096, 512), (4096, 1792), (1536, 4096), (4096, 1024), (4096, 3584)])
def test_stream_linear(ops, device, M, N, K):
    """F.linear semantics (layers/linear.py:165-172) from the weight-streaming kernel, bf16 bar of
    test_fused_moe.py:31-44; deterministic (K slices are summed in slice order)."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(torch.bfloat16)
    w = (torch.randn(N, K) * 0.05).to(torch.bfloat16)
    want = x.float() @ w.float().T
    got = ops.stream_linear(x.to(device), w.to(device))
    torch.testing.assert_close(got.cpu().float(), want, rtol=2e-2, atol=2e-2 * float(want.abs().max()))
    assert torch.equal(got, ops.stream_linear(x.to(device), w.to(device)))


@pytest.mark.parametrize("M", [1, 16, 31, 64, 66, 96, 128])
@pytest.mark.parametrize("inter,K", [(14336, 4096), (1408, 2048), (48, 512), (1792, 4096), (3584, 4096)])   # last two: TP = 8 / 4
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_stream_linear_fused_silu_mul(ops, device, M, inter, K, dtype):
    """gate_up_proj + SiluAndMul in one launch (models/llama.py:88-92): the GEMM outputs are rounded to the
    activation type before the activation, so the result equals the oracle's two-step value up to the
    accumulation order of the GEMM."""
    torch.manual_seed(M + inter)
    x = torch.randn(M, K).to(dtype)
    w = (torch.randn(2 * inter, K) * 0.05).to(dtype)
    gate_up = (x.float() @ w.float().T).to(dtype)
    want = O.silu_and_mul(gate_up)
    got = ops.stream_linear(x.to(device), w.to(device), fuse_silu_mul=True)
    assert got.shape == (M, inter)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(got.cpu().float(), want.float(), **tol)
    # and the unfused product path: stream_linear, then the activation kernel.  Same K slicing; since round 5 every workgroup
    # walks its k-blocks from its own starting block (csrc/stream_linear.hip: rot) and the two kernels cut the rows into
    # workgroups differently, so a GEMM output may round the other way before the activation: the oracle's tolerance
    two_step = ops.silu_and_mul(ops.stream_linear(x.to(device), w.to(device)))
    torch.testing.assert_close(got.float(), two_step.float(), **tol)


def test_stream_linear_strided_rows_and_limits(ops, device):
    torch.manual_seed(5)
    big = torch.randn(9, 2 * 512).to(torch.float16)
    w = (torch.randn(256, 512) * 0.05).to(torch.float16)
    x = big.to(device)[:, 512:]  # row stride 1024, 16-byte aligned start
    got = ops.stream_linear(x, w.to(device))
    torch.testing.assert_close(got.cpu().float(), big[:, 512:].float() @ w.float().T, rtol=2e-3, atol=2e-2)
    assert ops.stream_linear_is_supported(torch.randn(128, 512, device=device, dtype=torch.float16), w.to(device))
    assert not ops.stream_linear_is_supported(torch.randn(129, 512, device=device, dtype=torch.float16), w.to(device))
    assert not ops.stream_linear_is_supported(torch.randn(8, 96, device=device, dtype=torch.float16),
                                              torch.randn(64, 96, device=device, dtype=torch.float16))  # k % 128
    with pytest.raises(RuntimeError):
        ops.stream_linear(torch.randn(129, 512, device=device, dtype=torch.float16), w.to(device))


@pytest.mark.parametrize("M", [1, 16, 40, 64, 72, 128])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 14336), (2048, 1280), (512, 1024)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_stream_linear_planes_into_fused_add_rmsnorm(ops, device, M, N, K, dtype):
    """o_proj / down_proj -> fused add + RMSNorm with the K-slice reduction done by the norm kernel: the bits of the
    unfused sequence (stream_linear, then fused_add_rmsnorm), and the oracle's values."""
    torch.manual_seed(M + N)
    x = torch.randn(M, K).to(dtype)
    w = (torch.randn(N, K) * 0.05).to(dtype)
    res = torch.randn(M, N).to(dtype)
    nw = (torch.rand(N) + 0.5).to(dtype)
    xd, wd = x.to(device), w.to(device)
    y = ops.stream_linear(xd, wd)
    r1 = res.to(device).clone()
    ops.fused_add_rmsnorm(y, r1, nw.to(device), 1e-5)          # y <- norm(y + r1), r1 <- y + r1
    planes = ops.stream_linear_planes(xd, wd)
    r2 = res.to(device).clone()
    out = ops.fused_add_rmsnorm_planes(planes, r2, nw.to(device), 1e-5)
    assert torch.equal(r1, r2) and torch.equal(y, out)
    gemm = (x.float() @ w.float().T).to(dtype)
    want, want_res = O.fused_add_rms_norm(gemm, res, nw, 1e-5)
    _close(out, want, dtype, rtol=3e-2, atol=3e-2)
    _close(r2, want_res, dtype, rtol=2e-2, atol=2e-2 * float(want_res.float().abs().max()))


@pytest.mark.parametrize("M,N,K", [(1357, 4096, 14336), (1024, 4096, 4096), (300, 768, 512), (129, 272, 1024), (2048, 4096, 14336),
                                   (65, 256, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_tall_planes_into_fused_add_rmsnorm(ops, device, M, N, K, dtype):
    """o_proj / down_proj of a prefill batch -> fused add + RMSNorm with the tiled GEMM's K-slice reduction done by the norm
    kernel (ops.gemm_tall_planes): the bits of gemm_tall followed by fused_add_rmsnorm for the split the launch picks by
    itself and for forced splits (one slice: the tensor comes back, no planes), and the oracle's values."""
    import os
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K).to(dtype)
    w = (torch.randn(N, K) * K ** -0.5).to(dtype)
    res = torch.randn(M, N).to(dtype)
    nw = (torch.rand(N) + 0.5).to(dtype)
    xd, wd, nwd = x.to(device), w.to(device), nw.to(device)
    seen = set()
    try:
        for ks in ("0", "1", "2", "3", "4"):
            os.environ["SEMIPD_G8_KS"] = ks
            y = ops.gemm_tall(xd, wd)
            r1 = res.to(device).clone()
            ops.fused_add_rmsnorm(y, r1, nwd, 1e-5)
            p = ops.gemm_tall_planes(xd, wd)
            r2 = res.to(device).clone()
            if isinstance(p, ops.SplitKPlanes):
                assert p.ksplit > 1 and p.shape == (M, N)
                seen.add(p.ksplit)
                out = ops.fused_add_rmsnorm_planes(p, r2, nwd, 1e-5)
            else:                                   # the launch did not slice K: a finished [M, N] tensor
                assert tuple(p.shape) == (M, N)
                seen.add(1)
                out = p
                ops.fused_add_rmsnorm(out, r2, nwd, 1e-5)
            assert torch.equal(r1, r2) and torch.equal(y, out), ks
    finally:
        os.environ.pop("SEMIPD_G8_KS", None)
    assert 1 in seen
    if (M, N, K) == (1357, 4096, 14336):
        assert 2 in seen
    if (M, N, K) == (2048, 4096, 14336):
        assert max(seen) == 2            # two planes of 2048 x 4096 fill the 64 MiB workspace: a forced 3 or 4 comes back as 2
    gemm = (x.float() @ w.float().T).to(dtype)
    want, want_res = O.fused_add_rms_norm(gemm, res, nw, 1e-5)
    _close(out, want, dtype, rtol=3e-2, atol=3e-2)
    _close(r2, want_res, dtype, rtol=2e-2, atol=2e-2 * float(want_res.float().abs().max()))


def test_gemm_tall_planes_must_be_consumed_before_the_next_gemm(ops, device):
    """The planes live in the stream's GEMM workspace: a GEMM in between makes the consumer refuse them."""
    import os
    x = torch.randn(300, 1024, device=device, dtype=torch.bfloat16)
    w = torch.randn(512, 1024, device=device, dtype=torch.bfloat16) * 0.03
    os.environ["SEMIPD_G8_KS"] = "2"
    try:
        p = ops.gemm_tall_planes(x, w)
        assert isinstance(p, ops.SplitKPlanes) and p.ksplit == 2
        ops.gemm_tall(x, w)
        with pytest.raises(RuntimeError, match="overwritten"):
            ops.fused_add_rmsnorm_planes(p, torch.zeros(300, 512, device=device, dtype=torch.bfloat16),
                                         torch.ones(512, device=device, dtype=torch.bfloat16), 1e-5)
    finally:
        os.environ.pop("SEMIPD_G8_KS", None)


def test_row_parallel_layer_defers_the_tiled_gemms_reduction_to_the_norm(ops, device, monkeypatch):
    """RowParallelLinear.forward(defer_reduce=True) above the streaming kernel's rows: where dense_linear takes the tiled
    GEMM the layer hands its planes to RMSNorm(x, residual) -- the bits of the reducing form (SEMIPD_TALL_PLANES=0)."""
    from semi_pd_amd.layers import basic as L
    torch.manual_seed(3)
    layer = L.RowParallelLinear(2048, 1024, params_dtype=torch.bfloat16).to(device)
    layer.weight.data.copy_((torch.randn(1024, 2048) * 0.02).to(torch.bfloat16))
    norm = L.RMSNorm(1024).to(device)
    norm.weight.data = (torch.rand(1024, device=device) + 0.5).to(torch.bfloat16)
    x = torch.randn(200, 2048, device=device, dtype=torch.bfloat16)
    res = torch.randn(200, 1024, device=device, dtype=torch.bfloat16)
    monkeypatch.setenv("SEMIPD_G8_KS", "2")
    assert L._takes_tiled_gemm(x, layer.weight)       # 65 .. 256 rows of an untuned layer: the tiled GEMM
    monkeypatch.setitem(L._STREAM_LINEAR, "enabled", True)   # what the model runner sets for the process
    monkeypatch.setattr(L, "_TALL_PLANES", True)
    p = layer(x, defer_reduce=True)
    assert isinstance(p, ops.SplitKPlanes)
    r1 = res.clone()
    y1, r1 = norm(p, r1)
    monkeypatch.setattr(L, "_TALL_PLANES", False)
    t = layer(x, defer_reduce=True)
    assert isinstance(t, torch.Tensor)
    r2 = res.clone()
    y2, r2 = norm(t, r2)
    assert torch.equal(y1, y2) and torch.equal(r1, r2)


_KEEP_ALIVE = []


def test_abort_of_a_failed_stream_capture(device):
    """A decode step that cannot be captured (a TP collective of a backend without capture support) must leave the process
    able to run eagerly.  On this ROCm an invalidated capture cannot be ended, and while its stream exists every
    synchronising call of the process fails: semipd_stream_abort_capture destroys the (caller-owned) stream."""
    import ctypes
    from semi_pd_amd import _lib
    lib = _lib.load()
    x = torch.ones(1024, device=device)

    def own_stream():
        raw = ctypes.c_void_p()
        _lib.check(lib.semipd_stream_create(0, ctypes.addressof(raw)), "stream_create")
        st = torch.cuda.ExternalStream(raw.value, device=device)
        st.wait_stream(torch.cuda.current_stream())
        return raw.value, st

    raw, stream = own_stream()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception):
        with torch.cuda.graph(g, stream=stream):
            y = x * 2
            y.sum().item()          # a synchronising call invalidates the capture
    assert lib.semipd_stream_abort_capture(raw) == 0
    # what was allocated on the dead stream goes back to the allocator now; the framework's bookkeeping on that stream
    # fails ONE later call with hipErrorInvalidValue -- the engine's fallback path (model_runner.init_cuda_graphs) makes
    # that call a throw-away one, like here
    _KEEP_ALIVE.append((y, g, stream))   # freeing what lives on the dead stream makes the allocator touch it again
    try:                                # ONE later call fails with hipErrorInvalidValue: a throw-away one
        (x + 1).cpu()
    except Exception:
        pass
    raw1, stream1 = own_stream()        # a small clean capture on a fresh stream settles the framework's capture state
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=stream1):
        y1 = x * 5
    _KEEP_ALIVE.append((y1, g1, stream1))
    assert float((x + 1).cpu().sum()) == 2048.0
    t = torch.arange(8).pin_memory().to(device, non_blocking=True)             # the call that failed in the engine
    assert int(t.sum()) == 28
    raw2, stream2 = own_stream()                                               # and a fresh capture works
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=stream2):
        y2 = x * 3
    g2.replay()
    torch.cuda.synchronize()
    assert float(y2.sum()) == 3072.0


@pytest.mark.parametrize("M", [1, 7, 33, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv_dtype", [None, torch.float8_e4m3fn])
@pytest.mark.parametrize("heads", [(32, 8), (8, 1)], ids=["llama3_8b", "llama3_70b_tp8_rank"])
def test_rope_and_store_kv_from_the_qkv_planes(ops, device, M, dtype, kv_dtype, heads):
    """Decode step: the qkv GEMM stopped before its K-slice reduction + ONE kernel that sums the planes, rotates q / k
    and stores k / v has the bits of stream_linear followed by rope_and_store_kv (q, and the pool rows).  (The kernel
    spreads a token's items over 1 .. 7 workgroups by the batch size: M and the head counts walk through the shapes.)"""
    torch.manual_seed(M)
    (Hq, Hk), D, K = heads, 128, 4096
    x = torch.randn(M, K, device=device).to(dtype)
    w = (torch.randn((Hq + 2 * Hk) * D, K, device=device) * 0.03).to(dtype)
    pos = torch.randint(0, 4000, (M,), device=device, dtype=torch.int64)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j -> ij", torch.arange(4096, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(device)
    pool_dtype = kv_dtype or dtype
    loc = torch.randperm(200, device=device)[:M].to(torch.int64)

    def pools():
        return (torch.zeros(200, Hk, D, device=device).to(pool_dtype), torch.zeros(200, Hk, D, device=device).to(pool_dtype))

    kb1, vb1 = pools()
    qkv = ops.stream_linear(x, w)
    q1, k1, v1 = qkv.split([Hq * D, Hk * D, Hk * D], dim=-1)
    ops.rope_and_store_kv(pos, q1, k1, v1, D, cache, True, kb1, vb1, loc)
    kb2, vb2 = pools()
    planes = ops.stream_linear_planes(x, w)
    assert planes.ksplit > 1
    q2 = ops.rope_and_store_kv_planes(pos, planes, Hq, Hk, D, cache, kb2, vb2, loc)
    assert torch.equal(q2.view(torch.int16), q1.contiguous().view(torch.int16))
    assert torch.equal(kb2.view(torch.uint8), kb1.view(torch.uint8)) and torch.equal(vb2.view(torch.uint8), vb1.view(torch.uint8))


FUSED_DECODE_LENS = [
    [5, 33, 700],
    [1, 2, 3, 8, 9, 31, 32, 33, 34, 64, 255, 256, 257, 258, 1000],   # the new token alone, in a wave's first tile, on tile edges
    [2048, 1100],
]


@pytest.mark.parametrize("lens", FUSED_DECODE_LENS, ids=["three", "edges", "long"])
@pytest.mark.parametrize("heads", [(32, 8, 128, 4096), (8, 1, 128, 8192), (12, 6, 64, 768), (32, 2, 64, 2048)],
                         ids=["llama3_8b", "llama3_70b_tp8_rank", "g2_d64", "g16_d64"])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype,kv_dtype,cap", [(torch.bfloat16, None, 0.0), (torch.float16, None, 30.0),
                                                 (torch.bfloat16, torch.float8_e4m3fn, 0.0)],
                         ids=["bf16", "f16_cap", "bf16_fp8kv"])
def test_decode_rope_attention_planes_has_the_bits_of_the_three_launches(ops, device, lens, heads, waves, dtype, kv_dtype, cap):
    """Decode step between the qkv GEMM and o_proj in ONE launch (csrc/decode_attention_fused.hip: plane sum + RoPE + KV
    store + paged attention + merge of the kv splits inside the workgroup) against the launches it replaces --
    rope_and_store_kv_planes, then decode_attention_fwd with as many splits as the fused kernel has waves: same output
    bits, same pool rows -- and against the oracle's attention over the pool it left behind.  The new token is the last
    row of every request (layers/attention/triton_backend.py:96-118: kv_indices of a decode batch end at out_cache_loc)."""
    Hq, Hk, D, K = heads
    B = len(lens)
    assert ops.decode_rope_attention_planes_supported(Hq, Hk, D, dtype, kv_dtype or dtype)
    g = torch.Generator().manual_seed(sum(lens) + Hq + waves)
    total = int(sum(lens))
    slots = total + 13
    perm = (torch.randperm(slots - 1, generator=g)[:total] + 1).to(torch.int32)
    indptr = torch.zeros(B + 1, dtype=torch.int32)
    indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    loc = perm[(indptr[1:] - 1).long()].to(torch.int64)            # each request's last row = where the new token goes
    pool_dtype = kv_dtype or dtype
    k0 = torch.randn(slots, Hk, D, generator=g).to(dtype).to(pool_dtype)
    v0 = torch.randn(slots, Hk, D, generator=g).to(dtype).to(pool_dtype)
    x = torch.randn(B, K, generator=g).to(dtype).to(device)
    w = (torch.randn((Hq + 2 * Hk) * D, K, generator=g) * 0.03).to(dtype).to(device)
    pos = (torch.tensor(lens) - 1).to(torch.int64).to(device)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j -> ij", torch.arange(4096, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(device)
    sm_scale = 1.0 / (D ** 0.5)
    indptr_d, perm_d, loc_d = indptr.to(device), perm.to(device), loc.to(device)

    kb1, vb1 = k0.clone().to(device), v0.clone().to(device)
    planes = ops.stream_linear_planes(x, w)
    q1 = ops.rope_and_store_kv_planes(pos, planes, Hq, Hk, D, cache, kb1, vb1, loc_d)
    o1 = torch.empty(B, Hq, D, dtype=dtype, device=device)
    logits = torch.empty(B, Hq, waves, D + 1, dtype=torch.float32, device=device)
    ops.decode_attention_fwd(q1.view(B, Hq, D), kb1, vb1, o1, indptr_d, perm_d, logits, waves, sm_scale, cap)

    kb2, vb2 = k0.clone().to(device), v0.clone().to(device)
    planes = ops.stream_linear_planes(x, w)
    o2 = ops.decode_rope_attention_planes(pos, planes, Hq, Hk, D, cache, kb2, vb2, loc_d, indptr_d, perm_d, waves, sm_scale, cap)
    torch.cuda.synchronize()
    assert torch.equal(kb2.view(torch.uint8), kb1.view(torch.uint8)) and torch.equal(vb2.view(torch.uint8), vb1.view(torch.uint8))
    assert torch.equal(o2.view(torch.int16), o1.view(B, Hq * D).view(torch.int16))
    want = O.decode_attention(q1.view(B, Hq, D).cpu(), kb2.cpu().to(dtype), vb2.cpu().to(dtype), indptr, perm, sm_scale, cap)
    _close(o2.view(B, Hq, D), want, dtype, rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)


@pytest.mark.parametrize("lens", FUSED_DECODE_LENS + [[7], [3000, 40, 129]], ids=["three", "edges", "long", "one", "ragged"])
@pytest.mark.parametrize("heads", [(32, 8, 128, 4096), (8, 1, 128, 8192), (32, 2, 64, 2048)],
                         ids=["llama3_8b", "llama3_70b_tp8_rank", "g16_d64"])
@pytest.mark.parametrize("zsplits", [2, 3, 4])
@pytest.mark.parametrize("dtype,kv_dtype", [(torch.bfloat16, None), (torch.float16, torch.float8_e5m2)], ids=["bf16", "f16_fp8kv"])
def test_decode_rope_attention_planes_with_several_workgroups_per_pair(ops, device, lens, heads, zsplits, dtype, kv_dtype):
    """Small decode batches: `zsplits` workgroups per (request, kv head), each merging its 8 kv splits into one stage-1
    partial, stage 2 behind them (two launches where the separate form takes three).  Pool rows: the bits of
    rope_and_store_kv_planes; output: the oracle's attention over the pool it left behind, and the separate launches' result
    to the same tolerance (the hierarchical merge rounds differently).  More splits than tokens, one-token requests and
    requests whose new token sits in the first / last workgroup are in the length lists."""
    Hq, Hk, D, K = heads
    B = len(lens)
    g = torch.Generator().manual_seed(sum(lens) + Hq + zsplits)
    total = int(sum(lens))
    slots = total + 13
    perm = (torch.randperm(slots - 1, generator=g)[:total] + 1).to(torch.int32)
    indptr = torch.zeros(B + 1, dtype=torch.int32)
    indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    loc = perm[(indptr[1:] - 1).long()].to(torch.int64)
    pool_dtype = kv_dtype or dtype
    k0 = torch.randn(slots, Hk, D, generator=g).to(dtype).to(pool_dtype)
    v0 = torch.randn(slots, Hk, D, generator=g).to(dtype).to(pool_dtype)
    x = torch.randn(B, K, generator=g).to(dtype).to(device)
    w = (torch.randn((Hq + 2 * Hk) * D, K, generator=g) * 0.03).to(dtype).to(device)
    pos = (torch.tensor(lens) - 1).to(torch.int64).to(device)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j -> ij", torch.arange(4096, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(device)
    sm_scale = 1.0 / (D ** 0.5)
    indptr_d, perm_d, loc_d = indptr.to(device), perm.to(device), loc.to(device)

    kb1, vb1 = k0.clone().to(device), v0.clone().to(device)
    q1 = ops.rope_and_store_kv_planes(pos, ops.stream_linear_planes(x, w), Hq, Hk, D, cache, kb1, vb1, loc_d)
    o1 = torch.empty(B, Hq, D, dtype=dtype, device=device)
    lg1 = torch.empty(B, Hq, 8 * zsplits, D + 1, dtype=torch.float32, device=device)
    ops.decode_attention_fwd(q1.view(B, Hq, D), kb1, vb1, o1, indptr_d, perm_d, lg1, 8 * zsplits, sm_scale)

    kb2, vb2 = k0.clone().to(device), v0.clone().to(device)
    lg2 = torch.full((B, Hq, zsplits, D + 1), float("nan"), dtype=torch.float32, device=device)   # every partial must be written
    o2 = ops.decode_rope_attention_planes(pos, ops.stream_linear_planes(x, w), Hq, Hk, D, cache, kb2, vb2, loc_d, indptr_d,
                                          perm_d, 8, sm_scale, zsplits=zsplits, attn_logits=lg2)
    torch.cuda.synchronize()
    assert torch.equal(kb2.view(torch.uint8), kb1.view(torch.uint8)) and torch.equal(vb2.view(torch.uint8), vb1.view(torch.uint8))
    assert not torch.isnan(lg2[..., :D]).any() and not torch.isnan(o2.float()).any()
    tol = dict(rtol=2e-2, atol=4e-3 if dtype == torch.float16 else 1.5e-2)
    want = O.decode_attention(q1.view(B, Hq, D).cpu(), kb2.cpu().to(dtype), vb2.cpu().to(dtype), indptr, perm, sm_scale)
    _close(o2.view(B, Hq, D), want, dtype, **tol)
    _close(o2.view(B, Hq, D), o1.float(), dtype, **tol)
    with pytest.raises(RuntimeError, match="attn_logits"):
        ops.decode_rope_attention_planes(pos, ops.stream_linear_planes(x, w), Hq, Hk, D, cache, kb2, vb2, loc_d, indptr_d, perm_d,
                                         8, sm_scale, zsplits=zsplits)


def test_decode_rope_attention_planes_refuses_what_it_cannot_do(ops, device):
    assert not ops.decode_rope_attention_planes_supported(40, 1, 128, torch.bfloat16, torch.bfloat16)    # 40 q heads per kv head
    assert not ops.decode_rope_attention_planes_supported(12, 12, 64, torch.bfloat16, torch.bfloat16)    # MHA: the shuffle kernel's
    assert not ops.decode_rope_attention_planes_supported(32, 8, 96, torch.bfloat16, torch.bfloat16)     # head size
    assert not ops.decode_rope_attention_planes_supported(32, 8, 128, torch.float32, torch.float32)
    Hq, Hk, D, K, B = 8, 2, 128, 1024, 2
    x = torch.randn(B, K, device=device).to(torch.bfloat16)
    w = (torch.randn((Hq + 2 * Hk) * D, K, device=device) * 0.03).to(torch.bfloat16)
    kb, vb = (torch.zeros(16, Hk, D, dtype=torch.bfloat16, device=device) for _ in range(2))
    cache = torch.zeros(64, D, device=device)
    pos = torch.zeros(B, dtype=torch.int64, device=device)
    loc = torch.tensor([1, 2], dtype=torch.int64, device=device)
    indptr = torch.tensor([0, 1, 2], dtype=torch.int32, device=device)
    idx = torch.tensor([1, 2], dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="waves"):
        ops.decode_rope_attention_planes(pos, ops.stream_linear_planes(x, w), Hq, Hk, D, cache, kb, vb, loc, indptr, idx, 3, 0.1)


# --------------------------------------------------------------------------- prefill-sized dense layers on a CU share
def test_dense_gemm_with_measured_library_solution_matches_fp32(ops, device):
    """ops.dense_gemm = F.linear (UnquantizedLinearMethod.apply, layers/linear.py:165-172) through the hipBLASLt solution
    that was timed fastest on this process's CUs: whatever solution wins, the result is x @ W^T (+ bias) in fp32
    accumulation, one rounding -- rows at, between and far from the tuned row counts, a strided x, f16 and bf16."""
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(31)
    for dtype, tol in ((torch.bfloat16, 2e-2), (torch.float16, 4e-3)):
        N, K = 768, 512
        w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).to(device)
        b = (torch.randn(N, generator=g) * 0.1).to(dtype).to(device)
        assert not ops.dense_gemm_is_tuned(w) or dtype == torch.float16
        ops.dense_gemm_tune(N, K, [256, 1024], dtype, num_full_search=1, num_heuristics=16, max_solutions=24)
        assert ops.dense_gemm_is_tuned(w)
        report = ops.dense_gemm_report()
        assert f"n={N} k={K} rows=256" in report and f"n={N} k={K} rows=1024" in report
        for rows in (1, 65, 256, 300, 1024, 3000):
            xw = (torch.randn(rows, K + 64, generator=g)).to(dtype).to(device)
            for x in (xw[:, :K].contiguous(), xw[:, 32:32 + K]):   # the second one: row stride K + 64
                for bias in (None, b):
                    got = ops.dense_gemm(x, w, bias)
                    want = x.float() @ w.float().t() + (bias.float() if bias is not None else 0.0)
                    torch.testing.assert_close(got.float(), want, rtol=tol, atol=tol)
                    # and it is the library's arithmetic: same bits as F.linear up to the solution's summation order
                    torch.testing.assert_close(got.float(), F.linear(x, w, bias).float(), rtol=tol, atol=tol)
    with pytest.raises(RuntimeError):
        ops.dense_gemm(torch.zeros(4, 8, device=device), torch.zeros(4, 8, device=device))   # fp32: not this path
    # the table as a start-up cache: what the report prints is what the import reads; filed under another CU count
    # (semipd_dense_gemm_set_cus: an instance on another stream) it is found there and nowhere else, and the product of a
    # re-imported solution is still the product
    from semi_pd_amd import _lib
    lib = _lib.load()
    report = ops.dense_gemm_report()
    mine = "".join(ln + "\n" for ln in report.splitlines() if " n=768 k=512 " in ln)
    assert mine.count("\n") == 4                                  # two row counts x two dtypes
    import re
    cur = int(re.match(r"cus=(\d+) ", mine).group(1))      # whatever share this process last declared
    moved = re.sub(r"^cus=\d+ ", "cus=77 ", mine, flags=re.M)
    assert ops.dense_gemm_import(moved + "garbage line\ncus=77 dtype=1 n=768 k=512 rows=64 solution=-3 us=1 library_choice_us=1 "
                                 "candidates=1 wrong_results_rejected=0\n"
                                 "cus=77 dtype=1 n=1024 k=512 rows=64 solution=-3 us=1 library_choice_us=1 candidates=1 "
                                 "wrong_results_rejected=0\n", [(768, 512, torch.bfloat16), (1024, 512, torch.bfloat16)]) == 4
    assert "cus=77 " in ops.dense_gemm_report()
    # tuned = what the C side took: the shape whose only line it refused is NOT routed to ops.dense_gemm, and the
    # f16 lines that were taken do not mark a dtype nobody asked for
    from semi_pd_amd.ops import _DENSE_GEMM
    assert (77, 768, 512, torch.bfloat16) in _DENSE_GEMM["tuned"] and (77, 1024, 512, torch.bfloat16) not in _DENSE_GEMM["tuned"]
    assert (77, 768, 512, torch.float16) not in _DENSE_GEMM["tuned"]
    try:
        _lib.check(lib.semipd_dense_gemm_set_cus(77), "set_cus")
        x = torch.randn(256, 512, generator=g).to(torch.bfloat16).to(device)
        w = (torch.randn(768, 512, generator=g) * 0.05).to(torch.bfloat16).to(device)
        torch.testing.assert_close(ops.dense_gemm(x, w).float(), x.float() @ w.float().t(), rtol=2e-2, atol=2e-2)
    finally:
        _lib.check(lib.semipd_dense_gemm_set_cus(cur), "set_cus")


# --------------------------------------------------------------------------- tall decode batches: tiled ping-pong GEMM
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(65, 256, 128), (256, 512, 512), (200, 4096, 4096), (96, 1024, 14336), (256, 6144, 4096),
                                   (300, 768, 256), (1024, 1280, 1024), (129, 272, 64), (1, 16, 64)])
def test_gemm_tall_matches_fp32(ops, device, dtype, M, N, K):
    """ops.gemm_tall = F.linear for 65-row and taller batches (layers/linear.py:165-172): fp32 accumulation, one rounding.
    Every K split the kernel may pick (SEMIPD_G8_KS forces 1, 2 and 4) must agree with the fp32 product within the
    rounding of the output type; shapes cover ragged M / N tails, several row tiles, an odd number of K steps."""
    import os
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
    x = (torch.randn(M, K, generator=g)).to(dtype).to(device)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).to(device)
    want = x.float() @ w.float().t()
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    try:
        for geo in ("0", "64", "128"):       # 0 = by row count; 64 = the 128 x 512 tile, 128 = the 256 x 256 tile
            os.environ["SEMIPD_G8_XH"] = geo
            for ks in ("0", "1", "2", "4"):
                os.environ["SEMIPD_G8_KS"] = ks
                got = ops.gemm_tall(x, w)
                torch.testing.assert_close(got.float(), want, rtol=tol, atol=tol)
    finally:
        os.environ.pop("SEMIPD_G8_KS", None)
        os.environ.pop("SEMIPD_G8_XH", None)
    # a strided x (rows of a wider buffer) and a caller-provided out with a row stride
    xw = torch.zeros(M, K + 64, dtype=dtype, device=device)
    xw[:, 16:16 + K] = x
    buf = torch.full((M, N + 8), 7.0, dtype=dtype, device=device)
    ops.gemm_tall(xw[:, 16:16 + K], w, out=buf[:, :N])
    torch.testing.assert_close(buf[:, :N].float(), want, rtol=tol, atol=tol)
    assert float(buf[:, N:].float().min()) == 7.0 and float(buf[:, N:].float().max()) == 7.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,I,K", [(65, 128, 128), (256, 1408, 2048), (130, 14336, 4096), (256, 176, 192)])
def test_gemm_tall_silu_mul_equals_the_unfused_pair(ops, device, dtype, M, I, K):
    """gate_up GEMM with the SiLU * mul epilogue (models/llama.py:88-92): the bits of act_fn(gate_up_proj(x)) with the
    GEMM output rounded to the activation type first -- compared with ops.silu_and_mul of the kernel's own plain output
    (exact) and with the oracle's fused form on fp32 products (one ulp of slack for the summation order)."""
    import os
    g = torch.Generator(device="cpu").manual_seed(M + I + K)
    x = (torch.randn(M, K, generator=g)).to(dtype).to(device)
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5).to(dtype).to(device)
    try:
        for geo in ("64", "128"):
            os.environ["SEMIPD_G8_XH"] = geo
            for ks in ("1", "2"):
                os.environ["SEMIPD_G8_KS"] = ks
                fused = ops.gemm_tall(x, w, fuse_silu_mul=True)
                plain = ops.gemm_tall(x, w)
                assert torch.equal(fused, ops.silu_and_mul(plain))
    finally:
        os.environ.pop("SEMIPD_G8_KS", None)
        os.environ.pop("SEMIPD_G8_XH", None)
    want = O.silu_and_mul((x.float().cpu() @ w.float().cpu().t()).to(dtype))
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    torch.testing.assert_close(fused.float().cpu(), want.float(), rtol=tol, atol=tol)


@pytest.fixture
def four_wave_form(ops):
    """ops.gemm_tall's 256 x 256 tiles on the 4-wave kernel (csrc/gemm8p.hip: gemm4w_kernel) for the duration of a test."""
    import os
    ops.gemm_tall_set_form(4)
    os.environ["SEMIPD_G8_XH"] = "128"         # the 256 x 256 geometry whatever the row count
    try:
        yield
    finally:
        ops.gemm_tall_set_form(0)
        os.environ.pop("SEMIPD_G8_XH", None)
        os.environ.pop("SEMIPD_G8_KS", None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (200, 4096, 4096), (96, 1024, 14336), (300, 768, 256), (1024, 1280, 1024),
                                   (1411, 6144, 4096), (129, 272, 64), (1, 16, 128), (513, 272, 3200)])
def test_gemm_tall_four_wave_form_matches_fp32_and_the_eight_wave_form(ops, device, four_wave_form, dtype, M, N, K):
    """The 4-wave form of the tiled GEMM (one 128 x 128 quarter of the 256 x 256 tile per wave, accumulators in the
    accumulation registers, round 6) = F.linear (layers/linear.py:165-172): against the fp32 product for every K split, and
    -- without a split -- the BITS of the 8-wave form (the same MFMAs in the same order per output element).  K = 64 (an odd
    count of K steps) and the odd slices a split would give stay on the 8-wave kernel: covered by falling through."""
    import os
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
    x = (torch.randn(M, K, generator=g)).to(dtype).to(device)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).to(device)
    want = x.float() @ w.float().t()
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    for ks in ("0", "1", "2", "4"):
        os.environ["SEMIPD_G8_KS"] = ks
        got = ops.gemm_tall(x, w)
        torch.testing.assert_close(got.float(), want, rtol=tol, atol=tol)
    os.environ["SEMIPD_G8_KS"] = "1"
    four = ops.gemm_tall(x, w)
    ops.gemm_tall_set_form(8)
    assert torch.equal(four, ops.gemm_tall(x, w))
    ops.gemm_tall_set_form(4)
    # strided x, caller-provided out with a row stride
    xw = torch.zeros(M, K + 64, dtype=dtype, device=device)
    xw[:, 16:16 + K] = x
    buf = torch.full((M, N + 8), 7.0, dtype=dtype, device=device)
    ops.gemm_tall(xw[:, 16:16 + K], w, out=buf[:, :N])
    assert torch.equal(buf[:, :N], four)
    assert float(buf[:, N:].float().min()) == 7.0 and float(buf[:, N:].float().max()) == 7.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,I,K", [(256, 1408, 2048), (130, 14336, 4096), (256, 176, 256), (1100, 2816, 1024)])
def test_gemm_tall_four_wave_form_silu_mul_equals_the_unfused_pair(ops, device, four_wave_form, dtype, M, I, K):
    import os
    g = torch.Generator(device="cpu").manual_seed(M + I + K)
    x = (torch.randn(M, K, generator=g)).to(dtype).to(device)
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5).to(dtype).to(device)
    for ks in ("1", "2"):
        os.environ["SEMIPD_G8_KS"] = ks
        fused = ops.gemm_tall(x, w, fuse_silu_mul=True)
        plain = ops.gemm_tall(x, w)
        assert torch.equal(fused, ops.silu_and_mul(plain))
    want = O.silu_and_mul((x.float().cpu() @ w.float().cpu().t()).to(dtype))
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    torch.testing.assert_close(fused.float().cpu(), want.float(), rtol=tol, atol=tol)


@pytest.mark.parametrize("M,N,K,silu", [(256, 4096, 4096, False), (1024, 2048, 2048, False), (700, 1024, 512, True), (1411, 4096, 14336, False)])
def test_gemm_tall_four_wave_form_race_screen(ops, device, four_wave_form, M, N, K, silu):
    """The race screen of the 8-wave kernel (below) for the 4-wave form: its half-tile slots are re-filled ONE phase after
    their last read and read eight phases later, on hand-counted vmcnt(24) waits -- 150 launches next to a stream that
    saturates HBM must all give the bits of the first one, and that one the fp32 product."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(device)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(device)
    want = x.float() @ w.float().t()
    if silu:
        want = O.silu_and_mul(want.to(torch.bfloat16).cpu()).float().to(device)
    first = ops.gemm_tall(x, w, fuse_silu_mul=silu).clone()
    torch.testing.assert_close(first.float(), want, rtol=3e-2, atol=3e-2)
    junk_a = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    junk_b = torch.empty_like(junk_a)
    side = torch.cuda.Stream(device=device)
    outs = []
    for i in range(150):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk_b.copy_(junk_a)
                junk_a.copy_(junk_b)
        outs.append(ops.gemm_tall(x, w, fuse_silu_mul=silu))
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, first)]
    assert not bad, f"launches {bad[:8]} of 150 differ from the first one"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,E,topk,K,N", [(700, 8, 2, 256, 192), (1500, 16, 6, 2048, 1408), (300, 4, 1, 128, 64)])
@pytest.mark.parametrize("bm", [256, 128])
def test_moe_gemm_tall_matches_fp32_per_expert(ops, device, dtype, T, E, topk, K, N, bm):
    """The grouped form of the tiled kernel on 256-row blocks (256 x 256 tiles) and on 128-row blocks (128 rows x 512
    columns) (invoke_fused_moe_kernel, fused_moe.py:501-612): every routed
    entry id gets a[id // top_k] @ w[expert(id)]^T -- checked row by row against fp32 products; GEMM1 with the SiLU * mul
    epilogue equals silu_and_mul of its own plain output bit for bit; GEMM2 multiplies by the routed weight before the
    rounding; padding entries (>= num_valid) write nothing (c is pre-filled with a sentinel)."""
    g = torch.Generator(device="cpu").manual_seed(T + E + K)
    a = torch.randn(T, K, generator=g).to(dtype).to(device)
    w1 = (torch.randn(E, 2 * N, K, generator=g) * K ** -0.5).to(dtype).to(device)
    w2 = (torch.randn(E, K, N, generator=g) * N ** -0.5).to(dtype).to(device)
    logits = torch.randn(T, E, generator=g).to(device)
    tw, ti = ops.topk_softmax(logits, topk, True)
    numel = T * topk
    max_sorted = -(-(numel + E * (bm - 1)) // bm) * bm
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
    expert_ids = torch.empty(max_sorted // bm, dtype=torch.int32, device=device)
    npp = torch.empty(1, dtype=torch.int32, device=device)
    ops.moe_align_block_size(ti, E, bm, sorted_ids, expert_ids, npp, None, torch.empty(E + 1, dtype=torch.int32, device=device))
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    flat_e = ti.reshape(-1).long()
    rows = torch.arange(numel, device=device) // topk
    # GEMM1 plain and fused
    c1 = torch.full((numel, 2 * N), 77.0, dtype=dtype, device=device)
    ops.moe_gemm_tall(a, w1, c1, None, sorted_ids, expert_ids, npp, numel, topk, False, False, block_m=bm)
    def per_expert(x_rows, w, scale=None):       # fp32 reference, one expert at a time (a gather of w would not fit)
        want = torch.empty(numel, w.shape[1], dtype=torch.float32, device=device)
        for e in range(E):
            sel = (flat_e == e).nonzero().squeeze(1)
            if sel.numel():
                want[sel] = x_rows[sel].float() @ w[e].float().t()
        return want if scale is None else want * scale

    torch.testing.assert_close(c1.float(), per_expert(a[rows], w1), rtol=tol, atol=tol)
    c2 = torch.full((numel, N), 77.0, dtype=dtype, device=device)
    ops.moe_gemm_tall(a, w1, c2, None, sorted_ids, expert_ids, npp, numel, topk, False, True, block_m=bm)
    assert torch.equal(c2, ops.silu_and_mul(c1))
    # GEMM2 with the routed weight
    c3 = torch.full((numel, K), 77.0, dtype=dtype, device=device)
    ops.moe_gemm_tall(c2, w2, c3, tw.reshape(-1), sorted_ids, expert_ids, npp, numel, 1, True, False, block_m=bm)
    torch.testing.assert_close(c3.float(), per_expert(c2, w2, tw.reshape(-1, 1).float()), rtol=tol, atol=tol)


def test_fused_experts_takes_the_tall_kernel_for_prefill_sized_calls(ops, device, monkeypatch):
    """layers.moe.fused_experts above the row thresholds (256-row blocks, grouped ping-pong GEMM) against the same call
    below them (128-row blocks, the round-2 kernels) and the oracle's fused MoE."""
    from semi_pd_amd.layers import moe as M
    g = torch.Generator(device="cpu").manual_seed(77)
    T, E, k, K, N = 1024, 8, 2, 512, 384
    x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(device)
    w1 = (torch.randn(E, 2 * N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(device)
    w2 = (torch.randn(E, K, N, generator=g) * N ** -0.5).to(torch.bfloat16).to(device)
    tw, ti = ops.topk_softmax(torch.randn(T, E, generator=g).to(device), k, True)
    calls = []
    real = ops.moe_gemm_tall
    monkeypatch.setattr(ops, "moe_gemm_tall", lambda *a, **kw: (calls.append(kw.get("block_m")), real(*a, **kw))[1])
    monkeypatch.setattr(M, "MOE_TALL_MIN_ROWS", 1 << 30)
    monkeypatch.setattr(M, "MOE_MID_MIN_ROWS_PER_EXPERT", 1 << 30)
    base = M.fused_experts(x, w1, w2, tw, ti)
    assert not calls
    want = O.fused_moe(x.float().cpu(), w1.float().cpu(), w2.float().cpu(), tw.cpu(), ti.cpu().long())
    # 256 rows per expert here: the 256-row geometry above its bounds, the 128-row one between its bound and those
    for tall_min, mid_min, block in ((1024, 1 << 30, 256), (1 << 30, 40, 128)):
        monkeypatch.setattr(M, "MOE_TALL_MIN_ROWS", tall_min)
        monkeypatch.setattr(M, "MOE_TALL_MIN_ROWS_PER_EXPERT", 128)
        monkeypatch.setattr(M, "MOE_MID_MIN_ROWS_PER_EXPERT", mid_min)
        del calls[:]
        tall = M.fused_experts(x, w1, w2, tw, ti)
        assert calls == [block, block]
        torch.testing.assert_close(tall.float().cpu(), want.float(), rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(tall.float(), base.float(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,E,topk,K,N", [(1, 8, 2, 256, 128), (17, 16, 6, 2048, 1408), (33, 64, 6, 2048, 1408), (64, 8, 2, 384, 128),
                                          (48, 4, 1, 128, 64)])
def test_moe_stream_gemm_matches_fp32_per_expert(ops, device, dtype, T, E, topk, K, N):
    """The grouped form of the LDS-DMA streaming kernel (decode-sized invoke_fused_moe_kernel, fused_moe.py:501-612) on
    blocks of 16 ceil(T / 16) rows: every routed entry against fp32 products of its expert; GEMM1 with the SiLU * mul
    epilogue equals silu_and_mul of the plain output bit for bit; GEMM2 times the routed weight; padding entries write
    nothing; and layers.moe.fused_experts (which takes this path up to 64 tokens) agrees with the path it replaces."""
    g = torch.Generator(device="cpu").manual_seed(T * 3 + E + K)
    a = torch.randn(T, K, generator=g).to(dtype).to(device)
    w1 = (torch.randn(E, 2 * N, K, generator=g) * K ** -0.5).to(dtype).to(device)
    w2 = (torch.randn(E, K, N, generator=g) * N ** -0.5).to(dtype).to(device)
    tw, ti = ops.topk_softmax(torch.randn(T, E, generator=g).to(device), topk, True)
    numel, bm = T * topk, 16 * -(-T // 16)
    max_sorted = -(-(numel + E * (bm - 1)) // bm) * bm
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=device)
    expert_ids = torch.empty(max_sorted // bm, dtype=torch.int32, device=device)
    npp = torch.empty(1, dtype=torch.int32, device=device)
    ops.moe_align_block_size(ti, E, bm, sorted_ids, expert_ids, npp, None, torch.empty(E + 1, dtype=torch.int32, device=device))
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    flat_e = ti.reshape(-1).long()
    rows = torch.arange(numel, device=device) // topk

    def per_expert(x_rows, w, scale=None):
        want = torch.empty(numel, w.shape[1], dtype=torch.float32, device=device)
        for e in range(E):
            sel = (flat_e == e).nonzero().squeeze(1)
            if sel.numel():
                want[sel] = x_rows[sel].float() @ w[e].float().t()
        return want if scale is None else want * scale

    if N * 2 % 16 == 0 and K % 128 == 0:
        c1 = torch.full((numel, 2 * N), 77.0, dtype=dtype, device=device)
        ops.moe_stream_gemm(a, w1, c1, None, sorted_ids, expert_ids, npp, numel, topk, False, bm, False)
        torch.testing.assert_close(c1.float(), per_expert(a[rows], w1), rtol=tol, atol=tol)
        c2 = torch.full((numel, N), 77.0, dtype=dtype, device=device)
        ops.moe_stream_gemm(a, w1, c2, None, sorted_ids, expert_ids, npp, numel, topk, False, bm, True)
        assert torch.equal(c2, ops.silu_and_mul(c1))
        if N % 128 == 0:
            c3 = torch.full((numel, K), 77.0, dtype=dtype, device=device)
            ops.moe_stream_gemm(c2, w2, c3, tw.reshape(-1), sorted_ids, expert_ids, npp, numel, 1, True, bm, False)
            torch.testing.assert_close(c3.float(), per_expert(c2, w2, tw.reshape(-1, 1).float()), rtol=tol, atol=tol)
    from semi_pd_amd.layers import moe as M
    new = M.fused_experts(a, w1, w2, tw, ti)
    old_flag, M.MOE_STREAM_DECODE = M.MOE_STREAM_DECODE, False
    try:
        old = M.fused_experts(a, w1, w2, tw, ti)
    finally:
        M.MOE_STREAM_DECODE = old_flag
    torch.testing.assert_close(new.float(), old.float(), rtol=3e-2, atol=3e-2)


def test_logits_processor_above_64_rows_reads_the_head_once(ops, device):
    """_get_logits (layers/logits_processor.py:394-445) for a decode batch above the fused kernel's 64 rows: the tiled
    ping-pong GEMM on the (padded) vocabulary-sized weight, logits in the activation type then fp32 -- equal to the
    reference form torch.matmul(hidden, weight.T) within its rounding, same argmax wherever the top-2 gap is clear."""
    from types import SimpleNamespace as NS
    from semi_pd_amd.layers.basic import LogitsProcessor
    from semi_pd_amd.model_executor.forward_batch_info import ForwardMode
    g = torch.Generator(device="cpu").manual_seed(5)
    V, Vpad, H, B = 5000, 5056, 512, 100
    weight = (torch.randn(Vpad, H, generator=g) * 0.05).to(torch.bfloat16).to(device)
    hidden = torch.randn(B, H, generator=g).to(torch.bfloat16).to(device)
    lp = LogitsProcessor(V)
    fb = NS(forward_mode=ForwardMode.DECODE)
    out = lp(None, hidden, NS(weight=weight), fb)
    want = (hidden.float() @ weight[:V].float().t())
    assert out.next_token_logits.shape == (B, V) and out.next_token_logits.dtype == torch.float32
    torch.testing.assert_close(out.next_token_logits, want, rtol=2e-2, atol=2e-2)
    top2 = torch.topk(want, 2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 4e-2
    assert torch.equal(out.next_token_logits.argmax(-1)[clear], want.argmax(-1)[clear])


@pytest.mark.parametrize("M,N,K,silu", [(256, 4096, 4096, False), (128, 2048, 4096, True), (1024, 2048, 2048, False), (700, 1024, 512, True)])
def test_gemm_tall_race_screen_repeated_launches_under_memory_load(ops, device, M, N, K, silu):
    """The ping-pong GEMM orders its LDS-DMA by counted waits and barriers; a hole in that ordering shows as a rare wrong
    tile that comes and goes with timing.  The kernel is deterministic (fixed summation order), so: 150 launches on the
    same operands, next to a second stream that saturates HBM with copies, must all give the bits of the first launch
    -- and that launch the fp32 product."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(device)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(device)
    want = x.float() @ w.float().t()
    if silu:
        want = O.silu_and_mul(want.to(torch.bfloat16).cpu()).float().to(device)
    first = ops.gemm_tall(x, w, fuse_silu_mul=silu).clone()
    torch.testing.assert_close(first.float(), want, rtol=3e-2, atol=3e-2)
    junk_a = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    junk_b = torch.empty_like(junk_a)
    side = torch.cuda.Stream(device=device)
    outs = []
    for i in range(150):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk_b.copy_(junk_a)
                junk_a.copy_(junk_b)
        outs.append(ops.gemm_tall(x, w, fuse_silu_mul=silu))
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, first)]
    assert not bad, f"launches {bad[:8]} of 150 differ from the first one"


def test_grouped_kernels_race_screen_repeated_launches(ops, device):
    """The same screen for the grouped forms (fused-MoE expert GEMMs): decode-sized through the LDS-DMA streaming kernel,
    prefill-sized through the ping-pong tile kernel, 100 launches each next to a stream of HBM copies."""
    from semi_pd_amd.layers import moe as M
    g = torch.Generator(device="cpu").manual_seed(9)
    E, k, K, N = 16, 4, 1024, 512
    w1 = (torch.randn(E, 2 * N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(device)
    w2 = (torch.randn(E, K, N, generator=g) * N ** -0.5).to(torch.bfloat16).to(device)
    junk_a = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    junk_b = torch.empty_like(junk_a)
    side = torch.cuda.Stream(device=device)
    for T in (24, 4096):
        x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(device)
        tw, ti = ops.topk_softmax(torch.randn(T, E, generator=g).to(device), k, True)
        first = M.fused_experts(x, w1, w2, tw, ti).clone()
        outs = []
        for i in range(100):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    junk_b.copy_(junk_a)
            outs.append(M.fused_experts(x, w1, w2, tw, ti))
        torch.cuda.synchronize()
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, first)]
        assert not bad, f"T = {T}: launches {bad[:8]} of 100 differ from the first one"
