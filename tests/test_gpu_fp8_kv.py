"""fp8 KV cache (`--kv-cache-dtype fp8_e5m2 | fp8_e4m3`, reference: mem_cache/memory_pool.py:205-209,
326-336: rows are stored with `.to(fp8)` and read back with `.to(q.dtype)`; attention arithmetic stays
in the activation type).  The reference's conversion IS torch's, so torch on CPU is the oracle for the
bytes; the attention oracles run on the dequantised rows."""
import pytest
import torch

from oracle import ops as O
from oracle.model import OracleLlama

pytestmark = pytest.mark.gpu

F8 = [torch.float8_e5m2, torch.float8_e4m3fn]


@pytest.fixture(scope="module")
def ops():
    from semi_pd_amd import ops as _ops
    return _ops


def _bytes(t):
    return t.contiguous().view(torch.uint8)


@pytest.mark.parametrize("kv_dtype", F8)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,D", [(8, 128), (2, 64), (3, 20)])
def test_store_rows_matches_torch_conversion_bitwise(ops, device, kv_dtype, dtype, H, D):
    torch.manual_seed(H * D)
    T, N = 37, 64
    src = (torch.randn(T, H, D) * 3).to(dtype)
    src[0, 0, :8] = torch.tensor([0.0, -0.0, 1e-3, -1e-3, 1e-6, 0.5, -2.0, 7.75]).to(dtype)
    buf = torch.zeros(N, H, D, dtype=kv_dtype, device=device)
    loc = torch.randperm(N - 1)[:T] + 1
    ops.store_kv_rows(buf, loc.to(device), src.to(device))
    want = torch.zeros(N, H, D, dtype=kv_dtype)
    want[loc] = src.to(kv_dtype)
    assert torch.equal(_bytes(buf.cpu()), _bytes(want))


@pytest.mark.parametrize("kv_dtype", F8)
def test_rope_and_store_into_fp8_pool(ops, device, kv_dtype):
    torch.manual_seed(1)
    T, Hq, Hk, D, N = 19, 8, 2, 128, 40
    q = torch.randn(T, Hq * D).to(torch.bfloat16)
    k = torch.randn(T, Hk * D).to(torch.bfloat16)
    v = torch.randn(T, Hk * D).to(torch.bfloat16)
    pos = torch.randint(0, 200, (T,), dtype=torch.int64)
    cache = O.cos_sin_cache_from_inv_freq(O.rope_inv_freq(D, 10000.0), 256)
    kb = torch.zeros(N, Hk, D, dtype=kv_dtype, device=device)
    vb = torch.zeros(N, Hk, D, dtype=kv_dtype, device=device)
    loc = (torch.randperm(N - 1)[:T] + 1).to(torch.int64)
    qd, kd = q.to(device), k.to(device)
    ops.rope_and_store_kv(pos.to(device), qd, kd, v.to(device), D, cache.to(device), True, kb, vb, loc.to(device))
    # the rotated keys stay in the activation type in place; the pool holds exactly their fp8 rounding
    assert torch.equal(_bytes(kb.cpu()[loc]), _bytes(kd.cpu().view(T, Hk, D).to(kv_dtype)))
    assert torch.equal(_bytes(vb.cpu()[loc]), _bytes(v.view(T, Hk, D).to(kv_dtype)))
    qo, ko = O.apply_rope(pos, q, k, D, cache, True)
    torch.testing.assert_close(kd.cpu().float(), ko.float(), rtol=1e-2, atol=1e-2)


def _paged(B, lens, Hkv, D, kv_dtype, seed):
    g = torch.Generator().manual_seed(seed)
    total = sum(lens)
    N = total + 9
    k = torch.randn(N, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(N, Hkv, D, generator=g).to(torch.bfloat16)
    perm = torch.randperm(N - 1, generator=g)[:total] + 1
    indptr = torch.zeros(B + 1, dtype=torch.int32)
    indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    return k.to(kv_dtype), v.to(kv_dtype), indptr, perm.to(torch.int32)


@pytest.mark.parametrize("kv_dtype", F8)
@pytest.mark.parametrize("B,lens,Hq,Hkv,D,splits", [(3, [70, 1, 257], 32, 8, 128, 4), (2, [33, 500], 8, 1, 64, 8),
                                                     (1, [129], 12, 4, 96, 1),
                                                     # MHA (one query head per kv head: OPT, Llama-2-7B) and a head
                                                     # size without an MFMA instantiation: the shuffle kernel
                                                     (2, [33, 200], 4, 4, 64, 2), (1, [129], 12, 12, 128, 1),
                                                     (2, [70, 300], 8, 2, 80, 4), (1, [65], 2, 2, 256, 1)])
def test_decode_attention_fp8_pool(ops, device, kv_dtype, B, lens, Hq, Hkv, D, splits):
    k8, v8, indptr, idx = _paged(B, lens, Hkv, D, kv_dtype, B + D)
    q = torch.randn(B, Hq, D).to(torch.bfloat16)
    o = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=device)
    lg = torch.empty(B, Hq, splits, D + 1, dtype=torch.float32, device=device)
    ops.decode_attention_fwd(q.to(device), k8.to(device), v8.to(device), o, indptr.to(device), idx.to(device),
                             lg, splits, D ** -0.5)
    want = O.decode_attention(q, k8.to(torch.bfloat16), v8.to(torch.bfloat16), indptr, idx, D ** -0.5)
    torch.testing.assert_close(o.cpu().float(), want.float(), rtol=1.6e-2, atol=1e-2)


@pytest.mark.parametrize("kv_dtype", F8)
@pytest.mark.parametrize("pre,ext,Hq,Hkv,D", [([40, 0, 130], [17, 64, 200], 8, 2, 128), ([300], [5], 4, 4, 64)])
def test_extend_attention_fp8_prefix(ops, device, kv_dtype, pre, ext, Hq, Hkv, D):
    B = len(pre)
    k8, v8, kv_indptr, idx = _paged(B, pre, Hkv, D, kv_dtype, sum(ext))
    T = sum(ext)
    g = torch.Generator().manual_seed(T)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    qo = torch.zeros(B + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(torch.tensor(ext), 0)
    o = torch.empty(T, Hq, D, dtype=torch.bfloat16, device=device)
    ops.extend_attention_fwd(q.to(device), k.to(device), v.to(device), o, k8.to(device), v8.to(device),
                             qo.to(device), kv_indptr.to(device), idx.to(device), None, None, max(ext), D ** -0.5)
    want = O.extend_attention(q, k, v, k8.to(torch.bfloat16), v8.to(torch.bfloat16), qo, kv_indptr, idx, D ** -0.5)
    torch.testing.assert_close(o.cpu().float(), want.float(), rtol=1.6e-2, atol=1e-2)


def test_unsupported_fp8_paths_fail_loudly(ops, device):
    """Head sizes that are not a multiple of 8 only have the scalar kernel, which reads activation-type rows."""
    k8 = torch.zeros(10, 4, 20, dtype=torch.float8_e5m2, device=device)
    q = torch.zeros(1, 4, 20, dtype=torch.bfloat16, device=device)
    o = torch.empty_like(q)
    indptr = torch.tensor([0, 3], dtype=torch.int32, device=device)
    idx = torch.tensor([1, 2, 3], dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="fp8 KV"):
        ops.decode_attention_fwd(q, k8, k8, o, indptr, idx, None, 1, 0.1)


@pytest.mark.parametrize("kv", ["fp8_e5m2", "fp8_e4m3"])
def test_engine_with_fp8_kv_cache_matches_quantising_oracle(kv):
    """Unified and Semi-PD engines with an fp8 pool against the oracle model whose cache rounds rows
    the same way.  Prompts are prefilled in one piece (the oracle does the same), so the prompt tokens
    attend to each other unquantised and every decode step reads fp8 rows; fp8 prefix rows in prefill
    are covered at the op level above."""
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    from test_gpu_engine import check_against_oracle, make_prompts, server_args, tiny_llama
    cfg = tiny_llama()
    kvd = torch.float8_e5m2 if kv == "fp8_e5m2" else torch.float8_e4m3fn
    prompts = make_prompts(cfg.vocab_size, [5, 150, 64, 1, 90, 33], seed=5)
    sp = SamplingParams(max_new_tokens=10, ignore_eos=True)
    eng = Engine(server_args(cfg, kv_cache_dtype=kv))
    try:
        sd = {k: v.float().cpu() for k, v in eng.model_runner.model.state_dict().items()}
        assert eng.model_runner.token_to_kv_pool.k_buffer[0].dtype == kvd
        outs = eng.generate(prompts, sp)
    finally:
        eng.shutdown()
    oracle = OracleLlama(cfg, sd, kv_cache_dtype=kvd)
    # e5m2 keeps 2 mantissa bits: rounding the fp32 oracle's rows vs the engine's bf16 rows flips more
    # near-ties than the bf16 cache does, hence the wider margin
    check_against_oracle(oracle, prompts, outs, margin=0.15)
    eng = Engine(server_args(cfg, kv_cache_dtype=kv, enable_semi_pd=True, prefill_cu_percent=50,
                             decode_cu_percent=50))
    try:
        semi = eng.generate(prompts, sp, timeout=300)
    finally:
        eng.shutdown()
    check_against_oracle(oracle, prompts, semi, margin=0.15)


# ----------------------------------------------------------------------------------------------- MLA latent rows
def _mla_paged(B, lens, kv_dtype, seed):
    g = torch.Generator().manual_seed(seed)
    total = sum(lens)
    N = total + 5
    rows = torch.randn(N, 1, 576, generator=g).to(torch.bfloat16)
    perm = torch.randperm(N - 1, generator=g)[:total] + 1
    indptr = torch.zeros(B + 1, dtype=torch.int32)
    indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    return rows.to(kv_dtype), indptr, perm.to(torch.int32)


@pytest.mark.parametrize("kv_dtype", F8)
@pytest.mark.parametrize("B,lens,H,splits", [(3, [70, 1, 300], 16, 4), (2, [33, 129], 128, 1), (1, [500], 8, 8)])
def test_mla_decode_attention_fp8_latent_rows(ops, device, kv_dtype, B, lens, H, splits):
    """MLATokenToKVPool with an fp8 store dtype (memory_pool.py:439-452): the MLA decode kernel expands the 576-wide
    latent rows on their way into LDS; keys are the whole row, values its first 512 columns."""
    rows8, indptr, idx = _mla_paged(B, lens, kv_dtype, B + H)
    q = (torch.randn(B, H, 576) * 0.3).to(torch.bfloat16)
    o = torch.empty(B, H, 512, dtype=torch.bfloat16, device=device)
    lg = torch.empty(B, H, splits, 513, dtype=torch.float32, device=device)
    buf = rows8.to(device)
    ops.decode_attention_fwd(q.to(device), buf, buf[..., :512], o, indptr.to(device), idx.to(device), lg, splits, 576 ** -0.5)
    rows = rows8.to(torch.bfloat16)
    want = O.decode_attention(q, rows, rows[..., :512], indptr, idx, 576 ** -0.5)
    torch.testing.assert_close(o.cpu().float(), want.float(), rtol=1.6e-2, atol=1e-2)


@pytest.mark.parametrize("kv_dtype", F8)
def test_mla_extend_with_fp8_prefix_rows(ops, device, kv_dtype):
    """A chunked MLA prefill continues behind cached latent rows (forward_absorb + extend attention with a prefix):
    the one-wave-per-(token, head) kernel reads the pool in either storage type."""
    pre, ext, H = [40, 0, 97], [9, 20, 33], 8
    B = len(pre)
    rows8, kv_indptr, idx = _mla_paged(B, pre, kv_dtype, 11)
    T = sum(ext)
    g = torch.Generator().manual_seed(T)
    q = (torch.randn(T, H, 576, generator=g) * 0.3).to(torch.bfloat16)
    kx = torch.randn(T, 1, 576, generator=g).to(torch.bfloat16)
    qo = torch.zeros(B + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(torch.tensor(ext), 0)
    o = torch.empty(T, H, 512, dtype=torch.bfloat16, device=device)
    buf, kxd = rows8.to(device), kx.to(device)
    ops.extend_attention_fwd(q.to(device), kxd, kxd[..., :512], o, buf, buf[..., :512], qo.to(device), kv_indptr.to(device),
                             idx.to(device), None, None, max(ext), 576 ** -0.5)
    rows = rows8.to(torch.bfloat16)
    want = O.extend_attention(q, kx, kx[..., :512], rows, rows[..., :512], qo, kv_indptr, idx, 576 ** -0.5)
    torch.testing.assert_close(o.cpu().float(), want.float(), rtol=1.6e-2, atol=1e-2)


@pytest.mark.parametrize("kv", ["fp8_e5m2", "fp8_e4m3"])
def test_deepseek_engine_with_fp8_latent_cache(kv):
    """DeepSeek (MLA) with --kv-cache-dtype fp8_*: unified, chunked prefill (prefix rows are fp8) and Semi-PD against
    the oracle whose cache rounds the latent rows the same way."""
    from oracle.model import OracleDeepseekV2
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    from test_gpu_deepseek import tiny_deepseek
    from test_gpu_engine import check_against_oracle, make_prompts, server_args
    cfg = tiny_deepseek()
    kvd = torch.float8_e5m2 if kv == "fp8_e5m2" else torch.float8_e4m3fn
    prompts = make_prompts(cfg.vocab_size, [5, 150, 64, 1, 90, 33], seed=5)
    sp = SamplingParams(max_new_tokens=8, ignore_eos=True)
    eng = Engine(server_args(cfg, kv_cache_dtype=kv))
    try:
        sd = {k: v.float().cpu() for k, v in eng.model_runner.model.state_dict().items()}
        assert eng.model_runner.token_to_kv_pool.kv_buffer[0].dtype == kvd
        outs = eng.generate(prompts, sp)
    finally:
        eng.shutdown()
    oracle = OracleDeepseekV2(cfg, sd, kv_cache_dtype=kvd)
    margin = 8e-2  # e5m2 keeps 2 mantissa bits of every cached latent value
    assert check_against_oracle(oracle, prompts, outs, margin=margin) > 0.7
    semi = Engine(server_args(cfg, kv_cache_dtype=kv, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50,
                              chunked_prefill_size=64))
    try:
        got = semi.generate(prompts, sp, timeout=300)
    finally:
        semi.shutdown()
    # chunks of 64: later chunks of the long prompts read fp8 prefix rows, which the one-piece oracle prefill does
    # not; the margin covers it
    check_against_oracle(oracle, prompts, got, margin=0.15)
