"""GPU parity of the block-scaled fp8 path against the oracle and the reference's golden vectors.

per_token_group_quant_fp8 is elementwise IEEE arithmetic: bytes and scales must be bit-identical to the
oracle.  The matmuls sum exact fp8 products in fp32 in a different order than the oracle: compared with the
reference's own criterion (mean |diff| / mean |ref| < 1e-3 for the matmul, < 2e-2 for the fused MoE,
test/test_block_fp8.py:276-280, 397-401) and with a 20x tighter bound on f32 outputs (the inputs are uniform in +-448, so sums cancel heavily)."""
import numpy as np
import pytest
import torch

from conftest import from_bits, load_golden
from oracle import ops as O
from semi_pd_amd import ops

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn
CODE = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


def rel_err(got, want):
    got, want = got.float().cpu(), want.float()
    return float((got - want).abs().mean() / want.abs().mean().clamp(min=1e-30))


# ----------------------------------------------------------------------------------------------- quant
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("group", [64, 128, 256, 512])
def test_quant_bit_exact_vs_oracle(device, dtype, group):
    g = torch.Generator().manual_seed(group)
    for rows, hidden in [(1, group), (7, 4 * group), (83, 5120 if 5120 % group == 0 else 4096), (2048, 512)]:
        x = (torch.randn(rows, hidden, generator=g) * (10.0 ** torch.randint(-4, 3, (rows, 1), generator=g))).to(dtype)
        x[0, :group] = 0
        q, s = ops.per_token_group_quant_fp8(x.to(device), group)
        q_ref, s_ref = O.per_token_group_quant_fp8(x, group)
        assert q.dtype == F8 and s.dtype == torch.float32 and s.shape == (rows, hidden // group)
        assert torch.equal(s.cpu(), s_ref)
        assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))


def test_quant_golden_and_errors(device):
    g = load_golden("block_fp8")
    for i, (rows, hidden, group, code) in enumerate(g["quant_cases"].tolist()):
        x = from_bits(g[f"quant{i}_x"], CODE[code])
        q, s = ops.per_token_group_quant_fp8(x.to(device), group)
        assert torch.equal(s.cpu(), torch.from_numpy(g[f"quant{i}_s"]))  # the reference kernel's scales, bit for bit
        q_ref, _ = O.per_token_group_quant_fp8(x, group)                 # bytes: see tests/test_oracle_golden.py
        assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    x = torch.randn(4, 3, 256, device=device, dtype=torch.bfloat16)      # leading dimensions are flattened
    q, s = ops.per_token_group_quant_fp8(x, 128)
    assert q.shape == x.shape and s.shape == (4, 3, 2)
    with pytest.raises(RuntimeError, match="cannot be divisible"):
        ops.per_token_group_quant_fp8(torch.randn(2, 200, device=device), 128)
    with pytest.raises(RuntimeError, match="not contiguous"):
        ops.per_token_group_quant_fp8(torch.randn(256, 2, device=device).t(), 128)
    with pytest.raises(RuntimeError, match="group size"):
        ops.per_token_group_quant_fp8(torch.randn(2, 96, device=device), 32)


# ----------------------------------------------------------------------------------------------- matmul
def make_mm(M, N, K, seed, block_n=128):
    g = torch.Generator().manual_seed(seed)
    A = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).to(F8)
    B = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).to(F8)
    As = torch.rand(M, -(-K // 128), generator=g) * 1e-2
    Bs = torch.rand(-(-N // block_n), -(-K // 128), generator=g) * 1e-2
    return A, B, As, Bs


@pytest.mark.parametrize("M", [1, 7, 64, 65, 83, 128, 200, 512])
def test_matmul_vs_oracle(device, M):
    # (N, K): DeepSeek-V3 shapes scaled down, ragged N (not a multiple of 16 / 64 / 128), K with a partial
    # last scale block (400 = 3 * 128 + 16) and K shorter than one chunk
    shapes = [(1536, 7168), (576, 1536), (4096, 512), (200, 1024), (272, 400), (130, 128), (512, 2304)]
    if M >= 128:
        shapes.append((3072, 7168))  # > 20 M elements: the 128-row blocks (smaller matrices take 64-row blocks up to M = 512)
    for i, (N, K) in enumerate(shapes):
        A, B, As, Bs = make_mm(M, N, K, 100 * M + i)
        for out_dtype in (torch.float32, torch.bfloat16):
            got = ops.w8a8_block_fp8_matmul(A.to(device), B.to(device), As.to(device), Bs.to(device), [128, 128], out_dtype)
            want = O.w8a8_block_fp8_matmul(A, B, As, Bs, [128, 128], out_dtype)
            assert got.shape == (M, N) and got.dtype == out_dtype
            r = rel_err(got, want)
            assert r < (5e-5 if out_dtype == torch.float32 else 1e-3), (M, N, K, out_dtype, r)


def test_matmul_golden_block_n_and_errors(device):
    g = load_golden("block_fp8")
    for i, (M, N, K, code) in enumerate(g["mm_cases"].tolist()):
        a = torch.from_numpy(g[f"mm{i}_a"].copy()).view(F8)
        b = torch.from_numpy(g[f"mm{i}_b"].copy()).view(F8)
        got = ops.w8a8_block_fp8_matmul(a.to(device), b.to(device), torch.from_numpy(g[f"mm{i}_as"]).to(device),
                                        torch.from_numpy(g[f"mm{i}_bs"]).to(device), [128, 128], CODE[code])
        r = rel_err(got, from_bits(g[f"mm{i}_c"], CODE[code]))
        assert r < {torch.float32: 5e-5, torch.bfloat16: 4e-3, torch.float16: 5e-4}[CODE[code]], (i, r)
    # other weight block heights (fused_moe tests use 64 too); leading activation dimensions
    for block_n in (64, 16, 256):
        A, B, As, Bs = make_mm(24, 320, 640, 7, block_n)
        got = ops.w8a8_block_fp8_matmul(A.view(2, 12, 640).to(device), B.to(device), As.view(2, 12, -1).to(device),
                                        Bs.to(device), [block_n, 128], torch.float32)
        assert got.shape == (2, 12, 320)
        assert rel_err(got.view(24, 320), O.w8a8_block_fp8_matmul(A, B, As, Bs, [block_n, 128], torch.float32)) < 5e-5
    A, B, As, Bs = (t.to(device) for t in make_mm(4, 128, 256, 1))
    with pytest.raises(RuntimeError, match="block_k must be 128"):
        ops.w8a8_block_fp8_matmul(A, B, As[:, :1].repeat(1, 4).contiguous(), Bs[:, :1].repeat(1, 4).contiguous(), [128, 64])
    with pytest.raises(RuntimeError, match="scale shapes"):
        ops.w8a8_block_fp8_matmul(A, B, As[:, :1].contiguous(), Bs, [128, 128])
    with pytest.raises(RuntimeError, match="must be torch.float8_e4m3fn"):
        ops.w8a8_block_fp8_matmul(A.view(torch.uint8), B, As, Bs, [128, 128])


def test_quant_then_matmul_approximates_the_bf16_linear(device):
    """The whole linear (apply_w8a8_block_fp8_linear, fp8_utils.py:91-134): quantise activations, block-fp8
    matmul against block-quantised weights; close to the unquantised product at fp8 accuracy."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(48, 2048, generator=g).to(torch.bfloat16)
    w = (torch.randn(1024, 2048, generator=g) * 0.02).to(torch.bfloat16)
    wb = w.float().view(8, 128, 16, 128)
    ws = wb.abs().amax(dim=(1, 3)) / 448.0
    wq = (wb / ws.view(8, 1, 16, 1)).clamp(-448, 448).to(F8).view(1024, 2048)
    xq, xs = ops.per_token_group_quant_fp8(x.to(device), 128)
    y = ops.w8a8_block_fp8_matmul(xq, wq.to(device), xs, ws.to(device), [128, 128], torch.bfloat16)
    ref = x.float() @ w.float().t()
    assert rel_err(y, ref) < 0.05


# ----------------------------------------------------------------------------------------------- fused MoE
@pytest.mark.parametrize("T,block_m", [(1, 64), (33, 64), (222, 64), (222, 128), (1500, 128)])
def test_fused_moe_block_fp8_vs_oracle(device, T, block_m):
    from semi_pd_amd.layers.moe import fused_experts_fp8
    E, topk, K, N = 8, 2, 512, 384
    g = torch.Generator().manual_seed(T)
    a = (torch.randn(T, K, generator=g) / 10).to(torch.bfloat16)
    w1 = ((torch.rand(E, 2 * N, K, generator=g) - 0.5) * 2 * 448).to(F8)
    w2 = ((torch.rand(E, K, N, generator=g) - 0.5) * 2 * 448).to(F8)
    w1_s = torch.rand(E, 2 * N // 128, K // 128, generator=g) * 1e-2
    w2_s = torch.rand(E, K // 128, N // 128, generator=g) * 1e-2
    score = torch.softmax(torch.randn(T, E, generator=g), dim=-1)
    tw, ti = torch.topk(score, topk)
    got = fused_experts_fp8(a.to(device), w1.to(device), w2.to(device), w1_s.to(device), w2_s.to(device), tw.to(device),
                            ti.to(torch.int32).to(device), [128, 128], block_m=block_m)
    want = O.fused_moe_block_fp8(a, w1, w2, w1_s, w2_s, tw, ti, [128, 128])
    assert got.shape == (T, K) and got.dtype == torch.bfloat16
    assert rel_err(got, want) < 0.02  # the reference's bar (test_block_fp8.py:397-401); measured far below
    assert rel_err(got, want) < 5e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_silu_and_mul_quant_fused_equals_the_two_calls(device, dtype):
    """ops.silu_and_mul_quant_fp8 == silu_and_mul then per_token_group_quant_fp8 (what fused_experts_impl runs between
    its two GEMMs, fused_moe.py:1104-1125): bit-identical bytes and scales, against the oracle and against the two
    separate kernels."""
    g = torch.Generator().manual_seed(9)
    for rows, d, group in [(1, 128, 128), (37, 1408, 128), (300, 2048, 64), (5, 512, 512)]:
        x = (torch.randn(rows, 2 * d, generator=g) * 2).to(dtype)
        q, s = ops.silu_and_mul_quant_fp8(x.to(device), group)
        act = O.silu_and_mul(x)
        q_ref, s_ref = O.per_token_group_quant_fp8(act, group)
        assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
        q2, s2 = ops.per_token_group_quant_fp8(ops.silu_and_mul(x.to(device)), group)
        assert torch.equal(q.view(torch.uint8), q2.view(torch.uint8)) and torch.equal(s, s2)
    with pytest.raises(RuntimeError, match="cannot be divisible"):
        ops.silu_and_mul_quant_fp8(torch.randn(2, 200, device=device, dtype=dtype), 128)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_add_rmsnorm_quant_equals_the_two_calls(device, dtype):
    """ops.fused_add_rmsnorm_quant_fp8 == fused_add_rmsnorm then per_token_group_quant_fp8: the same normalised rows
    and residual, bit-identical bytes and scales (against the two kernels and against the oracle)."""
    g = torch.Generator().manual_seed(4)
    for rows, hidden, group in [(1, 128, 128), (37, 2048, 128), (300, 7168, 128), (5, 512, 64), (3, 8192, 512)]:
        x = torch.randn(rows, hidden, generator=g).to(dtype)
        r = torch.randn(rows, hidden, generator=g).to(dtype)
        w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
        x1, r1 = x.to(device), r.to(device)
        q, s = ops.fused_add_rmsnorm_quant_fp8(x1, r1, w.to(device), 1e-6, group)
        x2, r2 = x.to(device), r.to(device)
        ops.fused_add_rmsnorm(x2, r2, w.to(device), 1e-6)
        q2, s2 = ops.per_token_group_quant_fp8(x2, group)
        assert torch.equal(x1, x2) and torch.equal(r1, r2)
        assert torch.equal(q.view(torch.uint8), q2.view(torch.uint8)) and torch.equal(s, s2)
        y, _ = O.fused_add_rms_norm(x, r, w, 1e-6)
        q_ref, s_ref = O.per_token_group_quant_fp8(x1.cpu(), group)   # quantiser on the kernel's own rows: bit-exact
        assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s.cpu(), s_ref)
        torch.testing.assert_close(x1.cpu().float(), y.float(), rtol=1e-2, atol=1e-2)
    with pytest.raises(RuntimeError, match="multiple of the group size"):
        ops.fused_add_rmsnorm_quant_fp8(torch.randn(2, 200, device=device, dtype=dtype),
                                        torch.randn(2, 200, device=device, dtype=dtype),
                                        torch.ones(200, device=device, dtype=dtype), 1e-6, 128)


# ----------------------------------------------------------------------------- per-tensor fp8: MLA absorption (bmm_fp8)
def test_input_to_float8_matches_reference_and_oracle(device):
    """Bytes and scale of the reference's input_to_float8 (golden/bmm_fp8.npz) on a transposed bf16 view, and the
    oracle on random tensors of both fp8 types."""
    g = load_golden("bmm_fp8")
    q_nope = from_bits(g["q_nope"], torch.bfloat16).to(device)
    q, s = ops.input_to_float8(q_nope.transpose(0, 1), torch.float8_e4m3fn)
    assert q.is_contiguous() and np.array_equal(q.view(torch.uint8).cpu().numpy(), g["q_nope_f8"])
    assert float(s) == float(g["q_nope_scale_inv"])
    torch.manual_seed(1)
    for dt, f8 in ((torch.bfloat16, torch.float8_e5m2), (torch.float16, torch.float8_e4m3fn), (torch.bfloat16, torch.float8_e4m3fn)):
        x = (torch.randn(5, 33, 64) * 7).to(dt)
        x[2, 3, 4] = 300.0
        qo, so = O.input_to_float8(x, f8)
        qg, sg = ops.input_to_float8(x.to(device), f8)
        assert torch.equal(qg.view(torch.uint8).cpu(), qo.view(torch.uint8)) and float(sg) == float(so)


def test_input_to_float8_in_a_replayed_graph(device):
    """Two quantisations inside one hipGraph (the absorbed MLA decode runs two per layer), a large tensor in front of a
    small one: the second must not inherit the first one's maximum, on the capture and on every replay with new data."""
    torch.manual_seed(2)
    big = torch.empty(8, 8, 128, dtype=torch.bfloat16, device=device)
    small = torch.empty(8, 8, 512, dtype=torch.bfloat16, device=device)
    stream = torch.cuda.Stream(device=device)

    def run():
        return ops.input_to_float8(big.transpose(0, 1)), ops.input_to_float8(small.transpose(0, 1))

    def fill(seed):
        g = torch.Generator().manual_seed(seed)
        big.copy_((torch.randn(8, 8, 128, generator=g) * 40).to(torch.bfloat16))
        small.copy_((torch.randn(8, 8, 512, generator=g) * 0.05).to(torch.bfloat16))

    fill(0)
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        run()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        (qb, sb), (qs, ss) = run()
    for seed in (1, 2, 3):
        fill(seed)
        graph.replay()
        torch.cuda.synchronize()
        for x, q, s in ((big, qb, sb), (small, qs, ss)):
            qo, so = O.input_to_float8(x.cpu().transpose(0, 1), torch.float8_e4m3fn)
            assert float(s) == float(so)
            assert torch.equal(q.view(torch.uint8).cpu(), qo.contiguous().view(torch.uint8))


@pytest.mark.parametrize("a_dt,b_dt", [(torch.float8_e4m3fn, torch.float8_e4m3fn), (torch.float8_e4m3fn, torch.float8_e5m2),
                                       (torch.float8_e5m2, torch.float8_e4m3fn)])
@pytest.mark.parametrize("res_dtype", [torch.bfloat16, torch.float16])
def test_bmm_fp8_reference_test_shapes(device, a_dt, b_dt, res_dtype):
    """sgl-kernel/tests/test_bmm_fp8.py: [16, 48, 64] x [16, 64, 80] (column-major), cosine similarity > 0.99 against
    the unquantised product; plus the oracle (same fp8 operands, fp32 accumulate) to rounding."""
    torch.manual_seed(0)
    a = torch.randn(16, 48, 64).to(torch.bfloat16)
    b = torch.randn(16, 80, 64).to(torch.bfloat16)          # memory [b, n, k]; mat2 = b.transpose(-2, -1)
    a8, a_s = O.input_to_float8(a, a_dt)
    b8, b_s = O.input_to_float8(b, b_dt)
    out = ops.bmm_fp8(a8.to(device), b8.to(device).transpose(1, 2), a_s.to(device), b_s.to(device), res_dtype)
    ref = torch.bmm(a.float(), b.float().transpose(1, 2))
    cos = torch.nn.functional.cosine_similarity(ref.reshape(-1), out.float().cpu().reshape(-1), dim=0)
    assert cos > 0.99
    want = O.bmm_fp8(a8, b8.transpose(1, 2), a_s, b_s, res_dtype)
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=1e-2 if res_dtype == torch.bfloat16 else 2e-3, atol=1e-2)


@pytest.mark.parametrize("T", [1, 9, 70])
def test_bmm_fp8_mla_absorb_shapes_and_strided_output(device, T):
    """The two products of forward_absorb (deepseek_v2.py:659-665, 690-700) with the outputs written through strides
    into their consumers' layouts: q_nope [H, T, 128] x W_kc -> q_input[T, H, :512]; attn [H, T, 512] x W_vc ->
    [T, H, 128]."""
    H = 16
    torch.manual_seed(T)
    q = torch.randn(T, H, 192).to(torch.bfloat16)
    w_kc = torch.randn(H, 512, 128).to(torch.bfloat16)       # column-major B: memory [H, n, k]
    w_kc8, ws = O.input_to_float8(w_kc, torch.float8_e4m3fn)
    qd = q.to(device)
    q8, qs = ops.input_to_float8(qd[..., :128].transpose(0, 1), torch.float8_e4m3fn)
    q8o, qso = O.input_to_float8(q[..., :128].transpose(0, 1), torch.float8_e4m3fn)
    assert torch.equal(q8.view(torch.uint8).cpu(), q8o.view(torch.uint8)) and float(qs) == float(qso)
    q_input = torch.full((T, H, 576), 7.0, dtype=torch.bfloat16, device=device)
    ops.bmm_fp8(q8, w_kc8.to(device).transpose(1, 2), qs, ws.to(device), torch.bfloat16,
                out=q_input[..., :512].transpose(0, 1))
    want = O.bmm_fp8(q8o, w_kc8.transpose(1, 2), qso, ws, torch.bfloat16).transpose(0, 1)
    torch.testing.assert_close(q_input[..., :512].cpu().float(), want.float(), rtol=1e-2, atol=1e-2)
    assert torch.all(q_input[..., 512:] == 7.0)              # nothing written past the 512 columns
    attn = torch.randn(T, H, 512).to(torch.bfloat16)
    w_vc = torch.randn(H, 128, 512).to(torch.bfloat16)       # [H, n = 128, k = 512]
    w_vc8, wvs = O.input_to_float8(w_vc, torch.float8_e4m3fn)
    a8, a_s = ops.input_to_float8(attn.to(device).transpose(0, 1), torch.float8_e4m3fn)
    out = torch.empty((T, H, 128), dtype=torch.bfloat16, device=device)
    ops.bmm_fp8(a8, w_vc8.to(device).transpose(1, 2), a_s, wvs.to(device), torch.bfloat16, out=out.transpose(0, 1))
    a8o, a_so = O.input_to_float8(attn.transpose(0, 1), torch.float8_e4m3fn)
    want = O.bmm_fp8(a8o, w_vc8.transpose(1, 2), a_so, wvs, torch.bfloat16).transpose(0, 1)
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=1e-2, atol=2e-2)
