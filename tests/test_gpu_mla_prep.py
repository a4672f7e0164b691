"""MLA decode step: the merged [q_proj | kv_a_proj_with_mqa] GEMM's planes through ONE launch (ops.mla_decode_prep) against the
launches it replaces -- the GEMM's reduction, kv_a_layernorm, the strided RoPE, the q_pe copy and set_kv_buffer of
DeepseekV2AttentionMLA.forward_absorb (models/deepseek_v2.py:633-706 in the reference).  Bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from semi_pd_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("T", [1, 7, 32, 64])
@pytest.mark.parametrize("H,hidden", [(16, 2048), (128, 1024)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv_dtype", [None, torch.float8_e4m3fn])
def test_mla_decode_prep_has_the_bits_of_the_launches_it_replaces(ops, T, H, hidden, dtype, kv_dtype):
    dev = torch.device("cuda:0")
    nope, rope, lora = 128, 64, 512
    g = torch.Generator().manual_seed(T * 131 + H)
    x = torch.randn(T, hidden, generator=g).to(dtype).to(dev)
    w = (torch.randn(H * (nope + rope) + lora + rope, hidden, generator=g) * 0.05).to(dtype).to(dev)
    nw = (torch.rand(lora, generator=g) + 0.5).to(dtype).to(dev)
    positions = torch.randint(0, 4000, (T,), generator=g).to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rope, 2, dtype=torch.float32) / rope))
    fr = torch.arange(4096, dtype=torch.float32)[:, None] * inv[None, :]
    cache = torch.cat([fr.cos(), fr.sin()], -1).contiguous().to(dev)
    slots = 300
    loc = torch.randperm(slots, generator=g)[:T].to(torch.int64).to(dev)
    pool_dtype = kv_dtype or dtype
    pool_a = torch.zeros((slots, 1, lora + rope), dtype=dtype, device=dev).to(pool_dtype)
    pool_b = pool_a.clone()
    # the launches it replaces, on the same GEMM
    y = ops.stream_linear(x, w)
    q = y[:, : H * (nope + rope)].view(T, H, nope + rope)
    latent = y[:, H * (nope + rope):].contiguous()
    ops.rmsnorm(latent[:, :lora], nw, 1e-6, out=latent[:, :lora])
    latent = latent.unsqueeze(1)
    ops.apply_rope_strided_inplace(positions, q[..., nope:], latent[..., lora:], cache, False)
    q_input_ref = torch.zeros((T, H, lora + rope), dtype=dtype, device=dev)
    q_input_ref[..., lora:] = q[..., nope:]
    ops.store_kv_rows(pool_a, loc, latent)
    # one launch
    planes = ops.stream_linear_planes(x, w)
    q_input = torch.zeros((T, H, lora + rope), dtype=dtype, device=dev)
    q_nope = ops.mla_decode_prep(planes, positions, cache, nw, 1e-6, H, nope, rope, lora, pool_b, loc, q_input)
    assert torch.equal(q_nope, q[..., :nope].contiguous())
    assert torch.equal(q_input, q_input_ref)
    assert torch.equal(pool_a.view(torch.uint8), pool_b.view(torch.uint8))
    assert pool_b.float().abs().sum() > 0


@pytest.mark.parametrize("M", [1, 16, 33, 64])
@pytest.mark.parametrize("H,N,K", [(16, 512, 128), (16, 128, 512), (3, 96, 160)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bmm_nk_matches_fp32_through_the_strides_of_forward_absorb(ops, M, H, N, K, dtype):
    """ops.bmm_nk in the two layouts DeepseekV2AttentionMLA.forward_absorb uses it in: x = a [T, H, K] tensor seen as
    [H, T, K], out = the [H, T, N] view of a [T, H, N + pad] tensor (q_input[..., :512])."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, H, K, generator=g).to(dtype).to(dev)
    w_nk = (torch.randn(H, N, K, generator=g) * 0.1).to(dtype).to(dev)
    holder = torch.full((M, H, N + 64), 7.0, dtype=dtype, device=dev)
    ops.bmm_nk(x.transpose(0, 1), w_nk, out=holder[..., :N].transpose(0, 1))
    want = torch.einsum("mhk,hnk->mhn", x.float(), w_nk.float())
    torch.testing.assert_close(holder[..., :N].float(), want, rtol=2e-2, atol=2e-2)
    assert torch.all(holder[..., N:] == 7.0)                  # nothing written past the view
    lib = torch.bmm(x.transpose(0, 1), w_nk.transpose(1, 2)).transpose(0, 1)
    torch.testing.assert_close(holder[..., :N].float(), lib.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("T", [1, 32, 64])
@pytest.mark.parametrize("H", [16, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv_dtype", [None, torch.float8_e5m2])
def test_mla_decode_prep_rows_has_the_bits_of_the_launches_it_replaces(ops, T, H, dtype, kv_dtype):
    """The q_lora / block-fp8 path: q and the latent are tensors already (strided views of wider GEMM outputs)."""
    dev = torch.device("cuda:0")
    nope, rope, lora = 128, 64, 512
    g = torch.Generator().manual_seed(T * 17 + H)
    q_wide = torch.randn(T, H * (nope + rope) + 64, generator=g).to(dtype).to(dev)
    q = q_wide[:, : H * (nope + rope)].view(T, H, nope + rope)                      # row stride wider than the row
    y = torch.randn(T, 1536 + lora + rope, generator=g).to(dtype).to(dev)
    lat = y[:, 1536:]                                                               # the tail of a merged q_a | kv_a output
    nw = (torch.rand(lora, generator=g) + 0.5).to(dtype).to(dev)
    positions = torch.randint(0, 4000, (T,), generator=g).to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rope, 2, dtype=torch.float32) / rope))
    fr = torch.arange(4096, dtype=torch.float32)[:, None] * inv[None, :]
    cache = torch.cat([fr.cos(), fr.sin()], -1).contiguous().to(dev)
    slots = 200
    loc = torch.randperm(slots, generator=g)[:T].to(torch.int64).to(dev)
    pool_dtype = kv_dtype or dtype
    pool_a = torch.zeros((slots, 1, lora + rope), dtype=dtype, device=dev).to(pool_dtype)
    pool_b = pool_a.clone()
    q_before, lat_before = q.clone(), lat.clone()
    # one launch (first: the separate launches below work in place)
    q_input = torch.zeros((T, H, lora + rope), dtype=dtype, device=dev)
    ops.mla_decode_prep_rows(q, lat, positions, cache, nw, 1e-6, nope, lora, pool_b, loc, q_input)
    assert torch.equal(q, q_before) and torch.equal(lat, lat_before)
    # the launches it replaces
    q1, l1 = q.clone(), lat.clone()
    ops.rmsnorm(l1[:, :lora], nw, 1e-6, out=l1[:, :lora])
    l1 = l1.unsqueeze(1)
    ops.apply_rope_strided_inplace(positions, q1[..., nope:], l1[..., lora:], cache, False)
    q_input_ref = torch.zeros((T, H, lora + rope), dtype=dtype, device=dev)
    q_input_ref[..., lora:] = q1[..., nope:]
    ops.store_kv_rows(pool_a, loc, l1)
    assert torch.equal(q_input, q_input_ref)
    assert torch.equal(pool_a.view(torch.uint8), pool_b.view(torch.uint8))


@pytest.mark.parametrize("T", [1, 32, 64])
@pytest.mark.parametrize("E,K,topk,groups,topk_group,biased", [(64, 2048, 6, 1, 1, False), (256, 7168, 8, 8, 4, True),
                                                               (160, 1024, 6, 8, 3, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_grouped_topk_on_the_router_gemms_planes_equals_the_two_launches(ops, T, E, K, topk, groups, topk_group, biased, dtype):
    """MoEGate on the weight-streaming GEMM, stopped before its K-slice reduction, + the routing kernel that sums the planes
    (DeepseekV2MoE.forward for decode batches) against the finished GEMM + grouped_topk: same ids, same weights."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T + E)
    x = torch.randn(T, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(E, K, generator=g) * 0.05).to(dtype).to(dev)
    bias = (torch.randn(E, generator=g) * 0.1).float().to(dev) if biased else None
    scoring = "sigmoid" if biased else "softmax"
    logits = ops.stream_linear(x, w)
    w1, i1 = ops.grouped_topk(logits, topk, True, groups, topk_group, bias, scoring)
    planes = ops.stream_linear_planes(x, w)
    w2, i2 = ops.grouped_topk(planes, topk, True, groups, topk_group, bias, scoring)
    assert torch.equal(i1, i2) and torch.equal(w1, w2)


@pytest.mark.parametrize("T", [1, 32, 64])
@pytest.mark.parametrize("topk,H,K", [(6, 2048, 2816), (8, 7168, 256)])
@pytest.mark.parametrize("scale", [1.0, 2.5])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_moe_sum_with_the_addends_planes_equals_the_finished_addend(ops, T, topk, H, K, scale, dtype):
    """The tail of DeepseekV2MoE.forward with the shared experts' down_proj still in K-slice planes: the launch that sums the
    top-k rows sums the planes too -- the bits of stream_linear followed by moe_sum_scale_add."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T * topk + H)
    x = torch.randn(T, topk, H, generator=g).to(dtype).to(dev)
    a = torch.randn(T, K, generator=g).to(dtype).to(dev)
    w = (torch.randn(H, K, generator=g) * 0.05).to(dtype).to(dev)
    want = ops.moe_sum_scale_add(x, scale, ops.stream_linear(a, w))
    got = ops.moe_sum_scale_add(x, scale, ops.stream_linear_planes(a, w))
    assert torch.equal(got, want)


@pytest.mark.parametrize("T", [1, 33, 64, 300])
@pytest.mark.parametrize("H,group", [(1536, 128), (512, 64), (4096, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rmsnorm_quant_fp8_has_the_bytes_of_the_two_launches(ops, T, H, group, dtype):
    """q_a_layernorm + the per-token-group quantisation in front of q_b_proj (block-fp8 DeepSeek-V3), on a strided view."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T + H)
    wide = torch.randn(T, H + 576, generator=g).to(dtype).to(dev)
    x = wide[:, :H]
    w = (torch.rand(H, generator=g) + 0.5).to(dtype).to(dev)
    want = ops.rmsnorm(x, w, 1e-6)
    want_q, want_s = ops.per_token_group_quant_fp8(want.contiguous(), group)
    out, (xq, xs) = ops.rmsnorm_quant_fp8(x, w, 1e-6, group)
    assert torch.equal(out, want)
    assert torch.equal(xq.view(torch.uint8), want_q.view(torch.uint8)) and torch.equal(xs, want_s)
