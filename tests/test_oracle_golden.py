"""Pin the CPU oracle (oracle/ops.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import numpy as np
import pytest
import torch

from conftest import DTYPES, from_bits, load_golden
from oracle import ops as O


def test_rmsnorm_matches_reference_bitwise():
    g = load_golden("rmsnorm")
    eps = float(g["eps"])
    for ci in range(int(g["n"])):
        tag = f"c{ci}_"
        dt = DTYPES[str(g[tag + "dtype"])]
        x, r, w = (from_bits(g[tag + k], dt) for k in ("x", "r", "w"))
        y = O.rms_norm(x, w, eps)
        y2, r2 = O.fused_add_rms_norm(x, r, w, eps)
        assert torch.equal(y, from_bits(g[tag + "y"], dt))
        assert torch.equal(y2, from_bits(g[tag + "y_fused"], dt))
        assert torch.equal(r2, from_bits(g[tag + "r_fused"], dt))


def test_rope_caches_match_reference():
    g = load_golden("rope")
    c = O.cos_sin_cache_from_inv_freq(O.rope_inv_freq(64, 10000), 128)
    assert torch.equal(c, torch.from_numpy(g["base_neox_cache"]))
    c = O.cos_sin_cache_from_inv_freq(O.rope_inv_freq(32, 10000), 128)
    assert torch.equal(c, torch.from_numpy(g["partial_neox_cache"]))
    c = O.cos_sin_cache_from_inv_freq(O.llama3_inv_freq(128, 500000, 8.0, 1.0, 4.0, 64), 256)
    assert torch.equal(c, torch.from_numpy(g["llama3_cache"]))
    c = O.deepseek_yarn_cos_sin_cache(64, 64, 10000, 4.0, mscale=0.707, mscale_all_dim=0.707)
    assert torch.equal(c, torch.from_numpy(g["deepseek_yarn_cache"]))


def test_rope_apply_matches_reference():
    g = load_golden("rope")
    for name in g["names"]:
        name = str(name)
        cache = torch.from_numpy(g[name + "_cache"])
        pos = torch.from_numpy(g[name + "_pos"])
        q, k = torch.from_numpy(g[name + "_q"]), torch.from_numpy(g[name + "_k"])
        qo, ko = O.apply_rope(pos, q, k, int(g[name + "_head"]), cache, bool(g[name + "_neox"]))
        # fp32 inputs: forward_native computes in fp32 too, differences are re-association only
        torch.testing.assert_close(qo, torch.from_numpy(g[name + "_qo"]), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(ko, torch.from_numpy(g[name + "_ko"]), rtol=1e-6, atol=1e-6)


def test_kv_indices_and_pools_match_reference():
    g = load_golden("kv_indices")
    r2t = torch.from_numpy(g["req_to_token"])
    rpi = torch.from_numpy(g["req_pool_indices"])
    indptr, ind = O.create_kv_indices(r2t, rpi, torch.from_numpy(g["nostart_lens"]))
    assert np.array_equal(indptr.numpy(), g["nostart_indptr"]) and np.array_equal(ind.numpy(), g["nostart_indices"])
    indptr, ind = O.create_kv_indices(r2t, rpi, torch.from_numpy(g["start_lens"]), torch.from_numpy(g["start"]))
    assert np.array_equal(indptr.numpy(), g["start_indptr"]) and np.array_equal(ind.numpy(), g["start_indices"])


def test_decode_attention_matches_reference_triton_kernel():
    g = load_golden("decode_attention")
    for name in g["names"]:
        name = str(name)
        q, k, v = (torch.from_numpy(g[f"{name}_{x}"]) for x in ("q", "k", "v"))
        indptr, indices = torch.from_numpy(g[name + "_indptr"]), torch.from_numpy(g[name + "_indices"])
        splits, sm_scale, cap = g[name + "_meta"]
        o = O.decode_attention(q, k, v, indptr, indices, float(sm_scale), float(cap))
        torch.testing.assert_close(o, torch.from_numpy(g[name + "_o"]), rtol=2e-5, atol=2e-5)
        o2, mid = O.decode_attention_split(q, k, v, indptr, indices, int(splits), float(sm_scale), float(cap))
        torch.testing.assert_close(o2, torch.from_numpy(g[name + "_o"]), rtol=2e-5, atol=2e-5)
        ref_mid = torch.from_numpy(g[name + "_logits"])
        # compare only splits the reference wrote (empty splits are left untouched = 0)
        written = ref_mid[..., -1] != 0
        torch.testing.assert_close(mid[written], ref_mid[written], rtol=2e-5, atol=2e-5)


def test_extend_attention_matches_reference_triton_kernel():
    g = load_golden("extend_attention")
    for name in g["names"]:
        name = str(name)
        q, k, v = (torch.from_numpy(g[f"{name}_{x}"]) for x in ("q", "k", "v"))
        kb, vb = torch.from_numpy(g[name + "_kbuf"]), torch.from_numpy(g[name + "_vbuf"])
        sm_scale, cap = g[name + "_meta"]
        o = O.extend_attention(q, k, v, kb, vb, torch.from_numpy(g[name + "_qo_indptr"]),
                               torch.from_numpy(g[name + "_kv_indptr"]),
                               torch.from_numpy(g[name + "_kv_indices"]), float(sm_scale), float(cap))
        torch.testing.assert_close(o, torch.from_numpy(g[name + "_o"]), rtol=2e-5, atol=2e-5)


def test_extend_attention_custom_mask_matches_reference_triton_kernel():
    """custom_mask / mask_indptr / skip_prefix_custom_mask (extend_attention.py:291-307): tree masks with and
    without prefix bits, a logit cap, ragged head sizes, extends that cross the reference's BLOCK_M."""
    g = load_golden("extend_attention_mask")
    for name in g["names"]:
        name = str(name)
        q, k, v = (torch.from_numpy(g[f"{name}_{x}"]) for x in ("q", "k", "v"))
        kb, vb = torch.from_numpy(g[name + "_kbuf"]), torch.from_numpy(g[name + "_vbuf"])
        sm_scale, cap, skip = g[name + "_meta"]
        args = (q, k, v, kb, vb, torch.from_numpy(g[name + "_qo_indptr"]), torch.from_numpy(g[name + "_kv_indptr"]),
                torch.from_numpy(g[name + "_kv_indices"]), float(sm_scale), float(cap))
        o = O.extend_attention(*args, torch.from_numpy(g[name + "_mask"]), torch.from_numpy(g[name + "_mask_indptr"]),
                               bool(skip))
        torch.testing.assert_close(o, torch.from_numpy(g[name + "_o"]), rtol=2e-5, atol=2e-5)
        if name.startswith("causal_as_mask"):   # the mask that spells the default out changes nothing
            torch.testing.assert_close(o, O.extend_attention(*args), rtol=0, atol=0)
        else:                                   # and a tree mask does change the answer
            assert (o - O.extend_attention(*args)).abs().max() > 1e-3


def _same_topk(w, ids, rw, rids, rtol=1e-6, atol=1e-6):
    """Order inside the top-k is unspecified (sorted=False in the reference): compare as sets."""
    for t in range(ids.shape[0]):
        a = sorted(zip(ids[t].tolist(), w[t].tolist()))
        b = sorted(zip(rids[t].tolist(), rw[t].tolist()))
        assert [x[0] for x in a] == [x[0] for x in b], (t, a, b)
        np.testing.assert_allclose([x[1] for x in a], [x[1] for x in b], rtol=rtol, atol=atol)


def test_moe_routing_matches_reference():
    g = load_golden("moe_topk")
    for name in ("native_e8", "native_e64"):
        k, ren = g[name + "_meta"]
        w, ids = O.fused_topk_native(torch.from_numpy(g[name + "_gate"]), int(k), bool(ren))
        _same_topk(w.numpy(), ids.numpy(), g[name + "_w"], g[name + "_ids"])
    for name, dt, scoring in (("grouped_f32", torch.float32, "softmax"), ("grouped_bf16", torch.bfloat16, "softmax"),
                              ("grouped_sigmoid", torch.float32, "sigmoid")):
        k, ren, ng, tg = g[name + "_meta"]
        w, ids = O.grouped_topk(from_bits(g[name + "_gate"], dt), int(k), bool(ren), int(ng), int(tg), scoring)
        _same_topk(w.numpy(), ids.numpy(), g[name + "_w"], g[name + "_ids"])
    for name, dt in (("biased_f32", torch.float32), ("biased_bf16", torch.bfloat16)):
        k, ren, ng, tg = g[name + "_meta"]
        w, ids = O.biased_grouped_topk(from_bits(g[name + "_gate"], dt), torch.from_numpy(g[name + "_bias"]),
                                       int(k), bool(ren), int(ng), int(tg))
        _same_topk(w.numpy(), ids.numpy(), g[name + "_w"], g[name + "_ids"])


def test_silu_and_mul_matches_reference_test_formula():
    # sgl-kernel/tests/test_activation.py:11-15: y_ref = silu(x[..., d:])... the reference test uses
    # flashinfer order (gate first); formula: F.silu(x[..., :d]) * x[..., d:]
    torch.manual_seed(0)
    x = torch.randn(5, 256).to(torch.bfloat16)
    y = O.silu_and_mul(x)
    ref = (torch.nn.functional.silu(x[..., :128].float()) * x[..., 128:].float()).to(torch.bfloat16)
    assert torch.equal(y, ref)


def test_moe_align_properties():
    # sgl-kernel/tests/test_moe_align.py:151-222 compares expert_ids and num_tokens_post_pad between
    # two implementations; here we check the defining properties on the oracle.
    torch.manual_seed(1)
    E, bs = 8, 4
    ids = torch.randint(0, E, (13, 2), dtype=torch.int32)
    sorted_ids, expert_ids, npp = O.moe_align_block_size(ids, bs, E)
    n = int(npp)
    assert n % bs == 0
    flat = ids.flatten()
    real = sorted_ids[:n][sorted_ids[:n] < flat.numel()]
    assert sorted(real.tolist()) == list(range(flat.numel()))
    for blk in range(n // bs):
        for i in sorted_ids[blk * bs:(blk + 1) * bs].tolist():
            if i < flat.numel():
                assert int(flat[i]) == int(expert_ids[blk])


def test_fused_moe_staged_close_to_naive():
    torch.manual_seed(2)
    T, K, N, E, k = 6, 32, 16, 4, 2
    a = torch.randn(T, K).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * N, K) * 0.2).to(torch.bfloat16)
    w2 = (torch.randn(E, K, N) * 0.2).to(torch.bfloat16)
    w, ids = O.fused_topk_native(torch.randn(T, E), k, True)
    ref = O.fused_moe(a, w1, w2, w, ids)
    got = O.fused_moe_staged(a, w1, w2, w, ids)
    torch.testing.assert_close(got.float(), ref, rtol=1e-1, atol=1e-2)  # test_fused_moe.py:31-44


def test_greedy_argmax_first_max():
    x = torch.tensor([[0.0, 3.0, 3.0, -1.0], [5.0, 1.0, 5.0, 5.0]])
    assert O.greedy_argmax(x).tolist() == [1, 0]


def _check_counts(counts: np.ndarray, dist: torch.Tensor, draws: int):
    """Empirical counts of the reference sampler vs the oracle distribution: same support, and every
    frequency within 5 binomial standard deviations (+1 count)."""
    dist = dist.double().numpy()
    assert ((counts > 0) <= (dist > 0)).all(), "reference sampled a token outside the oracle support"
    sigma = np.sqrt(draws * dist * (1 - dist))
    assert (np.abs(counts - draws * dist) <= 5 * sigma + 1).all()
    # tokens the oracle keeps with >= 1 % probability must have been seen
    assert (counts[dist >= 0.01] > 0).all()


def test_sampling_distribution_matches_reference_counts():
    g = load_golden("sampling")
    probs = torch.from_numpy(g["probs"])
    top_ks, top_ps, min_ps = (torch.from_numpy(g[k]) for k in ("top_ks", "top_ps", "min_ps"))
    draws = int(g["draws"][0])
    torch.testing.assert_close(O.softmax_temperature(torch.from_numpy(g["logits"]), torch.from_numpy(g["temperatures"])),
                               probs, rtol=1e-6, atol=1e-7)
    d = O.top_k_top_p_min_p_filter(probs, top_ks, top_ps, torch.zeros_like(min_ps), False)
    _check_counts(g["counts"], d, draws)
    d = O.top_k_top_p_min_p_filter(probs, top_ks, top_ps, min_ps, True)
    _check_counts(g["counts_min_p"], d, draws)
    torch.testing.assert_close(O.top_p_normalize_probs(probs, top_ps), torch.from_numpy(g["top_p_normalized"]),
                               rtol=1e-6, atol=1e-7)


def test_sampling_renorm_formulas_agree():
    """The two statements of top-p renormalisation the reference holds (sampler.py:234-243 and
    sgl-kernel/tests/test_sampling.py:57-81) and the top-k one agree on continuous random rows."""
    g = torch.Generator().manual_seed(2)
    probs = torch.rand(5, 200, generator=g)
    probs = probs / probs.sum(-1, keepdim=True)
    for p in (0.1, 0.5, 0.9):
        torch.testing.assert_close(O.top_p_renorm_prob(probs, p), O.top_p_normalize_probs(probs, torch.full((5,), p)),
                                   rtol=1e-5, atol=1e-7)
    for k in (1, 10, 200):
        kept = (O.top_k_renorm_prob(probs, k) > 0).sum(-1)
        assert (kept == k).all()
    m = O.top_k_top_p_joint_mask(probs, 20, 0.5)
    d = O.top_k_top_p_min_p_filter(probs, torch.full((5,), 20, dtype=torch.int32), torch.full((5,), 0.5),
                                   torch.zeros(5), False)
    assert ((d > 0).int() <= m).all()


FP8_CODE = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


def test_block_fp8_quant_matches_reference():
    """oracle.per_token_group_quant_fp8 against two outputs of the reference on the same inputs:
    * its Triton kernel (fp8_kernel.py:75-115) run by the Triton interpreter on the e4m3fn / 448 branch: the
      scales are bit-identical; the quantised bytes are identical except where the INTERPRETER's own
      f32 -> fp8 cast is wrong (it rounds ties away from zero, and a value that rounds up into the next binade
      comes out with half the magnitude, subnormal results are flushed) — every mismatch must be one of those;
    * the torch helper of its test (test_block_fp8.py:19-44, torch's RNE cast, but x / s instead of
      x * (1 / s)): scales bit-identical (but for the eps group, clamped in the input dtype there), bytes identical except a few neighbours-by-one-code where the two
      quotients differ in the last bit."""
    g = load_golden("block_fp8")

    def f8(arr):
        return torch.from_numpy(arr.copy()).view(torch.float8_e4m3fn).float()

    for i, (rows, hidden, group, code) in enumerate(g["quant_cases"].tolist()):
        dt = FP8_CODE[code]
        x = from_bits(g[f"quant{i}_x"], dt)
        q, s = O.per_token_group_quant_fp8(x, group)
        assert torch.equal(s, torch.from_numpy(g[f"quant{i}_s"])), i
        s_native = torch.from_numpy(g[f"quant{i}_s_native"])
        live = s > 1e-12  # the helper clamps to eps in the input dtype: bf16(1e-10) is not 1e-10, f16(1e-10) is 0
        assert torch.equal(s[live], s_native[live]), i
        assert float(s[0, 0]) == pytest.approx(1e-10 / 448.0)  # the all-zero group sits on eps
        mine = q.view(torch.uint8).numpy()
        scaled = (x.float().reshape(-1, group) * (1.0 / s.reshape(-1, 1))).reshape(x.shape)
        # --- Triton interpreter
        bad = np.argwhere(mine != g[f"quant{i}_q"])
        assert len(bad) < 0.06 * mine.size
        for r, c in bad:
            a, b, v = float(f8(mine[r:r + 1, c])[0]), float(f8(g[f"quant{i}_q"][r:r + 1, c])[0]), float(scaled[r, c])
            binade_roll_over = abs(a) == 2 * abs(b) and np.log2(abs(a)) == int(np.log2(abs(a)))
            lo, hi = sorted((a, b))
            tie = lo < v < hi and abs((v - lo) - (hi - v)) < 1e-6 * abs(v)
            subnormal = abs(v) < 2.0 ** -6  # below the smallest normal e4m3fn the interpreter flushes to zero
            assert binade_roll_over or tie or subnormal, (i, r, c, v, a, b)
        # --- torch helper
        live_el = live.repeat_interleave(group, dim=-1).reshape(x.shape).numpy()
        native = np.where(live_el, g[f"quant{i}_q_native"], mine)
        diff = mine.astype(np.int16) - native.astype(np.int16)
        assert (np.abs(diff) <= 1).all() and (diff != 0).mean() < 0.01, (i, np.abs(diff).max(), (diff != 0).mean())


def test_block_fp8_matmul_matches_reference_kernel():
    """oracle.w8a8_block_fp8_matmul vs the reference's Triton kernel (fp8_kernel.py:409-491).  Products of fp8
    values are exact in fp32; only the order of the fp32 additions differs, so the comparison is far inside
    the reference's own bar (mean |diff| / mean |ref| < 1e-3, test_block_fp8.py:276-280)."""
    g = load_golden("block_fp8")
    for i, (M, N, K, code) in enumerate(g["mm_cases"].tolist()):
        dt = FP8_CODE[code]
        a = torch.from_numpy(g[f"mm{i}_a"]).view(torch.float8_e4m3fn)
        b = torch.from_numpy(g[f"mm{i}_b"]).view(torch.float8_e4m3fn)
        want = from_bits(g[f"mm{i}_c"], dt).float()
        got = O.w8a8_block_fp8_matmul(a, b, torch.from_numpy(g[f"mm{i}_as"]), torch.from_numpy(g[f"mm{i}_bs"]),
                                      [128, 128], dt).float()
        rel = (got - want).abs().mean() / want.abs().mean()
        # 16-bit outputs: the interpreter truncates f32 -> bf16 / f16 where torch rounds to nearest, so those
        # cases agree to one unit in the last place only; the f32 cases carry the arithmetic
        assert rel < {torch.float32: 1e-6, torch.bfloat16: 4e-3, torch.float16: 5e-4}[dt], (i, float(rel))
        if dt == torch.float32:
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


# ------------------------------------------------------------------- round 2: ops formerly pinned by formula only
def test_silu_and_mul_matches_reference_module_bitwise():
    """layers/activation.py:41-44 SiluAndMul.forward_native, run by make_golden.py."""
    g = load_golden("silu_and_mul")
    for ci in range(int(g["n"])):
        dt = DTYPES[str(g[f"c{ci}_dtype"])]
        x = from_bits(g[f"c{ci}_x"], dt)
        want = from_bits(g[f"c{ci}_y"], dt)
        assert torch.equal(O.silu_and_mul_native(x), want), ci
        # the fused op's semantics (fp32, one rounding; activation.cu:22-24): what the HIP kernel is checked against.
        # It may differ from forward_native by the rounding of the activation, i.e. one ulp of the output.
        y = O.silu_and_mul(x).float()
        ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 2.0 ** -22}[dt]
        assert torch.all((y - want.float()).abs() <= ulp * want.float().abs() + 1e-30), ci


def test_moe_align_matches_reference_triton_implementation():
    """The four-stage Triton implementation of sgl-kernel/tests/test_moe_align.py (interpreter).  The test
    there compares expert_ids and num_tokens_post_pad between implementations; the order of tokens INSIDE an
    expert's segment differs between the reference's own two implementations (the Triton one walks the ids in
    per-program strides), so sorted ids are compared per expert segment as sets, plus the sentinel padding."""
    g = load_golden("moe_align")
    for ci in range(int(g["n"])):
        bs, T, k, E = (int(x) for x in g[f"c{ci}_meta"])
        ids = torch.from_numpy(g[f"c{ci}_topk_ids"])
        sorted_ids, expert_ids, npp = O.moe_align_block_size(ids, bs, E)
        n = int(g[f"c{ci}_n_post"][0])
        assert int(npp) == n
        nb = n // bs
        assert np.array_equal(expert_ids.numpy()[:nb], g[f"c{ci}_expert_ids"][:nb])
        ref_sorted = g[f"c{ci}_sorted"]
        numel = ids.numel()
        for blk in range(nb):
            a = sorted_ids.numpy()[blk * bs:(blk + 1) * bs]
            b = ref_sorted[blk * bs:(blk + 1) * bs]
            assert (a == numel).sum() == (b == numel).sum()
        # per expert: same token set
        eids = expert_ids.numpy()[:nb]
        for e in np.unique(eids):
            blks = np.nonzero(eids == e)[0]
            seg = slice(blks[0] * bs, (blks[-1] + 1) * bs)
            assert sorted(sorted_ids.numpy()[seg].tolist()) == sorted(ref_sorted[seg].tolist())
        # the oracle's own order is the stable one (ascending flat index inside an expert)
        for e in np.unique(eids):
            blks = np.nonzero(eids == e)[0]
            seg = sorted_ids.numpy()[blks[0] * bs:(blks[-1] + 1) * bs]
            real = seg[seg < numel]
            assert np.all(np.diff(real) > 0)


def test_fused_moe_matches_reference_native_implementations():
    """fused_moe_native.py (fused_moe_forward_native, moe_forward_native) and the reference test's
    torch_naive_moe, run by make_golden.py on fp32 inputs; the oracle takes the routing as input."""
    g = load_golden("fused_moe")
    for ci in range(int(g["n"])):
        m, n, k, e, topk = (int(x) for x in g[f"c{ci}_meta"])
        a, w1, w2, score = (torch.from_numpy(g[f"c{ci}_{x}"]) for x in ("a", "w1", "w2", "score"))
        w, ids = O.fused_topk_native(score, topk, False)
        out = O.fused_moe(a, w1, w2, w, ids)
        torch.testing.assert_close(out, torch.from_numpy(g[f"c{ci}_out_naive"]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(out, torch.from_numpy(g[f"c{ci}_out_native"]), rtol=1e-4, atol=1e-5)
        w, ids = O.fused_topk_native(score, topk, True)
        out = O.fused_moe(a, w1, w2, w, ids)
        torch.testing.assert_close(out, torch.from_numpy(g[f"c{ci}_out_renorm"]), rtol=1e-4, atol=1e-5)
        # the staged variant (rounding where fused_experts_impl rounds) is the same product on fp32 inputs
        torch.testing.assert_close(O.fused_moe_staged(a, w1, w2, w, ids), out, rtol=1e-4, atol=1e-5)


def test_attention_8c_shapes_match_reference_triton_kernels():
    """SURVEY 8(c) shape list, run through the reference's Triton kernels by make_golden.py: MLA 576 / 512 through
    the grouped decode kernel with 16 and 128 heads, head sizes 80 and 13, GQA group 16, 16 kv splits, a logit cap
    on the grouped path; extend attention at 192 / 128 (MLA prefill), 576 / 512, 80, 13, group 16."""
    g = load_golden("decode_attention_8c")
    for name in g["names"]:
        name = str(name)
        q, k = torch.from_numpy(g[name + "_q"]), torch.from_numpy(g[name + "_k"])
        splits, sm_scale, cap, dv = g[name + "_meta"]
        v = k[..., :int(dv)] if name.startswith("mla") else torch.from_numpy(g[name + "_v"])
        indptr, indices = torch.from_numpy(g[name + "_indptr"]), torch.from_numpy(g[name + "_indices"])
        want = torch.from_numpy(g[name + "_o"])
        torch.testing.assert_close(O.decode_attention(q, k, v, indptr, indices, float(sm_scale), float(cap)), want,
                                   rtol=3e-5, atol=3e-5)
        o2, _ = O.decode_attention_split(q, k, v, indptr, indices, int(splits), float(sm_scale), float(cap))
        torch.testing.assert_close(o2, want, rtol=3e-5, atol=3e-5)
    g = load_golden("extend_attention_8c")
    for name in g["names"]:
        name = str(name)
        q, k, v = (torch.from_numpy(g[f"{name}_{x}"]) for x in ("q", "k", "v"))
        kb, vb = torch.from_numpy(g[name + "_kbuf"]), torch.from_numpy(g[name + "_vbuf"])
        sm_scale, cap = g[name + "_meta"]
        o = O.extend_attention(q, k, v, kb, vb, torch.from_numpy(g[name + "_qo_indptr"]),
                               torch.from_numpy(g[name + "_kv_indptr"]),
                               torch.from_numpy(g[name + "_kv_indices"]), float(sm_scale), float(cap))
        torch.testing.assert_close(o, torch.from_numpy(g[name + "_o"]), rtol=3e-5, atol=3e-5)


def test_per_tensor_fp8_helpers_match_reference_bitwise():
    """input_to_float8 / block_quant_to_tensor_quant of the reference (fp8_utils.py:137-188, OCP branch), run by
    make_golden.py: bytes and scales bit for bit; bmm_fp8's oracle against the unquantised product at the reference
    test's bar (cosine similarity > 0.99, sgl-kernel/tests/test_bmm_fp8.py:42-44)."""
    g = load_golden("bmm_fp8")
    F8 = torch.float8_e4m3fn
    q_nope = from_bits(g["q_nope"], torch.bfloat16)
    q, s = O.input_to_float8(q_nope.transpose(0, 1), F8)
    assert np.array_equal(q.view(torch.uint8).numpy(), g["q_nope_f8"]) and float(s) == float(g["q_nope_scale_inv"])
    q, s = O.input_to_float8(torch.from_numpy(g["y"]), torch.float8_e5m2)
    assert np.array_equal(q.view(torch.uint8).numpy(), g["y_f8"]) and float(s) == float(g["y_scale_inv"])
    wq = torch.from_numpy(g["w_block_q"]).view(F8)
    tq, ts = O.block_quant_to_tensor_quant(wq, torch.from_numpy(g["w_block_s"]), [128, 128])
    assert np.array_equal(tq.view(torch.uint8).numpy(), g["w_tensor_q"]) and float(ts) == float(g["w_tensor_scale_inv"])
    a8 = torch.from_numpy(g["bmm_a8"]).view(F8)
    b8 = torch.from_numpy(g["bmm_b8"]).view(F8)          # memory [h, n, k]
    out = O.bmm_fp8(a8, b8.transpose(1, 2), torch.from_numpy(g["bmm_a_s"]), torch.from_numpy(g["bmm_b_s"]), torch.bfloat16)
    ref = torch.from_numpy(g["bmm_ref_unquantised"])
    cos = torch.nn.functional.cosine_similarity(ref.reshape(-1), out.float().reshape(-1), dim=0)
    assert cos > 0.99
