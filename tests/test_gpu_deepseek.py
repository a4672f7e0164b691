"""GPU parity of the DeepSeek-V2-Lite-shaped model (MLA absorbed decode + MoE grouped GEMM) against the
CPU oracle (non-absorbed formulation, pinned to HF DeepseekV2ForCausalLM), unified and Semi-PD."""
import pytest
import torch

from oracle.model import OracleDeepseekV2
from test_gpu_engine import check_against_oracle, make_prompts, server_args

pytestmark = pytest.mark.gpu


def tiny_deepseek(**kw):
    from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
    base = dict(vocab_size=1000, hidden_size=512, intermediate_size=1024, moe_intermediate_size=256,
                num_hidden_layers=3, num_attention_heads=8, n_shared_experts=2, n_routed_experts=16,
                num_experts_per_tok=4, max_position_embeddings=4096,
                rope_scaling={"type": "yarn", "factor": 4, "beta_fast": 32, "beta_slow": 1, "mscale": 0.707,
                              "mscale_all_dim": 0.707, "original_max_position_embeddings": 1024})
    base.update(kw)
    return DeepseekV2Config(**base)


@pytest.fixture(scope="module")
def unified_deepseek():
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg = tiny_deepseek()
    eng = Engine(server_args(cfg))
    sd = {k: v.float().cpu() for k, v in eng.model_runner.model.state_dict().items()}
    prompts = make_prompts(cfg.vocab_size, [5, 37, 130, 1, 64, 17])
    outs = eng.generate(prompts, SamplingParams(max_new_tokens=10, ignore_eos=True))
    yield cfg, sd, prompts, outs, eng
    eng.shutdown()


def test_deepseek_unified_matches_oracle(unified_deepseek):
    cfg, sd, prompts, outs, _ = unified_deepseek
    assert all(len(o) == 10 for o in outs)
    frac = check_against_oracle(OracleDeepseekV2(cfg, sd), prompts, outs)
    assert frac > 0.85


def test_deepseek_chunked_prefill_uses_absorbed_extend(unified_deepseek):
    """A prompt longer than chunked_prefill_size is prefetched in chunks: every chunk after the first has
    a prefix, which takes forward_absorb + extend attention over the paged 576-wide latent rows."""
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg, sd, _, _, _ = unified_deepseek
    prompts = make_prompts(cfg.vocab_size, [200, 30, 150], seed=5)
    eng = Engine(server_args(cfg, chunked_prefill_size=64))
    try:
        outs = eng.generate(prompts, SamplingParams(max_new_tokens=6, ignore_eos=True))
    finally:
        eng.shutdown()
    check_against_oracle(OracleDeepseekV2(cfg, sd), prompts, outs)


def test_deepseek_semi_pd(unified_deepseek):
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg, sd, prompts, outs, _ = unified_deepseek
    eng = Engine(server_args(cfg, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50))
    try:
        semi = eng.generate(prompts, SamplingParams(max_new_tokens=10, ignore_eos=True), timeout=300)
    finally:
        eng.shutdown()
    check_against_oracle(OracleDeepseekV2(cfg, sd), prompts, semi)


def test_deepseek_v3_style_routing(device):
    """noaux_tc: sigmoid scores + correction bias + group-limited top-k (biased_grouped_topk)."""
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg = tiny_deepseek(topk_method="noaux_tc", n_group=4, topk_group=2, norm_topk_prob=True,
                        routed_scaling_factor=2.5, num_hidden_layers=2)
    eng = Engine(server_args(cfg))
    try:
        sd = {k: v.float().cpu() for k, v in eng.model_runner.model.state_dict().items()}
        prompts = make_prompts(cfg.vocab_size, [9, 50, 21], seed=8)
        outs = eng.generate(prompts, SamplingParams(max_new_tokens=6, ignore_eos=True))
    finally:
        eng.shutdown()
    check_against_oracle(OracleDeepseekV2(cfg, sd), prompts, outs)


@pytest.mark.parametrize("absorb", ["bmm_fp8", "bf16"])
def test_deepseek_block_fp8_unified_and_semi_pd(device, monkeypatch, absorb):
    """DeepSeek-V3-style block-quantised model: every linear and the routed experts hold fp8
    e4m3fn weights with one fp32 scale per 128 x 128 block, activations are quantised per token and group of
    128 in front of each of them (quantization/fp8.py, fp8_utils.py:91-134, fused_moe.py:526-545).  The dummy
    weights are the block-quantised twin of the bf16 model of the same seed.  Tokens against the oracle, which
    runs the same quantised arithmetic on the CPU; the Semi-PD engine shares the fp8 tensors and their scales
    through IPC and is held to the same oracle.
    absorb = "bmm_fp8" (default): the absorbed MLA products go through input_to_float8 + bmm_fp8 with the
    per-tensor re-quantised W_kc / W_vc (the reference's CUDA path, deepseek_v2.py:659-665, 690-700) and the oracle
    runs that form for its decode steps; "bf16" (SEMIPD_MLA_ABSORB_BF16=1): W_kc / W_vc dequantised at load, torch.bmm
    (the reference's HIP branch), oracle in the exact form."""
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    if absorb == "bf16":
        monkeypatch.setenv("SEMIPD_MLA_ABSORB_BF16", "1")
    else:
        monkeypatch.delenv("SEMIPD_MLA_ABSORB_BF16", raising=False)
    qc = {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "dynamic"}
    cfg = tiny_deepseek(quantization_config=qc)
    prompts = make_prompts(cfg.vocab_size, [5, 37, 130, 1, 64, 17])
    sp = SamplingParams(max_new_tokens=8, ignore_eos=True)
    eng = Engine(server_args(cfg))
    try:
        sd_raw = eng.model_runner.model.state_dict()
        fp8_names = [k for k, v in sd_raw.items() if v.dtype == torch.float8_e4m3fn]
        assert len(fp8_names) >= 3 * 6 and all(k + "_scale_inv" in sd_raw for k in fp8_names)
        assert any("experts.w13_weight" in k for k in fp8_names) and any("kv_a_proj_with_mqa" in k for k in fp8_names)
        assert sd_raw["lm_head.weight"].dtype == torch.bfloat16 and sd_raw["model.layers.1.mlp.gate.weight"].dtype == torch.bfloat16
        sd = {k: v.float().cpu() for k, v in sd_raw.items()}
        outs = eng.generate(prompts, sp)
        # decode through hipGraphs vs eager: the attention split count differs (a different fp32 summation
        # order), behind every attention sits a quantiser, and a seeded random model is full of near-ties, so
        # sequences may part ways after a flipped token; both must hold against the oracle step by step (teacher
        # forced), and the prefill (same kernels, same order) must give the same first token
        eager = Engine(server_args(cfg, disable_cuda_graph=True))
        try:
            eager_outs = eager.generate(prompts, sp)
        finally:
            eager.shutdown()
    finally:
        eng.shutdown()
    # the per-tensor activation scales of the bmm_fp8 form depend on which tokens share a step (as in the reference);
    # the oracle's steps hold all six requests, the engines' mostly do: a slightly wider tie margin covers the rest
    oracle = OracleDeepseekV2(cfg, sd, act_dtype=torch.bfloat16, absorb_fp8=(absorb == "bmm_fp8"))
    margin = 0.12 if absorb == "bf16" else 0.2
    frac = check_against_oracle(oracle, prompts, outs, margin=margin)
    assert frac > 0.8
    check_against_oracle(oracle, prompts, eager_outs, margin=margin)
    assert [o[0] for o in eager_outs] == [o[0] for o in outs]
    semi = Engine(server_args(cfg, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50))
    try:
        got = semi.generate(prompts, sp, timeout=300)
    finally:
        semi.shutdown()
    check_against_oracle(oracle, prompts, got, margin=margin)
    assert [o[0] for o in got] == [o[0] for o in outs]


def test_block_fp8_config_and_loader_rules(device):
    from semi_pd_amd.layers.fp8 import Fp8Config, block_dequantize_weight, block_quantize_weight
    assert Fp8Config.from_hf(None) is None
    assert Fp8Config.from_hf({"quant_method": "fp8", "weight_block_size": [128, 128]}).weight_block_size == (128, 128)
    for bad in ({"quant_method": "awq"}, {"quant_method": "fp8"}, {"quant_method": "fp8", "weight_block_size": [128, 64]},
                {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "static"}):
        with pytest.raises(ValueError):
            Fp8Config.from_hf(bad)
    w = torch.randn(3, 200, 300, device=device) * 0.02
    q, s = block_quantize_weight(w, (128, 128))
    assert q.dtype == torch.float8_e4m3fn and s.shape == (3, 2, 3)
    back = block_dequantize_weight(q, s, (128, 128), torch.float32)
    assert float((back - w).abs().max()) <= float(s.max()) * 16  # half an fp8 step at the top of a block's range
    assert float((back - w).abs().mean() / w.abs().mean()) < 0.03


def test_block_fp8_rejects_ragged_inputs_at_build_time(device):
    """A K that is not a whole number of quantisation groups is an error when the layer is created (DeepSeek-V2-Lite's
    dense FFN, 10944 wide, is such a layer), and an engine that fails to start leaves no process behind."""
    import time
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.layers.basic import RowParallelLinear
    from semi_pd_amd.layers.fp8 import Fp8Config
    with pytest.raises(ValueError, match="not a multiple of the quantisation group"):
        RowParallelLinear(10944, 2048, params_dtype=torch.bfloat16, quant_config=Fp8Config())
    qc = {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "dynamic"}
    cfg = tiny_deepseek(quantization_config=qc, intermediate_size=1000)
    t0 = time.time()
    with pytest.raises(RuntimeError, match="failed to initialise"):
        Engine(server_args(cfg, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50))
    assert time.time() - t0 < 120
    import multiprocessing as mp
    assert not [p for p in mp.active_children() if p.is_alive()]


def test_expert_parallel_partial_outputs_sum_to_the_full_moe(device):
    """--enable-ep-moe at the op level: each "rank" holds E / n whole experts, ids outside its range are dropped
    by moe_align_block_size, its output holds only its experts' contributions; the sum over ranks is the MoE
    (ep_moe/layer.py:190-360).  bf16 and block-fp8."""
    from oracle import ops as O
    from semi_pd_amd.layers.fp8 import block_quantize_weight
    from semi_pd_amd.layers.moe import fused_experts, fused_experts_fp8
    E, topk, K, N, T, ranks = 8, 3, 256, 128, 37, 4
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(T, K, generator=g) / 4).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * N, K, generator=g) * 0.05).to(torch.bfloat16)
    w2 = (torch.randn(E, K, N, generator=g) * 0.05).to(torch.bfloat16)
    tw, ti = torch.topk(torch.softmax(torch.randn(T, E, generator=g), -1), topk)
    want = O.fused_moe(x, w1, w2, tw, ti).float()
    xd, twd, tid = x.to(device), tw.to(device), ti.to(torch.int32).to(device)
    e_local = E // ranks
    total = torch.zeros(T, K, device=device)
    for r in range(ranks):
        sl = slice(r * e_local, (r + 1) * e_local)
        part = fused_experts(xd, w1[sl].to(device), w2[sl].to(device), twd, tid, expert_offset=r * e_local, partial_experts=True)
        total += part.float()
    assert float((total.cpu() - want).abs().max()) < 0.02 * float(want.abs().max())
    q1, s1 = block_quantize_weight(w1.float(), (128, 128))
    q2, s2 = block_quantize_weight(w2.float(), (128, 128))
    want8 = O.fused_moe_block_fp8(x, q1, q2, s1, s2, tw, ti, [128, 128]).float()
    total = torch.zeros(T, K, device=device)
    for r in range(ranks):
        sl = slice(r * e_local, (r + 1) * e_local)
        part = fused_experts_fp8(xd, q1[sl].to(device), q2[sl].to(device), s1[sl].to(device), s2[sl].to(device), twd, tid,
                                 [128, 128], expert_offset=r * e_local, partial_experts=True)
        total += part.float()
    assert float((total.cpu() - want8).abs().max()) < 0.02 * float(want8.abs().max())


def test_deepseek_tp2_expert_parallel_on_one_gpu(unified_deepseek):
    """DeepSeek with TP = 2 and --enable-ep-moe, both ranks on the one GPU (gloo + the peer-memory all-reduce):
    attention and the shared experts are tensor parallel, the routed experts are split 8 + 8 by expert.  Same
    tokens as the oracle of the unsharded model; the plain TP = 2 engine is held to the same."""
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg, sd, prompts, outs, _ = unified_deepseek
    oracle = OracleDeepseekV2(cfg, sd)
    for ep in (True, False):
        eng = Engine(server_args(cfg, tp_size=2, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50,
                                 dist_backend="gloo", enable_ep_moe=ep), gpu_ids={0: 0, 1: 0})
        try:
            got = eng.generate(prompts, SamplingParams(max_new_tokens=10, ignore_eos=True), timeout=600)
        finally:
            eng.shutdown()
        assert all(len(o) == 10 for o in got)
        check_against_oracle(oracle, prompts, got)


def test_deepseek_tp2_expert_all_to_all_on_one_gpu(unified_deepseek):
    """--enable-ep-moe --enable-ep-all-to-all with TP = 2, both ranks on the one GPU (gloo + the peer-memory regions): each
    rank routes its half of the tokens, the rows travel to the rank that owns their expert and back (semipd_ep_dispatch /
    semipd_ep_combine, csrc/all_reduce.hip), the halves are all-gathered; the decode steps run the same collectives from
    their hipGraphs.  Same tokens as the oracle of the unsharded model, like the all-reduce form of expert parallelism
    (test_deepseek_tp2_expert_parallel_on_one_gpu) that it replaces (the reference's only form: ep_moe/layer.py:190)."""
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg, sd, prompts, outs, _ = unified_deepseek
    oracle = OracleDeepseekV2(cfg, sd)
    eng = Engine(server_args(cfg, tp_size=2, enable_semi_pd=True, prefill_cu_percent=50, decode_cu_percent=50,
                             dist_backend="gloo", enable_ep_moe=True, enable_ep_all_to_all=True), gpu_ids={0: 0, 1: 0})
    try:
        got = eng.generate(prompts, SamplingParams(max_new_tokens=10, ignore_eos=True), timeout=600)
    finally:
        eng.shutdown()
    assert all(len(o) == 10 for o in got)
    check_against_oracle(oracle, prompts, got)
