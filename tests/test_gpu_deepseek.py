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
