"""Pin the CPU model oracles (oracle/model.py) against independent implementations: HF
LlamaForCausalLM / OPTForCausalLM with seeded random weights (BASELINE.md §3 cross-check), tolerance
5e-2 on logits and identical greedy ids (test/srt/models/test_generation_models.py:43-45)."""
import pytest
import torch

transformers = pytest.importorskip("transformers")

from oracle.hf_convert import llama_from_hf, opt_from_hf
from oracle.model import OracleLlama, OracleOPT
from semi_pd_amd.models.llama import LlamaConfig
from semi_pd_amd.models.opt import OPTConfig


def _prompts(vocab):
    g = torch.Generator().manual_seed(7)
    return [torch.randint(0, vocab, (n,), generator=g).tolist() for n in (5, 17, 9)]


def _hf_greedy(model, prompt, n):
    ids = torch.tensor([prompt])
    logits_all = []
    with torch.no_grad():
        for _ in range(n):
            lg = model(ids).logits[0, -1].float()
            logits_all.append(lg)
            ids = torch.cat([ids, lg.argmax().view(1, 1)], 1)
    return ids[0, len(prompt):].tolist(), torch.stack(logits_all)


@pytest.mark.parametrize("rope_scaling", [None, {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                                 "high_freq_factor": 4.0, "original_max_position_embeddings": 64}])
def test_oracle_llama_matches_hf(rope_scaling):
    torch.manual_seed(0)
    hf_cfg = transformers.LlamaConfig(vocab_size=320, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                                      num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256,
                                      rms_norm_eps=1e-5, rope_theta=10000.0, rope_scaling=rope_scaling,
                                      tie_word_embeddings=False)
    hf = transformers.LlamaForCausalLM(hf_cfg).eval()
    cfg = LlamaConfig(vocab_size=320, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0,
                      rope_scaling=rope_scaling, max_position_embeddings=256)
    oracle = OracleLlama(cfg, llama_from_hf(hf.state_dict(), 2))
    prompts = _prompts(320)
    toks, logits = oracle.generate(prompts, 6)
    for b, p in enumerate(prompts):
        want_toks, want_logits = _hf_greedy(hf, p, 6)
        assert toks[b] == want_toks
        torch.testing.assert_close(logits[b], want_logits, rtol=1e-4, atol=1e-4)


def test_oracle_opt_matches_hf():
    torch.manual_seed(1)
    hf_cfg = transformers.OPTConfig(vocab_size=272, hidden_size=64, ffn_dim=128, num_hidden_layers=2,
                                    num_attention_heads=4, max_position_embeddings=128, word_embed_proj_dim=64,
                                    do_layer_norm_before=True)
    hf = transformers.OPTForCausalLM(hf_cfg).eval()
    cfg = OPTConfig(vocab_size=272, hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4,
                    max_position_embeddings=128)
    oracle = OracleOPT(cfg, opt_from_hf(hf.state_dict(), 2))
    prompts = _prompts(272)
    toks, logits = oracle.generate(prompts, 6)
    for b, p in enumerate(prompts):
        want_toks, want_logits = _hf_greedy(hf, p, 6)
        assert toks[b] == want_toks
        torch.testing.assert_close(logits[b], want_logits, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("q_lora_rank", [None, 24])
def test_oracle_deepseek_v2_matches_hf(q_lora_rank):
    from oracle.hf_convert import deepseek_v2_from_hf
    from oracle.model import OracleDeepseekV2
    from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
    torch.manual_seed(2)
    kw = dict(vocab_size=300, hidden_size=64, intermediate_size=96, moe_intermediate_size=32, num_hidden_layers=3,
              num_attention_heads=4, n_shared_experts=2, n_routed_experts=8, num_experts_per_tok=3, kv_lora_rank=32,
              q_lora_rank=q_lora_rank, qk_rope_head_dim=16, qk_nope_head_dim=16, v_head_dim=16, first_k_dense_replace=1,
              n_group=1, topk_group=1, topk_method="greedy", norm_topk_prob=False, routed_scaling_factor=1.0,
              max_position_embeddings=256, rms_norm_eps=1e-6)
    hf = transformers.DeepseekV2ForCausalLM(transformers.DeepseekV2Config(num_key_value_heads=4, **kw)).eval()
    cfg = DeepseekV2Config(rope_scaling=None, rope_theta=10000.0, **kw)
    oracle = OracleDeepseekV2(cfg, deepseek_v2_from_hf(hf.state_dict(), cfg))
    prompts = _prompts(300)
    toks, logits = oracle.generate(prompts, 5)
    for b, p in enumerate(prompts):
        want_toks, want_logits = _hf_greedy(hf, p, 5)
        assert toks[b] == want_toks
        torch.testing.assert_close(logits[b], want_logits, rtol=2e-4, atol=2e-4)
