"""`--load-format auto` on CPU: a seeded HF model saved with save_pretrained (config.json +
safetensors) must load into the product model exactly as the oracle's HF-name conversion lays it
out (reference: model_loader/loader.py + models/*.py load_weights stacked-parameter mappings)."""
import pytest
import torch

transformers = pytest.importorskip("transformers")
pytest.importorskip("safetensors")

from oracle.hf_convert import deepseek_v2_from_hf, llama_from_hf, opt_from_hf
from semi_pd_amd.model_loader import (config_from_hf_dict, load_hf_config, load_weights,
                                      safetensors_weights_iterator)
from semi_pd_amd.model_executor.model_runner import build_model


def _load(tmp_path, hf):
    hf.save_pretrained(tmp_path, safe_serialization=True)
    cfg = load_hf_config(str(tmp_path))
    model = build_model(cfg, torch.float32)
    loaded = load_weights(model, cfg, safetensors_weights_iterator(str(tmp_path)))
    assert len(loaded) == len(set(loaded))
    return cfg, model


def _compare(model, want):
    got = dict(model.named_parameters())
    for k, w in want.items():
        g = got[k]
        if g.shape != w.shape:  # vocab rows padded to a multiple of 64 with zeros
            assert g.shape[0] >= w.shape[0] and torch.count_nonzero(g[w.shape[0]:]) == 0
            g = g[: w.shape[0]]
        assert torch.equal(g, w.to(g.dtype)), k
    assert set(got) <= set(want) | {"lm_head.weight"}


def test_llama_checkpoint(tmp_path):
    torch.manual_seed(0)
    hf = transformers.LlamaForCausalLM(transformers.LlamaConfig(
        vocab_size=320, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
        num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5, rope_theta=10000.0,
        tie_word_embeddings=False))
    cfg, model = _load(tmp_path, hf)
    assert (cfg.num_key_value_heads, cfg.intermediate_size, cfg.vocab_size) == (2, 96, 320)
    _compare(model, llama_from_hf(hf.state_dict(), 2))


def test_opt_checkpoint(tmp_path):
    torch.manual_seed(1)
    hf = transformers.OPTForCausalLM(transformers.OPTConfig(
        vocab_size=272, hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4,
        max_position_embeddings=128, word_embed_proj_dim=64, do_layer_norm_before=True))
    cfg, model = _load(tmp_path, hf)
    assert cfg.ffn_dim == 128
    _compare(model, opt_from_hf(hf.state_dict(), 2))


def test_deepseek_v2_checkpoint(tmp_path):
    torch.manual_seed(2)
    kw = dict(vocab_size=300, hidden_size=64, intermediate_size=96, moe_intermediate_size=32, num_hidden_layers=3,
              num_attention_heads=4, n_shared_experts=2, n_routed_experts=8, num_experts_per_tok=3, kv_lora_rank=32,
              q_lora_rank=None, qk_rope_head_dim=16, qk_nope_head_dim=16, v_head_dim=16, first_k_dense_replace=1,
              n_group=1, topk_group=1, topk_method="greedy", norm_topk_prob=False, routed_scaling_factor=1.0,
              max_position_embeddings=256, rms_norm_eps=1e-6)
    hf = transformers.DeepseekV2ForCausalLM(transformers.DeepseekV2Config(num_key_value_heads=4, **kw))
    cfg, model = _load(tmp_path, hf)
    assert cfg.n_routed_experts == 8 and cfg.kv_lora_rank == 32
    _compare(model, deepseek_v2_from_hf(hf.state_dict(), cfg))


def test_per_expert_names_are_stacked():
    """Hub checkpoints store experts one by one (models/deepseek_v2.py:1170-1200 expert_params_mapping)."""
    from semi_pd_amd.model_loader import product_items
    from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
    cfg = DeepseekV2Config(n_routed_experts=3, moe_intermediate_size=4, hidden_size=6)
    g = torch.Generator().manual_seed(0)
    ws = {}
    for e in (2, 0, 1):
        for which, shape in (("gate_proj", (4, 6)), ("up_proj", (4, 6)), ("down_proj", (6, 4))):
            ws[f"model.layers.1.mlp.experts.{e}.{which}.weight"] = torch.randn(shape, generator=g)
    out = dict(product_items(cfg, ws.items()))
    assert set(out) == {"model.layers.1.mlp.experts.w13_weight", "model.layers.1.mlp.experts.w2_weight"}
    w13 = out["model.layers.1.mlp.experts.w13_weight"]
    assert w13.shape == (3, 8, 6)
    assert torch.equal(w13[1, :4], ws["model.layers.1.mlp.experts.1.gate_proj.weight"])
    assert torch.equal(w13[2, 4:], ws["model.layers.1.mlp.experts.2.up_proj.weight"])
    assert torch.equal(out["model.layers.1.mlp.experts.w2_weight"][0], ws["model.layers.1.mlp.experts.0.down_proj.weight"])


def test_unsupported_architecture_is_rejected():
    with pytest.raises(ValueError):
        config_from_hf_dict({"architectures": ["GPT2LMHeadModel"], "hidden_size": 8})


def test_block_fp8_checkpoint_names_and_flags(tmp_path):
    """Block-quantised checkpoints (DeepSeek-V3): `weight_scale_inv` tensors travel with the fp8 weights and
    stack the same way (gate/up -> gate_up_proj, experts.{e}.* -> w13 / w2); `quantization_config` is read
    from config.json; --quantization fp8 must agree with it."""
    import argparse
    import json
    from semi_pd_amd.model_loader import product_items
    from semi_pd_amd.models.deepseek_v2 import DeepseekV2Config
    from semi_pd_amd.server_args import add_cli_args, from_cli_args
    cfg = DeepseekV2Config(n_routed_experts=2, moe_intermediate_size=256, hidden_size=256)
    g = torch.Generator().manual_seed(0)
    ws = {}
    for e in (1, 0):
        for which, shape in (("gate_proj", (256, 256)), ("up_proj", (256, 256)), ("down_proj", (256, 256))):
            ws[f"model.layers.1.mlp.experts.{e}.{which}.weight"] = torch.randn(shape, generator=g).to(torch.float8_e4m3fn)
            ws[f"model.layers.1.mlp.experts.{e}.{which}.weight_scale_inv"] = torch.rand(2, 2, generator=g)
    for which in ("gate_proj", "up_proj"):
        ws[f"model.layers.0.mlp.{which}.weight"] = torch.randn(256, 256, generator=g).to(torch.float8_e4m3fn)
        ws[f"model.layers.0.mlp.{which}.weight_scale_inv"] = torch.rand(2, 2, generator=g)
    out = dict(product_items(cfg, ws.items()))
    assert set(out) == {"model.layers.1.mlp.experts.w13_weight", "model.layers.1.mlp.experts.w2_weight",
                        "model.layers.1.mlp.experts.w13_weight_scale_inv", "model.layers.1.mlp.experts.w2_weight_scale_inv",
                        "model.layers.0.mlp.gate_up_proj.weight", "model.layers.0.mlp.gate_up_proj.weight_scale_inv"}
    s13 = out["model.layers.1.mlp.experts.w13_weight_scale_inv"]
    assert s13.shape == (2, 4, 2) and out["model.layers.1.mlp.experts.w13_weight"].dtype == torch.float8_e4m3fn
    assert torch.equal(s13[1, 2:], ws["model.layers.1.mlp.experts.1.up_proj.weight_scale_inv"])
    assert torch.equal(out["model.layers.0.mlp.gate_up_proj.weight_scale_inv"][:2],
                       ws["model.layers.0.mlp.gate_proj.weight_scale_inv"])

    base = {"architectures": ["DeepseekV3ForCausalLM"], "vocab_size": 320, "hidden_size": 256, "intermediate_size": 256,
            "moe_intermediate_size": 128, "num_hidden_layers": 2, "num_attention_heads": 4, "n_routed_experts": 4,
            "num_experts_per_tok": 2, "kv_lora_rank": 128, "max_position_embeddings": 256}
    qc = {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "dynamic", "fmt": "e4m3"}
    parse = lambda argv: from_cli_args(add_cli_args(argparse.ArgumentParser()).parse_args(argv))  # noqa: E731
    (tmp_path / "config.json").write_text(json.dumps(dict(base, quantization_config=qc)))
    assert parse(["--model-path", str(tmp_path)]).model_config.quantization_config["weight_block_size"] == [128, 128]
    assert parse(["--model-path", str(tmp_path), "--quantization", "fp8"]).model_config.quantization_config == qc
    (tmp_path / "config.json").write_text(json.dumps(base))
    sa = parse(["--model-path", str(tmp_path), "--quantization", "fp8", "--load-format", "dummy"])
    assert sa.model_config.quantization_config["quant_method"] == "fp8"          # dummy weights: quantised on the fly
    with pytest.raises(ValueError, match="block-quantised checkpoint"):
        parse(["--model-path", str(tmp_path), "--quantization", "fp8"])           # a bf16 checkpoint is not quantised on load
    (tmp_path / "config.json").write_text(json.dumps({
        "architectures": ["LlamaForCausalLM"], "vocab_size": 320, "hidden_size": 64, "intermediate_size": 96,
        "num_hidden_layers": 2, "num_attention_heads": 4, "num_key_value_heads": 2, "quantization_config": qc}))
    with pytest.raises(ValueError, match="DeepSeek family"):
        parse(["--model-path", str(tmp_path)])


def test_behaviour_changing_config_keys_are_rejected(tmp_path):
    """A Llama checkpoint with attention / MLP biases (or another activation) must not load as if it had none."""
    base = dict(architectures=["LlamaForCausalLM"], vocab_size=320, hidden_size=64, intermediate_size=96,
                num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2)
    config_from_hf_dict(dict(base, attention_bias=False, mlp_bias=False, hidden_act="silu"))
    for extra in (dict(attention_bias=True), dict(mlp_bias=True), dict(hidden_act="gelu")):
        with pytest.raises(ValueError):
            config_from_hf_dict(dict(base, **extra))
    # and a stray tensor that matches no parameter is an error, not a silent skip
    torch.manual_seed(0)
    hf = transformers.LlamaForCausalLM(transformers.LlamaConfig(
        vocab_size=320, hidden_size=64, intermediate_size=96, num_hidden_layers=1, num_attention_heads=4,
        num_key_value_heads=2, max_position_embeddings=64, tie_word_embeddings=False))
    hf.save_pretrained(tmp_path, safe_serialization=True)
    cfg = load_hf_config(str(tmp_path))
    model = build_model(cfg, torch.float32)
    weights = list(safetensors_weights_iterator(str(tmp_path))) + [("model.layers.0.self_attn.o_proj.bias", torch.zeros(64))]
    with pytest.raises(KeyError):
        load_weights(model, cfg, weights)
