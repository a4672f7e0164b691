"""CPU checks of the host side of the block-fp8 path (no kernels): config parsing, the block quantiser that makes
the seeded dummy weights, scale sharding rules, and the oracle's two fp8 primitives against each other."""
import pytest
import torch

from oracle import ops as O
from semi_pd_amd.layers.fp8 import (Fp8Config, block_dequantize_weight, block_quantize_weight, check_quantisable_input,
                                    scale_shape, shard_rows_of_scale)


def test_fp8_config_rules():
    assert Fp8Config.from_hf(None) is None and Fp8Config.from_hf({}) is None
    cfg = Fp8Config.from_hf({"quant_method": "fp8", "weight_block_size": [128, 128], "fmt": "e4m3"})
    assert cfg.weight_block_size == (128, 128) and cfg.activation_scheme == "dynamic"
    for bad in ({"quant_method": "awq"}, {"quant_method": "fp8"}, {"quant_method": "fp8", "weight_block_size": [128, 64]},
                {"quant_method": "fp8", "weight_block_size": [100, 128]},
                {"quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "static"}):
        with pytest.raises(ValueError):
            Fp8Config.from_hf(bad)
    assert scale_shape(576, 7168, (128, 128)) == (5, 56)
    assert shard_rows_of_scale([(256, 512), (1024, 1280)], 128) == [(2, 4), (8, 10)]
    with pytest.raises(ValueError, match="weight blocks"):
        shard_rows_of_scale([(64, 192)], 128)
    check_quantisable_input(7168, (128, 128), "x")
    with pytest.raises(ValueError, match="quantisation group"):
        check_quantisable_input(10944, (128, 128), "DeepSeek-V2-Lite dense FFN")


def test_block_quantiser_round_trip_and_layout():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(3, 200, 300, generator=g) * 0.02
    q, s = block_quantize_weight(w, (128, 128))
    assert q.dtype == torch.float8_e4m3fn and q.shape == w.shape and s.shape == (3, 2, 3) and s.dtype == torch.float32
    # every block uses the whole fp8 range: its largest code is +-448
    qa = q.float().abs()
    assert float(qa[:, :128, :128].amax()) == 448.0 and float(qa[:, 128:, 256:].amax()) == 448.0
    back = block_dequantize_weight(q, s, (128, 128), torch.float32)
    assert float((back - w).abs().mean() / w.abs().mean()) < 0.03
    assert float((back - w).abs().max()) <= float(s.max()) * 16  # half a step at the top of the range
    z, sz = block_quantize_weight(torch.zeros(128, 128), (128, 128))   # an all-zero block does not divide by zero
    assert float(z.float().abs().max()) == 0.0 and torch.isfinite(sz).all()


def test_oracle_linear_is_the_dequantised_product_up_to_activation_rounding():
    """quant(x) @ block-quantised W through the oracle's two primitives == x @ dequant(W) within fp8 activation
    error; ties the oracle's matmul, the quantiser and the dequantiser together."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(9, 384, generator=g).to(torch.bfloat16)
    w = torch.randn(200, 384, generator=g) * 0.05
    q, s = block_quantize_weight(w, (128, 128))
    xq, xs = O.per_token_group_quant_fp8(x, 128)
    y = O.w8a8_block_fp8_matmul(xq, q, xs, s, [128, 128], torch.float32)
    ref = x.float() @ block_dequantize_weight(q, s, (128, 128), torch.float32).t()
    assert float((y - ref).abs().mean() / ref.abs().mean()) < 0.03
