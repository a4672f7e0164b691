"""Block-scaled fp8 (w8a8) linear method: the host side of csrc/fp8_gemm.hip.

Reference: layers/quantization/fp8.py (`Fp8Config` with `weight_block_size`, `Fp8LinearMethod.create_weights`
:205-330 — fp8 `weight` [out, in] plus fp32 `weight_scale_inv` [ceil(out/bn), ceil(in/bk)] —, `Fp8MoEMethod`
:470-620 — `w13_weight_scale_inv`, `w2_weight_scale_inv`), layers/quantization/fp8_utils.py:91-134
(`apply_w8a8_block_fp8_linear`: quantise the activations per token and group of block_k, block matmul, bias).
The fp8 format is OCP e4m3fn (gfx950); checkpoints of DeepSeek-V3 are stored in it.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import torch

from semi_pd_amd import ops

FP8_DTYPE = ops.FP8_DTYPE
FP8_MAX = 448.0


@dataclasses.dataclass(frozen=True)
class Fp8Config:
    """`quantization_config` of a block-quantised checkpoint (fp8.py:60-130)."""
    weight_block_size: tuple = (128, 128)
    activation_scheme: str = "dynamic"

    @classmethod
    def from_hf(cls, d: Optional[dict]) -> Optional["Fp8Config"]:
        if not d:
            return None
        if d.get("quant_method") != "fp8":
            raise ValueError(f"unsupported quantization method {d.get('quant_method')!r}: only block-scaled fp8")
        block = d.get("weight_block_size")
        if not block:
            raise ValueError("fp8 checkpoints without weight_block_size (per-tensor scales) are not supported")
        if d.get("activation_scheme", "dynamic") != "dynamic":
            raise ValueError("block-wise fp8 needs dynamic activation quantisation (fp8.py:100-104)")
        if int(block[1]) != 128 or int(block[0]) % 16 != 0:
            raise ValueError(f"weight_block_size {block}: the kernels need block_k = 128 and block_n % 16 == 0")
        return cls(weight_block_size=(int(block[0]), int(block[1])))


def scale_shape(rows: int, cols: int, block) -> tuple:
    return (-(-rows // block[0]), -(-cols // block[1]))


def check_quantisable_input(input_size: int, block, what: str) -> None:
    """The activations in front of a block-quantised layer are quantised in whole groups of block_k
    (per_token_group_quant_fp8 asserts it, fp8_kernel.py:183-186); say so when the model is built, not in the
    middle of the first forward pass."""
    if input_size % int(block[1]):
        raise ValueError(f"{what}: input size {input_size} is not a multiple of the quantisation group {block[1]}")


def quantize_activation(x: torch.Tensor, block_size):
    """The per-token-group quantisation in front of a block-fp8 layer, as a value: layers that read the SAME
    activation (q_proj and kv_a_proj_with_mqa; the shared experts and the routed experts) quantise it once and pass
    the result on (`x_quant=`), instead of once each as apply_w8a8_block_fp8_linear does (fp8_utils.py:91-134)."""
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    return ops.per_token_group_quant_fp8(x2, int(block_size[1]))


def apply_w8a8_block_fp8_linear(x: torch.Tensor, weight: torch.Tensor, block_size, weight_scale: torch.Tensor,
                                bias: Optional[torch.Tensor] = None, x_quant=None) -> torch.Tensor:
    """fp8_utils.py:91-134.  x_quant = quantize_activation(x, block_size) computed by the caller, if it has it."""
    q, s = x_quant if x_quant is not None else quantize_activation(x, block_size)
    out = ops.w8a8_block_fp8_matmul(q, weight, s, weight_scale, block_size, output_dtype=x.dtype)
    if bias is not None:
        out = out + bias
    return out.view(*x.shape[:-1], weight.shape[0])


def block_quantize_weight(w: torch.Tensor, block) -> tuple:
    """[..., N, K] float -> (fp8 weight, fp32 scale [..., ceil(N/bn), ceil(K/bk)]) with scale = absmax of the tile /
    448.  How block-quantised checkpoints are produced (DeepSeek-V3's inference/fp8_cast); used here for the
    seeded dummy weights and by tests."""
    bn, bk = int(block[0]), int(block[1])
    *lead, N, K = w.shape
    nb, kb = -(-N // bn), -(-K // bk)
    wp = torch.zeros(*lead, nb * bn, kb * bk, dtype=torch.float32, device=w.device)
    wp[..., :N, :K] = w.float()
    tiles = wp.view(*lead, nb, bn, kb, bk)
    scale = tiles.abs().amax(dim=(-3, -1)).clamp(min=1e-12) / FP8_MAX
    q = (tiles / scale[..., :, None, :, None]).clamp(-FP8_MAX, FP8_MAX).view(*lead, nb * bn, kb * bk)[..., :N, :K]
    return q.contiguous().to(FP8_DTYPE), scale.contiguous()


def block_dequantize_weight(q: torch.Tensor, scale: torch.Tensor, block, dtype=torch.bfloat16) -> torch.Tensor:
    """Inverse of block_quantize_weight (the reference dequantises kv_b_proj for the absorbed MLA matrices,
    deepseek_v2.py:1195-1209)."""
    bn, bk = int(block[0]), int(block[1])
    *lead, N, K = q.shape
    s = scale.repeat_interleave(bn, dim=-2)[..., :N, :].repeat_interleave(bk, dim=-1)[..., :K]
    return (q.float() * s).to(dtype)


def block_quant_to_tensor_quant(q: torch.Tensor, scale: torch.Tensor, block):
    """Block-wise quantised weight -> ONE scale for the whole tensor (fp8_utils.py:152-188: dequantise in fp32,
    input_to_float8): what the reference feeds bmm_fp8 with for the absorbed MLA matrices (deepseek_v2.py
    :1195-1209).  Load-time only: plain torch on the weight's device.  Returns (fp8 tensor, 1 / scale as fp32)."""
    bn, bk = int(block[0]), int(block[1])
    N, K = q.shape
    s = scale.repeat_interleave(bn, dim=0)[:N].repeat_interleave(bk, dim=1)[:, :K]
    x = q.to(torch.float32) * s
    fmax = torch.finfo(q.dtype).max
    min_val, max_val = x.aminmax()
    amax = torch.maximum(min_val.abs(), max_val.abs()).clamp(min=1e-12)
    sc = fmax / amax
    return (x * sc).clamp(min=-fmax, max=fmax).to(q.dtype).contiguous(), sc.float().reciprocal()


def shard_rows_of_scale(ranges: List[tuple], block_n: int):
    """Row ranges of a weight -> the same cut on its scale rows; every boundary must sit on a block edge
    (fp8.py:228-244 raises for partitions that are not multiples of block_n)."""
    out = []
    for a, b in ranges:
        if a % block_n or b % block_n:
            raise ValueError(f"tensor-parallel shard rows [{a}, {b}) do not sit on weight blocks of {block_n} rows")
        out.append((a // block_n, b // block_n))
    return out
