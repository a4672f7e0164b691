"""Batched sampling parameters of one forward (reference: sampling/sampling_batch_info.py:20-160;
penalizers, grammars and custom logit processors are outside the Semi-PD hot path)."""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import torch


@dataclasses.dataclass
class SamplingBatchInfo:
    temperatures: Optional[torch.Tensor]  # fp32 [B, 1]
    top_ps: Optional[torch.Tensor]        # fp32 [B]
    top_ks: Optional[torch.Tensor]        # int32 [B]
    min_ps: Optional[torch.Tensor]        # fp32 [B]
    is_all_greedy: bool
    need_min_p_sampling: bool
    vocab_size: int
    device: str = "cuda"

    @classmethod
    def from_reqs(cls, reqs: List, vocab_size: int, device) -> "SamplingBatchInfo":
        """sampling_batch_info.py:59-160.  A greedy batch (every top_k <= 1) carries no tensors."""
        sps = [r.sampling_params for r in reqs]
        is_all_greedy = all(sp.top_k <= 1 for sp in sps)
        if is_all_greedy:
            return cls(None, None, None, None, True, False, vocab_size, str(device))
        temperatures = torch.tensor([sp.temperature for sp in sps], dtype=torch.float32).view(-1, 1).to(
            device, non_blocking=True)
        top_ps = torch.tensor([sp.top_p for sp in sps], dtype=torch.float32).to(device, non_blocking=True)
        top_ks = torch.tensor([min(sp.top_k, 1 << 30) for sp in sps], dtype=torch.int32).to(
            device, non_blocking=True)
        min_ps = torch.tensor([sp.min_p for sp in sps], dtype=torch.float32).to(device, non_blocking=True)
        return cls(temperatures, top_ps, top_ks, min_ps, False, any(sp.min_p > 0 for sp in sps), vocab_size,
                   str(device))

    def __len__(self):
        return 0 if self.temperatures is None else self.temperatures.shape[0]
