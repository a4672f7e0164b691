"""Batched sampling parameters of one forward (reference: sampling/sampling_batch_info.py:20-160;
grammars and custom logit processors are outside the Semi-PD hot path).

Penalties (sampling/penaltylib/*): the reference keeps dense [B, vocab] fp32 tensors per penalizer, updated by
scatter with every step and filtered / concatenated with the batch.  The values are a pure function of each
request's generated tokens, which the scheduler has on the host, so here they are rebuilt per step as a SPARSE list
of (row, token, value) entries and subtracted in place from the fp32 logits: a few hundred entries instead of
B x vocab floats, and nothing to filter or merge."""
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
    # sparse penalties: logits[pen_rows[i], pen_toks[i]] -= pen_vals[i]; None when no request asks for any
    pen_rows: Optional[torch.Tensor] = None   # int64 [n]
    pen_toks: Optional[torch.Tensor] = None   # int64 [n]
    pen_vals: Optional[torch.Tensor] = None   # fp32 [n] (+inf bans a token)

    @classmethod
    def from_reqs(cls, reqs: List, vocab_size: int, device) -> "SamplingBatchInfo":
        """sampling_batch_info.py:59-160.  A greedy batch (every top_k <= 1) carries no tensors."""
        sps = [r.sampling_params for r in reqs]
        is_all_greedy = all(sp.top_k <= 1 for sp in sps)
        pen = cls.penalty_entries(reqs, vocab_size, device) if any(sp.needs_penalties for sp in sps) else (None,) * 3
        if is_all_greedy:
            return cls(None, None, None, None, True, False, vocab_size, str(device), *pen)
        temperatures = torch.tensor([sp.temperature for sp in sps], dtype=torch.float32).view(-1, 1).to(
            device, non_blocking=True)
        top_ps = torch.tensor([sp.top_p for sp in sps], dtype=torch.float32).to(device, non_blocking=True)
        top_ks = torch.tensor([min(sp.top_k, 1 << 30) for sp in sps], dtype=torch.int32).to(
            device, non_blocking=True)
        min_ps = torch.tensor([sp.min_p for sp in sps], dtype=torch.float32).to(device, non_blocking=True)
        return cls(temperatures, top_ps, top_ks, min_ps, False, any(sp.min_p > 0 for sp in sps), vocab_size,
                   str(device), *pen)

    @staticmethod
    def penalty_entries(reqs: List, vocab_size: int, device):
        """One entry per (request, distinct generated token): frequency_penalty * count + presence_penalty
        (frequency_penalty.py:49-57, presence_penalty.py:49-57: output tokens only, never the prompt), and +inf
        for every stop / EOS token of a request that has generated fewer than min_new_tokens tokens
        (min_new_tokens.py:36-79)."""
        from collections import Counter
        rows, toks, vals = [], [], []
        limit = vocab_size if vocab_size > 0 else 1 << 62   # unknown here: apply_penalties guards against the width
        for i, r in enumerate(reqs):
            sp = r.sampling_params
            generated = r.output_ids
            k = getattr(r, "retracted_output_len", 0)
            if k:  # prefill instance, request re-sent after a retraction: its generated tokens end the prompt
                generated = r.origin_input_ids[max(0, len(r.origin_input_ids) - k):] + r.output_ids
            if (sp.frequency_penalty != 0.0 or sp.presence_penalty != 0.0) and generated:
                for t, c in Counter(generated).items():
                    if 0 <= t < limit:
                        rows.append(i), toks.append(t), vals.append(sp.frequency_penalty * c + sp.presence_penalty)
            if len(generated) < sp.min_new_tokens:
                for t in set(sp.stop_token_ids or ()) | set(getattr(r, "eos_token_ids", None) or ()):
                    if 0 <= t < limit:
                        rows.append(i), toks.append(t), vals.append(float("inf"))
        if not rows:
            return None, None, None
        from semi_pd_amd.managers.schedule_batch import host_list_to_device
        return (host_list_to_device(rows, torch.int64, device), host_list_to_device(toks, torch.int64, device),
                host_list_to_device(vals, torch.float32, device))

    @property
    def has_penalties(self) -> bool:
        return self.pen_rows is not None

    def apply_penalties(self, logits: torch.Tensor) -> None:
        """sampling_batch_info.py:188-191 apply_logits_bias: in place on the fp32 logits, before temperature.
        (row, token) pairs are distinct per cause and accumulate, so the result does not depend on the order."""
        if self.pen_rows is not None:
            # a user-supplied stop id beyond the vocabulary must not become an out-of-bounds write: such entries
            # are pointed at the last column with value 0 (no host round trip, the stream is not synchronised)
            width = logits.shape[1]
            inside = self.pen_toks < width
            toks = torch.where(inside, self.pen_toks, torch.full_like(self.pen_toks, width - 1))
            vals = torch.where(inside, self.pen_vals, torch.zeros_like(self.pen_vals))
            logits.index_put_((self.pen_rows, toks), -vals, accumulate=True)

    def __len__(self):
        return 0 if self.temperatures is None else self.temperatures.shape[0]
