"""Wire structs between tokenizer / prefill scheduler / decode scheduler / detokenizer.
Field names follow managers/io_struct.py of the reference (TokenizedGenerateReqInput :286-330,
BatchTokenIDOut :380-420, Semi-PD messages :733-755)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional


@dataclass
class SamplingParams:
    """Subset of sampling/sampling_params.py used by the greedy configs."""
    max_new_tokens: int = 128
    temperature: float = 0.0
    ignore_eos: bool = False
    stop_token_ids: Optional[List[int]] = None

    @property
    def is_greedy(self) -> bool:
        return self.temperature == 0.0


@dataclass
class TokenizedGenerateReqInput:
    rid: str
    input_text: Optional[str]
    input_ids: List[int]
    sampling_params: SamplingParams
    stream: bool = True
    return_logprob: bool = False
    is_retracted: bool = False  # Semi-PD: re-sent to P after a decode retraction


@dataclass
class GetNextPrefillBatchInput:
    """P -> D: "which of these requests may I prefill?" (io_struct.py:733-736)."""
    rids: List[str]


@dataclass
class GetNextPrefillBatchOutput:
    """D -> P over the bridge socket (io_struct.py:739-746)."""
    rids: List[str]
    chunked_rid: Optional[str]
    req_pool_indices: List[int]
    prefix_lens: List[int]
    extend_input_lens: List[int]


@dataclass
class BatchProcessPrefillResultReq:
    """P -> D after the prefill forward (io_struct.py:749-755).  next_token_logits is only filled
    when a request needs logits on the decode side (never for greedy)."""
    next_token_ids: List[int]
    next_token_logits: Optional[Any] = None


@dataclass
class BatchTokenIDOut:
    """D -> detokenizer / client: newly produced token ids per request."""
    rids: List[str]
    finished_reasons: List[Optional[str]]
    output_ids: List[List[int]]          # tokens emitted since the previous message
    timestamps: List[float] = field(default_factory=list)


@dataclass
class FlushCacheReq:
    pass


@dataclass
class ShutdownReq:
    pass


@dataclass
class StatsReq:
    """Ask a scheduler for its counters (kernel timing samples, steps, tokens)."""
    reset: bool = False
