"""Wire structs between tokenizer / prefill scheduler / decode scheduler / detokenizer.
Field names follow managers/io_struct.py of the reference (TokenizedGenerateReqInput :286-330,
BatchTokenIDOut :380-420, Semi-PD messages :733-755)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional


@dataclass
class SamplingParams:
    """sampling/sampling_params.py:27-110 (the fields the Semi-PD path consumes).  The default
    temperature here is 0 (the benchmark configs are greedy); the HTTP layer applies the reference's
    default of 1.0 when a request does not specify one."""
    max_new_tokens: int = 128
    temperature: float = 0.0
    top_p: float = 1.0
    top_k: int = -1
    min_p: float = 0.0
    ignore_eos: bool = False
    stop_token_ids: Optional[List[int]] = None
    frequency_penalty: float = 0.0   # logits -= f * (times the token was generated)   (penaltylib/frequency_penalty.py)
    presence_penalty: float = 0.0    # logits -= p * (the token was generated at all)  (penaltylib/presence_penalty.py)
    min_new_tokens: int = 0          # stop / EOS tokens cannot be sampled before this many (penaltylib/min_new_tokens.py)

    def __post_init__(self):
        self.verify()
        self.normalize()

    def normalize(self):
        """sampling_params.py:78-85: temperature ~ 0 means greedy == top_k 1; top_k -1 = whole vocab."""
        if self.temperature < 1e-6:
            self.temperature = 1.0
            self.top_k = 1
        if self.top_k == -1:
            self.top_k = 1 << 30

    def verify(self):
        """sampling_params.py:87-110."""
        if self.temperature < 0.0:
            raise ValueError(f"temperature must be non-negative, got {self.temperature}.")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError(f"top_p must be in (0, 1], got {self.top_p}.")
        if not 0.0 <= self.min_p <= 1.0:
            raise ValueError(f"min_p must be in [0, 1], got {self.min_p}.")
        if self.top_k < -1 or self.top_k == 0:
            raise ValueError(f"top_k must be -1 (disable), or at least 1, got {self.top_k}.")
        if self.max_new_tokens is not None and self.max_new_tokens < 0:
            raise ValueError(f"max_new_tokens must be at least 0, got {self.max_new_tokens}.")
        if not -2.0 <= self.frequency_penalty <= 2.0:
            raise ValueError(f"frequency_penalty must be in [-2, 2], got {self.frequency_penalty}.")
        if not -2.0 <= self.presence_penalty <= 2.0:
            raise ValueError(f"presence_penalty must be in [-2, 2], got {self.presence_penalty}.")
        if self.min_new_tokens < 0:
            raise ValueError(f"min_new_tokens must be in (0, max_new_tokens], got {self.min_new_tokens}.")
        if self.max_new_tokens is not None and self.min_new_tokens > self.max_new_tokens:
            raise ValueError(f"min_new_tokens must be in (0, max_new_tokens({self.max_new_tokens})], "
                             f"got {self.min_new_tokens}.")

    @property
    def is_greedy(self) -> bool:
        return self.top_k <= 1

    @property
    def needs_penalties(self) -> bool:
        """True when the logits of this request depend on its own generated tokens (penaltylib/orchestrator.py
        is_required): the overlapped decode loop must then have every earlier token on the host first."""
        return self.frequency_penalty != 0.0 or self.presence_penalty != 0.0 or self.min_new_tokens > 0


@dataclass
class TokenizedGenerateReqInput:
    rid: str
    input_text: Optional[str]
    input_ids: List[int]
    sampling_params: SamplingParams
    stream: bool = True
    return_logprob: bool = False
    top_logprobs_num: int = 0
    is_retracted: bool = False  # Semi-PD: re-sent to P after a decode retraction
    # how many of the LAST input_ids of a retracted request are tokens it generated (penalties count those; the
    # reference drops this history on the prefill instance)
    retracted_output_len: int = 0


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
    # KV slots of the tokens to prefill, per request.  The reference has P read them from the shared req_to_token
    # table, which needs the table writes to be complete in HBM first (a GPU synchronisation on the decode side, one
    # whole decode step long when the loop is overlapped); the decode instance knows them on the host, so they
    # travel with the reply and the table is only needed for prefix tokens (chunked prefill, re-sent requests).
    extend_slots: Optional[List[List[int]]] = None


@dataclass
class BatchProcessPrefillResultReq:
    """P -> D after the prefill forward (io_struct.py:749-755).  next_token_logits is only filled
    when a request needs logits on the decode side (never for greedy)."""
    next_token_ids: List[int]
    next_token_logits: Optional[Any] = None
    # per request: logprob of the sampled first token (+ top-k) when the request asked for logprobs
    next_token_logprobs: Optional[dict] = None


@dataclass
class BatchTokenIDOut:
    """D -> detokenizer / client: newly produced token ids per request."""
    rids: List[str]
    finished_reasons: List[Optional[str]]
    output_ids: List[List[int]]          # tokens emitted since the previous message
    timestamps: List[float] = field(default_factory=list)
    # aligned with output_ids for requests with return_logprob, else None entries
    output_token_logprobs: Optional[List[Optional[List[float]]]] = None
    output_top_logprobs: Optional[List[Optional[List[List[tuple]]]]] = None


@dataclass
class AbortReq:
    """Client went away or a stop string matched (io_struct.py AbortReq; scheduler.py:1565-1584)."""
    rid: str


@dataclass
class FlushCacheReq:
    pass


@dataclass
class ShutdownReq:
    pass


@dataclass
class SyntheticLoadReq:
    """Prefill instance -> decode instance at start-up: replay a captured decode step in a loop while idle (on) /
    stop (off).  The prefill instance times its GEMM candidates next to that stream of weight reads, i.e. under the
    HBM contention they will serve in (entrypoints/engine.py)."""
    on: bool = True


@dataclass
class StatsReq:
    """Ask a scheduler for its counters (kernel timing samples, steps, tokens)."""
    reset: bool = False
