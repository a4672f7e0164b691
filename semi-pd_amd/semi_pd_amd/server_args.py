"""ServerArgs / SemiPDPortArgs: the flag subset of the reference that the Semi-PD hot path reads
(server_args.py:146 enable_semi_pd, :225-236 mem fraction / chunked prefill defaults, :325-331
radix off, :870-875 CLI flag, :1117-1195 SemiPDPortArgs)."""
from __future__ import annotations

import dataclasses
import os
import tempfile
from typing import Any, List, Optional

import torch

from semi_pd_amd.semi_pd.utils import DECODE_ENGINE_SM_PERCENTILE, PREFILL_ENGINE_SM_PERCENTILE


@dataclasses.dataclass
class ServerArgs:
    model_config: Any = None                 # LlamaConfig / OPTConfig / DeepseekV2Config (dummy weights)
    load_format: str = "dummy"
    dtype: str = "bfloat16"
    context_length: int = 4096
    tp_size: int = 1
    base_gpu_id: int = 0
    mem_fraction_static: Optional[float] = None
    max_running_requests: int = 256
    max_total_tokens: Optional[int] = None
    chunked_prefill_size: int = 8192
    max_prefill_tokens: int = 16384
    schedule_conservativeness: float = 1.0
    enable_semi_pd: bool = False
    disable_radix_cache: bool = True
    disable_cuda_graph: bool = False
    cuda_graph_max_bs: int = 256
    attention_backend: str = "hip"
    sampling_backend: str = "hip"
    watchdog_timeout: float = 300.0
    random_seed: int = 0
    eos_token_ids: Optional[List[int]] = None
    # Semi-PD compute split (semi_pd/utils.py:10-11 env knobs; BASELINE config 2 asks 50/50)
    prefill_cu_percent: int = PREFILL_ENGINE_SM_PERCENTILE
    decode_cu_percent: int = DECODE_ENGINE_SM_PERCENTILE
    cu_mask_mode: str = "env"                # "env" (process-wide HSA_CU_MASK) | "none"
    dist_init_addr: str = "127.0.0.1"
    nccl_port_base: Optional[int] = None
    dist_backend: str = "nccl"               # "nccl" = RCCL over xGMI; "gloo" only for single-GPU TP tests
    collect_kernel_timing: bool = False

    def __post_init__(self):
        if self.mem_fraction_static is None:
            self.mem_fraction_static = 0.88 if self.tp_size == 1 else 0.85
            if self.enable_semi_pd:
                # two processes keep activations resident (server_args.py:225-226)
                self.mem_fraction_static *= 0.9
        if self.enable_semi_pd:
            self.disable_radix_cache = True  # server_args.py:325-331

    @property
    def torch_dtype(self) -> torch.dtype:
        return {"bfloat16": torch.bfloat16, "float16": torch.float16, "half": torch.float16}[self.dtype]


@dataclasses.dataclass
class SemiPDPortArgs:
    """Socket names of one Semi-PD engine (server_args.py:1117-1161)."""
    tokenizer_ipc_name: str                # D -> client (token stream); also stats replies
    p_scheduler_input_ipc_name: str        # client / D -> P
    d_scheduler_input_ipc_name: str        # client / P -> D
    bridge_ipc_name: str                   # D -> P replies to GetNextPrefillBatchInput
    p_nccl_port: int
    d_nccl_port: int

    @staticmethod
    def init_new(server_args: ServerArgs, base_dir: Optional[str] = None) -> "SemiPDPortArgs":
        d = base_dir or tempfile.mkdtemp(prefix="semipd_")
        base = server_args.nccl_port_base or (20000 + (os.getpid() * 7) % 20000)
        return SemiPDPortArgs(
            tokenizer_ipc_name=os.path.join(d, "tokenizer"),
            p_scheduler_input_ipc_name=os.path.join(d, "p_in"),
            d_scheduler_input_ipc_name=os.path.join(d, "d_in"),
            bridge_ipc_name=os.path.join(d, "bridge"),
            p_nccl_port=base + 1, d_nccl_port=base + 2)
