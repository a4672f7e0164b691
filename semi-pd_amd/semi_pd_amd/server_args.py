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


CU_MASK_MODES = ("env", "none", "dynamic")
# The operating point the bench line reports (profiles/r05_step_pacer_sweep_v2.txt, r05_step_pacer_slo_sweep.txt): prefill on 224
# of 256 CUs, a decode step older than 8.5 ms holds the prefill instance's launches, the deadline then follows a 12 ms
# objective for the 99th percentile of the time between tokens
DEFAULT_DECODE_STEP_DEADLINE_MS = 8.5
DEFAULT_DECODE_TBT_SLO_MS = 12.0


@dataclasses.dataclass
class ServerArgs:
    model_config: Any = None                 # LlamaConfig / OPTConfig / DeepseekV2Config; read from
    #                                          <model_path>/config.json when not given
    model_path: Optional[str] = None
    tokenizer_path: Optional[str] = None
    skip_tokenizer_init: bool = False
    served_model_name: Optional[str] = None
    host: str = "127.0.0.1"
    port: int = 30000
    load_format: str = "dummy"               # "dummy" (seeded random weights) | "auto" (HF safetensors)
    dummy_lm_head_scale: float = 1.0         # dummy weights: lm_head drawn this much wider (sharper logits for parity tests)
    dtype: str = "bfloat16"
    kv_cache_dtype: str = "auto"             # "auto" | "fp8_e5m2" | "fp8_e4m3" (MHA / GQA pools)
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
    disable_overlap_schedule: bool = False   # server_args.py --disable-overlap-schedule (decode instance of Semi-PD)
    enable_ep_moe: bool = False              # server_args.py --enable-ep-moe: routed experts partitioned by expert over TP
    # with --enable-ep-moe: the routed tokens travel to the ranks that own their experts and back (peer-memory all-to-all,
    # csrc/all_reduce.hip) instead of every rank computing on every token and all-reducing; no reference counterpart
    enable_ep_all_to_all: bool = False
    disable_custom_all_reduce: bool = False  # server_args.py --disable-custom-all-reduce: TP all-reduce through RCCL only
    cuda_graph_max_bs: int = 256
    # server_args.py:168, 321-323 --triton-attention-num-kv-splits (8, 16 on HIP): given, the decode attention uses that
    # split count like the reference; None (default) = chosen per batch (layers/attention_backend.py: choose_kv_splits)
    triton_attention_num_kv_splits: Optional[int] = None
    attention_backend: str = "hip"
    sampling_backend: str = "hip"
    watchdog_timeout: float = 300.0
    random_seed: int = 0
    eos_token_ids: Optional[List[int]] = None
    # Semi-PD compute split (semi_pd/utils.py:10-11 env knobs; BASELINE config 2 asks 50/50)
    prefill_cu_percent: int = PREFILL_ENGINE_SM_PERCENTILE
    decode_cu_percent: int = DECODE_ENGINE_SM_PERCENTILE
    # "env": process-wide HSA_CU_MASK per instance (static shares) | "none" | "dynamic": unmasked processes, each with a
    # CU-masked stream over its share and a stream over every CU, chosen per decode step / prefill batch from what the
    # other instance has in flight (semi_pd/share_board.py, model_executor/cu_share.py).  The default is the policy the
    # bench line reports (bench.py: P80 / D100 work-conserving): a server launched from the CLI runs what was measured
    cu_mask_mode: str = "dynamic"
    # dynamic mode: from this many waiting prompt tokens on a prefill batch takes every CU even while the decode
    # instance is busy (an overloaded GPU: throughput first).  0 = never
    prefill_backlog_full_tokens: int = 8192
    # Semi-PD, dynamic mode: a decode step older than this many milliseconds makes the prefill instance stop launching at its
    # next layer boundary until the step is over (semi_pd/step_pacer.py: host-side pacing over the share board); 0 = off
    # None = the measured default in dynamic mode (DEFAULT_DECODE_STEP_DEADLINE_MS), off otherwise
    decode_step_deadline_ms: Optional[float] = None
    # with a deadline: let it follow this objective for the 99th percentile of the time between tokens (ms; 0 = fixed
    # deadline): the latest deadline whose tail still meets it.  None = DEFAULT_DECODE_TBT_SLO_MS with the default deadline
    decode_tbt_slo_ms: Optional[float] = None
    test_plugin: Optional[str] = None        # tests only: a file every scheduler process executes at start-up (fault injection)
    prefill_stream_priority: int = 0         # HIP stream priority of the instance's compute stream: 0 normal, -1 high
    decode_stream_priority: int = 0
    # prefill-sized dense layers: time the library's GEMM solutions on the instance's own CU share at start-up and use
    # the winners (csrc/dense_gemm.hip).  None = when the instance that prefills runs under a CU mask
    tune_prefill_gemm: Optional[bool] = None
    # decode-sized GEMMs: size the K split for the instance's own CU share instead of the device (one round of workgroups on
    # a small static share; the last bit of a sum may then differ between the instances: ModelRunner.set_owned_cus)
    k_split_by_share: bool = False
    disable_stream_linear: bool = False      # dense layers of decode batches through hipBLASLt instead of csrc/stream_linear.hip
    library_gemm_grid: bool = False          # also size hipBLASLt's stream-K grids to the share (TENSILE_STREAMK_MAX_CUS)
    dist_init_addr: str = "127.0.0.1"
    nccl_port_base: Optional[int] = None
    dist_backend: str = "nccl"               # "nccl" = RCCL over xGMI; "gloo" only for single-GPU TP tests
    collect_kernel_timing: bool = False

    def __post_init__(self):
        if self.cu_mask_mode not in CU_MASK_MODES:
            raise ValueError(f"cu_mask_mode must be one of {CU_MASK_MODES}, got {self.cu_mask_mode!r}")
        # (one GPU per instance only: under tensor parallelism every rank would hold on its own GPU's board while the other
        #  ranks have already launched the layer's peer-memory all-reduce, whose blocks then spin next to the decode instance
        #  the hold is meant to relieve; a rank-0 decision broadcast per layer -- as CuShare.decide does per forward -- is
        #  not built, so tp_size > 1 runs the static frontier point)
        paced = self.enable_semi_pd and self.cu_mask_mode == "dynamic" and self.tp_size == 1
        if self.decode_step_deadline_ms is None:
            self.decode_step_deadline_ms = DEFAULT_DECODE_STEP_DEADLINE_MS if paced else 0.0
            if self.decode_tbt_slo_ms is None:
                self.decode_tbt_slo_ms = DEFAULT_DECODE_TBT_SLO_MS if paced else 0.0
        if self.decode_tbt_slo_ms is None:
            self.decode_tbt_slo_ms = 0.0
        if self.decode_step_deadline_ms < 0:
            raise ValueError("decode_step_deadline_ms must be >= 0 (0 = no deadline)")
        if self.decode_tbt_slo_ms < 0 or (self.decode_tbt_slo_ms > 0 and self.decode_step_deadline_ms <= 0):
            raise ValueError("decode_tbt_slo_ms adapts decode_step_deadline_ms: give a positive starting deadline with it")
        if self.decode_step_deadline_ms > 0 and self.enable_semi_pd and self.tp_size > 1:
            raise ValueError("decode_step_deadline_ms is for tp_size 1: the ranks of a tensor-parallel prefill instance have no "
                             "agreement on a hold (each would wait on its own GPU's board inside a layer whose all-reduce the "
                             "others have launched)")
        if self.decode_step_deadline_ms > 0 and self.enable_semi_pd and self.cu_mask_mode != "dynamic":
            raise ValueError("decode_step_deadline_ms needs --cu-mask-mode dynamic (the instances meet on the share board)")
        if self.enable_semi_pd and self.cu_mask_mode == "dynamic" and self.tp_size > 1 and self.disable_custom_all_reduce:
            raise ValueError("--disable-custom-all-reduce with --cu-mask-mode dynamic and tp_size > 1: the backend's collectives "
                             "run on their own unmasked stream and would leave the instance's CU share; use --cu-mask-mode "
                             "env or none with it")
        if self.prefill_backlog_full_tokens < 0:
            raise ValueError("prefill_backlog_full_tokens must be >= 0 (0 = never take every CU because of the backlog)")
        for name in ("prefill_cu_percent", "decode_cu_percent"):
            if not 1 <= int(getattr(self, name)) <= 100:
                raise ValueError(f"{name} must be in 1 .. 100, got {getattr(self, name)}")
        if self.cu_mask_mode == "dynamic" and self.enable_semi_pd and (self.prefill_stream_priority or self.decode_stream_priority):
            # model_executor/cu_share.py installs the NULL stream / the CU-masked stream as the instance's current stream:
            # a prioritised stream set up before it would be silently dropped
            raise ValueError("stream priorities need --cu-mask-mode env or none: in dynamic mode every instance runs on its "
                             "CU-masked stream or the NULL stream (model_executor/cu_share.py)")
        if self.model_config is None and self.model_path:
            from semi_pd_amd.model_loader import load_hf_config
            self.model_config = load_hf_config(self.model_path)
        if self.tokenizer_path is None:
            self.tokenizer_path = self.model_path
        if self.served_model_name is None:
            self.served_model_name = self.model_path or type(self.model_config).__name__
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


# ------------------------------------------------------------------------------------- CLI
def add_cli_args(parser):
    """The subset of server_args.py:380-1195 add_cli_args the Semi-PD path reads, same flag names.
    Flags the reference accepts but that have no effect here are still parsed, so that its launch
    commands (evaluation/benchmark_*_semi_pd.sh:14-16) run unchanged."""
    p = parser
    p.add_argument("--model-path", "--model", type=str, required=True)
    p.add_argument("--tokenizer-path", type=str, default=None)
    p.add_argument("--host", type=str, default="127.0.0.1")
    p.add_argument("--port", type=int, default=30000)
    p.add_argument("--skip-tokenizer-init", action="store_true")
    p.add_argument("--load-format", type=str, default="auto", choices=["auto", "safetensors", "dummy"])
    p.add_argument("--dtype", type=str, default="auto", choices=["auto", "half", "float16", "bfloat16"])
    p.add_argument("--kv-cache-dtype", type=str, default="auto", choices=["auto", "fp8_e5m2", "fp8_e4m3"])
    p.add_argument("--context-length", type=int, default=None)
    p.add_argument("--served-model-name", type=str, default=None)
    p.add_argument("--mem-fraction-static", type=float, default=None)
    p.add_argument("--max-running-requests", type=int, default=256)
    p.add_argument("--max-total-tokens", type=int, default=None)
    p.add_argument("--chunked-prefill-size", type=int, default=8192)
    p.add_argument("--max-prefill-tokens", type=int, default=16384)
    p.add_argument("--schedule-conservativeness", type=float, default=1.0)
    p.add_argument("--tensor-parallel-size", "--tp-size", "--tp", dest="tp_size", type=int, default=1)
    p.add_argument("--base-gpu-id", type=int, default=0)
    p.add_argument("--random-seed", type=int, default=0)
    p.add_argument("--watchdog-timeout", type=float, default=300.0)
    p.add_argument("--dist-init-addr", "--nccl-init-addr", dest="dist_init_addr", type=str, default="127.0.0.1")
    p.add_argument("--nccl-port", type=int, default=None)
    p.add_argument("--disable-cuda-graph", action="store_true")
    p.add_argument("--enable-ep-moe", action="store_true",
                   help="expert parallelism for the routed experts: E / tp whole experts per rank (ep_moe/layer.py)")
    p.add_argument("--enable-ep-all-to-all", action="store_true",
                   help="with --enable-ep-moe: expert all-to-all over the peer-memory regions (xGMI) instead of the all-reduce "
                        "of partial expert outputs")
    p.add_argument("--disable-custom-all-reduce", action="store_true",
                   help="TP all-reduce through RCCL only (default: peer-memory kernel up to 16 MB)")
    p.add_argument("--cuda-graph-max-bs", type=int, default=256)
    p.add_argument("--enable-semi-pd", action="store_true", help="prefill and decode instance on the same GPUs")
    p.add_argument("--prefill-cu-percent", type=int, default=PREFILL_ENGINE_SM_PERCENTILE,
                   help="share of the CUs given to the prefill instance (SEMI_PD_PREFILL_SM_PERCENTILE).  Default 88 (224 of 256 "
                        "CUs; the reference's MPS default is 80): shares come in whole groups of 32 CUs, 80 would be 192")
    p.add_argument("--decode-cu-percent", type=int, default=DECODE_ENGINE_SM_PERCENTILE,
                   help="share of the CUs given to the decode instance (SEMI_PD_DECODE_SM_PERCENTILE)")
    p.add_argument("--cu-mask-mode", type=str, default="dynamic", choices=list(CU_MASK_MODES),
                   help="how the shares are enforced: dynamic = unmasked processes, a CU-masked stream per instance, every CU "
                        "while the other instance is idle (the measured default; the reference's shares are static: env); "
                        "env = static HSA_CU_MASK per process; none = no mask.  dynamic refuses --*-stream-priority")
    p.add_argument("--prefill-backlog-full-tokens", type=int, default=8192,
                   help="dynamic mode: waiting prompt tokens from which a prefill batch takes every CU (0 = never)")
    p.add_argument("--prefill-stream-priority", type=int, default=0, choices=[-1, 0, 1],
                   help="HIP stream priority of the prefill instance (env / none modes; -1 = high)")
    p.add_argument("--decode-stream-priority", type=int, default=0, choices=[-1, 0, 1],
                   help="HIP stream priority of the decode instance (env / none modes; -1 = high)")
    p.add_argument("--decode-step-deadline-ms", type=float, default=None,
                   help="a decode step older than this makes the prefill instance yield at its next layer boundary until the "
                        f"step is over (0 = off).  Default: {DEFAULT_DECODE_STEP_DEADLINE_MS} with --cu-mask-mode dynamic and --tp-size 1 "
                        "(no reference counterpart), off otherwise; refused with --tp-size > 1")
    p.add_argument("--decode-tbt-slo-ms", type=float, default=None,
                   help="with --decode-step-deadline-ms: adapt the deadline so that the 99th percentile of the time between tokens "
                        f"meets this objective (0 = keep the deadline fixed).  Default {DEFAULT_DECODE_TBT_SLO_MS} with the default "
                        "deadline; the deadline moves at most 0.5 ms down / 2 ms up from where it started")
    p.add_argument("--k-split-by-share", action="store_true",
                   help="decode-sized GEMMs: K split sized for the instance's CU share instead of the device (faster on small "
                        "static shares; gives up bit-equal sums between the instances)")
    p.add_argument("--quantization", type=str, default=None, choices=[None, "fp8"],
                   help="fp8 = block-scaled e4m3fn checkpoint (quantization_config with weight_block_size); with "
                        "--load-format dummy it makes the seeded weights block-quantised")
    p.add_argument("--triton-attention-num-kv-splits", type=int, default=None,
                   help="fixed split-KV count of the decode attention (the reference's flag); default: chosen per batch")
    p.add_argument("--attention-backend", type=str, default="hip")
    p.add_argument("--sampling-backend", type=str, default="hip")
    p.add_argument("--log-level", type=str, default="info")
    # accepted for command-line compatibility; no effect on this path
    p.add_argument("--disable-overlap-schedule", action="store_true",
                   help="decode instance: look at the tokens of a step before launching the next one")
    for flag in ("--trust-remote-code", "--disable-radix-cache", "--enable-metrics",
                 "--enable-mixed-chunk", "--show-time-cost"):
        p.add_argument(flag, action="store_true")
    p.add_argument("--dist-timeout", type=int, default=None)
    p.add_argument("--stream-interval", type=int, default=1)
    return p


def from_cli_args(args) -> ServerArgs:
    dtype = "bfloat16" if args.dtype == "auto" else args.dtype
    sa = ServerArgs(
        model_path=args.model_path, tokenizer_path=args.tokenizer_path, host=args.host, port=args.port,
        skip_tokenizer_init=args.skip_tokenizer_init,
        load_format="dummy" if args.load_format == "dummy" else "auto", dtype=dtype,
        kv_cache_dtype=args.kv_cache_dtype,
        served_model_name=args.served_model_name, mem_fraction_static=args.mem_fraction_static,
        max_running_requests=args.max_running_requests, max_total_tokens=args.max_total_tokens,
        chunked_prefill_size=args.chunked_prefill_size, max_prefill_tokens=args.max_prefill_tokens,
        schedule_conservativeness=args.schedule_conservativeness, tp_size=args.tp_size,
        base_gpu_id=args.base_gpu_id, random_seed=args.random_seed, watchdog_timeout=args.watchdog_timeout,
        dist_init_addr=args.dist_init_addr, nccl_port_base=args.nccl_port,
        disable_cuda_graph=args.disable_cuda_graph, disable_custom_all_reduce=args.disable_custom_all_reduce,
        enable_ep_moe=args.enable_ep_moe, enable_ep_all_to_all=args.enable_ep_all_to_all, disable_overlap_schedule=args.disable_overlap_schedule, cuda_graph_max_bs=args.cuda_graph_max_bs,
        enable_semi_pd=args.enable_semi_pd, prefill_cu_percent=args.prefill_cu_percent,
        decode_cu_percent=args.decode_cu_percent, cu_mask_mode=args.cu_mask_mode,
        prefill_backlog_full_tokens=args.prefill_backlog_full_tokens,
        prefill_stream_priority=args.prefill_stream_priority, decode_stream_priority=args.decode_stream_priority,
        k_split_by_share=args.k_split_by_share, decode_step_deadline_ms=args.decode_step_deadline_ms,
        decode_tbt_slo_ms=args.decode_tbt_slo_ms,
        attention_backend=args.attention_backend,
        sampling_backend=args.sampling_backend, triton_attention_num_kv_splits=args.triton_attention_num_kv_splits)
    if args.quantization == "fp8":
        # server_args.py --quantization: the checkpoint decides (config.json: quantization_config); the flag
        # must agree with it.  There is no on-line weight quantisation, except for dummy weights.
        import dataclasses
        cfg = sa.model_config
        if not hasattr(cfg, "quantization_config"):
            raise ValueError(f"--quantization fp8 is supported for the DeepSeek family, not {type(cfg).__name__}")
        if cfg.quantization_config is None:
            if sa.load_format != "dummy":
                raise ValueError("--quantization fp8 needs a block-quantised checkpoint (quantization_config with "
                                 "weight_block_size in config.json); bf16 checkpoints are not quantised on load")
            sa.model_config = dataclasses.replace(cfg, quantization_config={
                "quant_method": "fp8", "weight_block_size": [128, 128], "activation_scheme": "dynamic"})
    ctx = args.context_length or getattr(sa.model_config, "max_position_embeddings", 4096)
    sa.context_length = int(ctx)
    eos = getattr(sa, "eos_token_ids", None)
    if eos is None:
        try:
            import json
            with open(os.path.join(args.model_path, "config.json")) as f:
                e = json.load(f).get("eos_token_id")
            if e is not None:
                sa.eos_token_ids = [int(x) for x in (e if isinstance(e, list) else [e])]
        except OSError:
            pass
    return sa
