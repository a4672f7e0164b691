"""ModelRunner: owns the model, the KV pools and the attention backend of ONE process
(prefill or decode) and implements the weight / KV-cache sharing of Semi-PD.

Reference: model_executor/model_runner.py — initialize :180-230, get_ipc_info :346-479,
share_params_from_ipc :481-624, init_memory_pool :972-1092, init_attention_backend :1127-1172,
forward / sample :1225-1325; decode graphs: cuda_graph_runner.py:109-150, 455-531.

The decode instance (role DECODE) loads the weights, allocates the KV slab and req_to_token and
exports one (hipIpcMemHandle, byte offset) per tensor.  The prefill instance (role PREFILL) builds
the same module tree on the `meta` device and maps every tensor from those handles, so weights and
KV cache exist once in HBM.
"""
from __future__ import annotations

import hashlib
import logging
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from semi_pd_amd import ops
from semi_pd_amd.distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                                     init_distributed_environment)
from semi_pd_amd.layers.attention_backend import HipAttnBackend
from semi_pd_amd.layers.basic import Sampler
from semi_pd_amd.mem_cache.memory_pool import (MHATokenToKVPool, MLATokenToKVPool, ReqToTokenPool,
                                               TokenToKVPoolAllocator)
from semi_pd_amd.model_executor.forward_batch_info import ForwardBatch, ForwardMode
from semi_pd_amd.semi_pd.utils import (InstanceRole, IPCInfo, convert_ipc_handle_to_tensor, get_device_sm_count,
                                       get_ipc_handle)

logger = logging.getLogger(__name__)


def build_model(model_config, dtype):
    arch = model_config.architectures[0]
    if arch == "LlamaForCausalLM":
        from semi_pd_amd.models.llama import LlamaForCausalLM
        return LlamaForCausalLM(model_config, dtype)
    if arch == "OPTForCausalLM":
        from semi_pd_amd.models.opt import OPTForCausalLM
        return OPTForCausalLM(model_config, dtype)
    if arch in ("DeepseekV2ForCausalLM", "DeepseekV3ForCausalLM"):
        from semi_pd_amd.models.deepseek_v2 import DeepseekV2ForCausalLM
        return DeepseekV2ForCausalLM(model_config, dtype)
    raise ValueError(f"unsupported architecture {arch}")


def _seed_of(name: str, seed: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:7], "little")


@torch.no_grad()
def dummy_init_weights(model: nn.Module, device: torch.device, seed: int = 0, std: float = 0.02,
                       lm_head_scale: float = 1.0):
    """`--load-format dummy` (model_loader/loader.py: DummyModelLoader): deterministic random weights.
    Every parameter is drawn on the device from a generator seeded by its *name*, norm weights are 1,
    so two processes (or a unified and a Semi-PD engine) build bit-identical models.  lm_head_scale multiplies the
    vocabulary projection: i.i.d. logits over a 128 k vocabulary are nearly flat (top-2 gap mostly below the 5e-2 tie
    margin of the parity tests); a scaled head spreads them so that most steps of a test discriminate."""
    params = dict(model.named_parameters())
    base_std = std
    for name, p in params.items():
        std = base_std * (lm_head_scale if name == "lm_head.weight" else 1.0)
        if p.device.type == "meta":
            continue
        if name.endswith("_scale_inv"):
            continue  # written together with the fp8 weight it belongs to
        if p.dtype == torch.float8_e4m3fn:
            # block-quantised layer: the SAME full-size draw as the unquantised model, quantised per weight
            # block, so an fp8 model is its bf16 twin up to quantisation error
            from semi_pd_amd.layers.fp8 import block_quantize_weight
            g = torch.Generator(device=device)
            g.manual_seed(_seed_of(name, seed))
            full_shape = getattr(p, "tp_full_shape", tuple(p.shape))
            full = torch.randn(full_shape, generator=g, device=device, dtype=torch.float32).mul_(std)
            sp = params[name + "_scale_inv"]
            q, sc = block_quantize_weight(full, p.weight_block_size)
            p.copy_(p.tp_shard(q) if hasattr(p, "tp_shard") else q)
            sp.copy_(sp.tp_shard(sc) if hasattr(sp, "tp_shard") else sc)
            continue
        leaf = name.rsplit(".", 2)[-2] if "." in name else name
        if "norm" in leaf and name.endswith("weight"):
            p.fill_(1.0)
        elif "norm" in leaf and name.endswith("bias"):
            p.zero_()
        else:
            g = torch.Generator(device=device)
            g.manual_seed(_seed_of(name, seed))
            # tensor-parallel shards are slices of ONE full-size draw, so TP=n computes the same model
            # as TP=1 (weight_loader semantics of layers/linear.py)
            full_shape = getattr(p, "tp_full_shape", tuple(p.shape))
            full = torch.randn(full_shape, generator=g, device=device, dtype=torch.float32).mul_(std).to(p.dtype)
            p.copy_(p.tp_shard(full) if hasattr(p, "tp_shard") else full)


@torch.no_grad()
def load_full_state_dict(model: nn.Module, full_sd: Dict[str, torch.Tensor]):
    """Load an unsharded state_dict (product parameter names): each rank keeps its shard."""
    params = dict(model.named_parameters(remove_duplicate=False))
    bufs = dict(model.named_buffers(remove_duplicate=False))
    for name, full in full_sd.items():
        if name in params:
            p = params[name]
            src = full.to(p.device, p.dtype)
            p.copy_(p.tp_shard(src) if (hasattr(p, "tp_shard") and tuple(src.shape) != tuple(p.shape)) else src)
        elif name in bufs:
            bufs[name].copy_(full.to(bufs[name].device, bufs[name].dtype))
        else:
            raise KeyError(f"unexpected parameter {name}")


class ModelRunner:
    def __init__(self, model_config, *, gpu_id: int = 0, tp_rank: int = 0, tp_size: int = 1,
                 dtype: torch.dtype = torch.bfloat16, context_length: int = 4096,
                 max_running_requests: int = 256, mem_fraction_static: float = 0.8,
                 max_total_tokens: Optional[int] = None, nccl_init_method: Optional[str] = None,
                 dist_backend: str = "nccl", instance_role: InstanceRole = InstanceRole.OTHER,
                 bypass_load_weight: bool = False, seed: int = 0, cu_percent: int = 100,
                 disable_cuda_graph: bool = False, cuda_graph_max_bs: int = 256,
                 load_state_dict: Optional[Dict[str, torch.Tensor]] = None, model_path: Optional[str] = None,
                 load_format: str = "dummy", kv_cache_dtype: str = "auto", disable_custom_all_reduce: bool = False,
                 enable_ep_moe: bool = False, enable_ep_all_to_all: bool = False, disable_stream_linear: bool = False,
                 num_kv_splits: Optional[int] = None, dummy_lm_head_scale: float = 1.0, k_split_by_share: bool = False,
                 step_deadline_ms: float = 0.0, tbt_slo_ms: float = 0.0):
        self.model_config = model_config
        # decode-step deadline (semi_pd/step_pacer.py): the prefill instance paces its launches layer by layer and stops
        # launching while the decode instance's step in flight is older than this (init_step_pacer)
        self.step_deadline_ms = float(step_deadline_ms)
        self.tbt_slo_ms = float(tbt_slo_ms or 0.0)
        self.step_pacer = None
        # the K split of the decode-sized streaming GEMM (and with it the order of its fp32 partial sums) is sized for the
        # DEVICE's CU count in every instance unless this is set: see set_owned_cus
        self.k_split_by_share = bool(k_split_by_share) or os.environ.get("SEMIPD_KSPLIT_BY_SHARE") == "1"
        self.model_path = model_path
        self.num_kv_splits = num_kv_splits        # --triton-attention-num-kv-splits; None = per batch
        self.gpu_id, self.tp_rank, self.tp_size = gpu_id, tp_rank, tp_size
        self.dtype = dtype
        # --kv-cache-dtype (server_args.py kv_cache_dtype; model_runner.py:690-710): pool rows in the
        # activation type or in OCP fp8
        try:
            self.kv_cache_dtype = {"auto": dtype, "fp8_e5m2": torch.float8_e5m2,
                                   "fp8_e4m3": torch.float8_e4m3fn}[kv_cache_dtype]
        except KeyError:
            raise ValueError(f"unsupported kv_cache_dtype {kv_cache_dtype!r}") from None
        self.max_context_len = context_length
        self.max_running_requests = max_running_requests
        self.mem_fraction_static = mem_fraction_static
        self.instance_role = instance_role
        self.bypass_load_weight = bypass_load_weight
        self.disable_cuda_graph = disable_cuda_graph
        self.cuda_graph_max_bs = cuda_graph_max_bs
        self.device = torch.device("cuda", gpu_id)
        torch.cuda.set_device(self.device)
        if tp_size > 1:
            from semi_pd_amd.layers.moe import EP_REGION_CAPACITY
            init_distributed_environment(tp_size, tp_rank, nccl_init_method, dist_backend, self.device,
                                         use_custom_all_reduce=not disable_custom_all_reduce,
                                         peer_region_capacity=EP_REGION_CAPACITY if (enable_ep_moe and enable_ep_all_to_all) else None)
        else:
            init_distributed_environment(1, 0, "", dist_backend)
        self.num_cus = get_device_sm_count(gpu_id)
        self.num_cus_owned = self.num_cus
        self.cu_share = None                     # --cu-mask-mode dynamic: model_executor/cu_share.py (init_cu_share)
        if cu_percent < 100:
            # what the launcher's HSA_CU_MASK really enables (whole groups of 8 logical CUs, one per XCD); the decode
            # instance's share is taken from the top of the range, which rounds differently on devices whose CU count is
            # not a multiple of 16
            from semi_pd_amd.semi_pd.utils import cu_mask_words
            from_top = instance_role == InstanceRole.DECODE
            self.num_cus_owned = sum(bin(w).count("1") for w in cu_mask_words(self.num_cus, cu_percent, from_top))
        # the decode-sized GEMMs fill whole rounds of the CUs this process owns (csrc/stream_linear.hip: sg_pick_ksplit)
        self.set_owned_cus(self.num_cus_owned)
        # --random-seed reaches the stochastic sampler through the default device generator, the same on every
        # TP rank and in both instances (model_runner.py: set_random_seed in every worker)
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        # dense layers with at most 64 rows (decode steps) stream their weights through LDS-DMA rings
        # (csrc/stream_linear.hip) instead of the library GEMM
        from semi_pd_amd.layers.basic import set_stream_linear
        set_stream_linear(not disable_stream_linear)

        # ---- model -------------------------------------------------------------------------
        from semi_pd_amd.layers.moe import set_expert_all_to_all, set_expert_parallel
        set_expert_parallel(enable_ep_moe)
        set_expert_all_to_all(enable_ep_moe and enable_ep_all_to_all)
        torch.set_default_dtype(dtype)
        try:
            if bypass_load_weight:
                # prefill instance: structure only, tensors arrive through IPC (model_runner.py:663-675)
                with torch.device("meta"):
                    self.model = build_model(model_config, dtype)
            else:
                with torch.device(self.device):
                    self.model = build_model(model_config, dtype)
                if load_state_dict is not None:
                    load_full_state_dict(self.model, load_state_dict)
                elif load_format != "dummy":
                    # --load-format auto: HF safetensors under --model-path (model_loader/loader.py)
                    from semi_pd_amd.model_loader import load_weights, safetensors_weights_iterator
                    if not model_path:
                        raise ValueError("load_format=%r needs a model_path" % load_format)
                    load_weights(self.model, model_config, safetensors_weights_iterator(model_path))
                else:
                    dummy_init_weights(self.model, self.device, seed, lm_head_scale=dummy_lm_head_scale)
                # the initialisation temporaries (fp32 draws of full-size tensors) go back to the driver before
                # anything else is allocated: a small tensor carved out of a cached multi-GiB block would be
                # exported with that block's size (see _make_ipc_safe)
                torch.cuda.empty_cache()
                if hasattr(self.model, "post_load_weights"):
                    # e.g. MLA's W_kc / W_vc buffers (deepseek_v2.py:1228-1249); the prefill instance
                    # receives them through IPC like any other buffer
                    self.model.post_load_weights()
                if instance_role == InstanceRole.DECODE:
                    self._make_ipc_safe()
        finally:
            torch.set_default_dtype(torch.float32)
        # give the initialisation temporaries (fp32 draws of full-size tensors) back to the driver: later
        # allocations that are exported through IPC (KV slab, req_to_token) must not be carved out of a
        # cached block whose size the importer cannot map (see csrc/ipc.hip)
        torch.cuda.empty_cache()
        self.model.eval()
        geo = self.model.kv_geometry
        self.kv_geometry = geo
        self.num_attention_heads_local = geo["num_heads"]
        self.num_kv_heads_local = geo["num_kv_heads"]
        self.v_head_dim = geo["v_head_dim"]
        self.sampler = Sampler()

        # ---- pools ---------------------------------------------------------------------------
        self.init_memory_pool(max_total_tokens)
        self.attn_backend = None
        self.graph_runner = None

    # ------------------------------------------------------------------------------------ CU share
    def set_owned_cus(self, cus: int) -> None:
        """The CUs of the stream the following launches go to: grids, K splits and split-KV counts are sized for them
        (csrc/stream_linear.hip: sg_pick_ksplit, csrc/gemm8p.hip, csrc/dense_gemm.hip's table, choose_kv_splits)."""
        from semi_pd_amd import _lib
        import os
        lib = _lib.load()
        # experiments: SEMIPD_DECLARED_CUS_DECODE / _PREFILL size an instance's grids for another CU count than it owns (an
        # unmasked decode instance next to a masked prefill instance finds only part of the chip free at any moment)
        over = os.environ.get("SEMIPD_DECLARED_CUS_" + self.instance_role.name)
        if over:
            cus = int(over)
        self.num_cus_owned = int(cus)
        # The weight-streaming GEMM of batches of at most 64 rows cuts K into slices by the CU count it is told, and the
        # slices set the order of its fp32 sums.  Every instance on a GPU -- prefill on its share or on every CU, decode,
        # the unified engine -- declares the DEVICE's count, so a decode-sized batch has the same bits wherever it runs (a
        # short prompt in the prefill instance, the same rows in a decode step).  --k-split-by-share sizes the split for
        # the share instead: one round of workgroups on a small static share, at the price of that guarantee.  (First
        # this call, then gemm_tall's: both record the owned count for the kernels that size grids by it, the last wins.)
        _lib.check(lib.semipd_stream_linear_set_cus(int(cus) if self.k_split_by_share else int(self.num_cus)),
                   "stream_linear_set_cus")
        _lib.check(lib.semipd_gemm_tall_set_cus(int(cus)), "gemm_tall_set_cus")
        _lib.check(lib.semipd_dense_gemm_set_cus(int(cus)), "dense_gemm_set_cus")
        ops.set_declared_cus(int(cus))

    def init_cu_share(self, role: InstanceRole, percent: int, board=None):
        """--cu-mask-mode dynamic: this (unmasked) process gets a CU-masked stream over its own share next to a stream over
        every CU and picks between them per unit of work (model_executor/cu_share.py)."""
        from semi_pd_amd.model_executor.cu_share import CuShare
        torch.cuda.synchronize(self.device)
        self.cu_share = CuShare(self, role, percent, board)
        if self.tp_size > 1:
            from semi_pd_amd.distributed import get_tp_cpu_group
            self.cu_share.tp = (self.tp_rank, get_tp_cpu_group())
        return self.cu_share

    # ------------------------------------------------------------------------------------ memory
    def profile_max_num_token(self) -> int:
        """model_runner.py:930-966: tokens that fit in mem_fraction_static of what is free now."""
        free, total = torch.cuda.mem_get_info(self.device)
        geo = self.kv_geometry
        if geo["kind"] == "mla":
            cell = (geo["kv_lora_rank"] + geo["qk_rope_head_dim"]) * geo["num_layers"] * self.kv_cache_dtype.itemsize
        else:
            cell = (geo["num_kv_heads"] * (geo["head_dim"] + geo["v_head_dim"]) * geo["num_layers"]
                    * self.kv_cache_dtype.itemsize)
        rest = free - total * (1 - self.mem_fraction_static)
        return max(int(rest // cell), 0)

    def init_memory_pool(self, max_total_tokens: Optional[int]):
        bypass = self.bypass_load_weight
        if max_total_tokens is None:
            if bypass:
                raise ValueError("the prefill instance needs the decode instance's max_total_tokens "
                                 "(model_runner.py:972-978)")
            max_total_tokens = self.profile_max_num_token()
            cap = self.max_running_requests * self.max_context_len
            max_total_tokens = min(max_total_tokens, cap)
        if max_total_tokens <= 0:
            raise RuntimeError("Not enough memory. Please try to increase --mem-fraction-static.")
        self.max_total_num_tokens = int(max_total_tokens)
        geo = self.kv_geometry
        dev = str(self.device)
        # +4 columns like the reference (model_runner.py:1040: max_context_len + 4)
        self.req_to_token_pool = ReqToTokenPool(self.max_running_requests + 1, self.max_context_len + 4, dev,
                                                bypass_create_buffers=bypass)
        if geo["kind"] == "mla":
            self.token_to_kv_pool = MLATokenToKVPool(self.max_total_num_tokens, 1, self.kv_cache_dtype,
                                                     geo["kv_lora_rank"], geo["qk_rope_head_dim"],
                                                     geo["num_layers"], dev, bypass_create_buffers=bypass)
        else:
            self.token_to_kv_pool = MHATokenToKVPool(self.max_total_num_tokens, 1, self.kv_cache_dtype,
                                                     geo["num_kv_heads"], geo["head_dim"], geo["num_layers"],
                                                     dev, bypass_create_buffers=bypass)
        self.token_to_kv_pool_allocator = TokenToKVPoolAllocator(self.max_total_num_tokens, self.kv_cache_dtype, dev,
                                                                 self.token_to_kv_pool)

    # ------------------------------------------------------------------------------------ IPC export
    @torch.no_grad()
    def _make_ipc_safe(self):
        """Re-home parameters and buffers whose backing allocation another process cannot import (ROCm 7.2:
        hipIpcOpenMemHandle hangs for allocation sizes with size mod 4 GiB >= 2 GiB, csrc/ipc.hip): a tensor of
        2 GiB or more owns its allocation, so it is copied into one that is padded to the next multiple of 4 GiB
        (DeepSeek-V3's fused expert weights are 7 GiB per layer and rank at TP = 1)."""
        from semi_pd_amd.mem_cache.memory_pool import ipc_safe_zeros
        moved = 0
        tensors = [(p, True) for _, p in self.model.named_parameters()] + \
                  [(b, False) for _, b in self.model.named_buffers()]
        for t, _ in tensors:
            nbytes = t.numel() * t.element_size()
            alloc = -(-nbytes // (2 << 20)) * (2 << 20)
            if nbytes < (1 << 31) or (alloc & 0xFFFFFFFF) < (1 << 31) or not t.is_contiguous():
                continue
            new = ipc_safe_zeros(tuple(t.shape), torch.uint8 if t.dtype.itemsize == 1 else t.dtype, t.device)
            new = new.view(t.dtype) if new.dtype != t.dtype else new
            new.copy_(t.data)
            t.data = new
            moved += 1
            # the old block must go back to the driver now: left in the allocator's cache, the next padded
            # allocation would be carved out of it and inherit its size
            torch.cuda.empty_cache()
        if moved:
            logger.info("moved %d tensors of 2 GiB+ into IPC-importable allocations", moved)

    def get_ipc_info(self) -> IPCInfo:
        """model_runner.py:346-479.  One (handle, offset) per parameter / buffer / KV layer /
        req_to_token; zero-size tensors are "BYPASS"."""
        params_info, weight_handles, buffer_handles = {}, {}, {}

        def describe(t: torch.Tensor):
            return {"shape": tuple(t.shape), "dtype": t.dtype, "stride": tuple(t.stride()),
                    "numel": t.numel(), "contiguous": t.is_contiguous()}

        for name, p in self.model.named_parameters(remove_duplicate=False):
            params_info[name] = describe(p)
            weight_handles[name] = "BYPASS" if p.numel() == 0 else get_ipc_handle(p.data)
        for name, b in self.model.named_buffers(remove_duplicate=False):
            params_info["buffer::" + name] = describe(b)
            buffer_handles[name] = "BYPASS" if b.numel() == 0 else get_ipc_handle(b)
        pool = self.token_to_kv_pool
        kv_handles: List[list] = []
        if isinstance(pool, MHATokenToKVPool):
            for i in range(pool.layer_num):
                kv_handles.append([get_ipc_handle(pool.k_buffer[i]), get_ipc_handle(pool.v_buffer[i])])
            kv_info = {"kind": "mha", "shape": tuple(pool.k_buffer[0].shape), "dtype": pool.dtype,
                       "numel": pool.k_buffer[0].numel(), "layer_num": pool.layer_num}
        else:
            for i in range(pool.layer_num):
                kv_handles.append([get_ipc_handle(pool.kv_buffer[i])])
            kv_info = {"kind": "mla", "shape": tuple(pool.kv_buffer[0].shape), "dtype": pool.dtype,
                       "numel": pool.kv_buffer[0].numel(), "layer_num": pool.layer_num}
        kv_info["max_total_num_tokens"] = self.max_total_num_tokens
        r2t = self.req_to_token_pool.req_to_token
        return IPCInfo(params_info=params_info, weight_handles=weight_handles,
                       register_buffer_handles=buffer_handles, kv_cache_handles=kv_handles,
                       kvcache_info=kv_info, req_to_token_handle=list(get_ipc_handle(r2t)),
                       req_to_token_info={"shape": tuple(r2t.shape), "dtype": r2t.dtype, "numel": r2t.numel()})

    # ------------------------------------------------------------------------------------ IPC import
    def _import(self, handle, info) -> torch.Tensor:
        flat = convert_ipc_handle_to_tensor(tuple(handle), info["numel"], info["dtype"], self.device)
        if info.get("contiguous", True):
            return flat.view(info["shape"])
        return torch.as_strided(flat, info["shape"], info["stride"])

    @staticmethod
    def _set_by_path(root: nn.Module, name: str, value, is_buffer: bool):
        mod = root
        parts = name.split(".")
        for p in parts[:-1]:
            mod = getattr(mod, p)
        if is_buffer:
            mod._buffers[parts[-1]] = value
        else:
            mod._parameters[parts[-1]] = nn.Parameter(value, requires_grad=False)

    def share_params_from_ipc(self, ipc_info: IPCInfo):
        """model_runner.py:481-624: re-materialise every tensor from the decode instance's handles."""
        cache: Dict[tuple, torch.Tensor] = {}

        def imp(handle, info):
            if handle == "BYPASS":
                return torch.empty(info["shape"], dtype=info["dtype"], device=self.device)
            key = (tuple(handle[0]), handle[1], info["dtype"], info["shape"])
            if key not in cache:  # tied weights: one tensor, several names
                cache[key] = self._import(handle, info)
            return cache[key]

        for name, handle in ipc_info.weight_handles.items():
            self._set_by_path(self.model, name, imp(handle, ipc_info.params_info[name]), False)
        for name, handle in ipc_info.register_buffer_handles.items():
            self._set_by_path(self.model, name, imp(handle, ipc_info.params_info["buffer::" + name]), True)
        left = [n for n, p in self.model.named_parameters() if p.device.type == "meta"]
        left += [n for n, b in self.model.named_buffers() if b.device.type == "meta"]
        if left:
            raise RuntimeError(f"tensors not covered by the IPC info: {left[:5]}")
        kvi = ipc_info.kvcache_info
        info = {"numel": kvi["numel"], "dtype": kvi["dtype"], "shape": kvi["shape"], "contiguous": True}
        pool = self.token_to_kv_pool
        if kvi["kind"] == "mha":
            pool.k_buffer = [self._import(h[0], info) for h in ipc_info.kv_cache_handles]
            pool.v_buffer = [self._import(h[1], info) for h in ipc_info.kv_cache_handles]
        else:
            pool.kv_buffer = [self._import(h[0], info) for h in ipc_info.kv_cache_handles]
        ri = ipc_info.req_to_token_info
        self.req_to_token_pool.req_to_token = self._import(
            ipc_info.req_to_token_handle, {"numel": ri["numel"], "dtype": ri["dtype"], "shape": ri["shape"]})

    # ------------------------------------------------------------------------------------ library GEMM selection
    def tune_dense_gemms(self, rows=(1024, 2048, 128, 256, 512, 1536, 3072, 4096, 6144, 8192), num_full_search: int = 2) -> str:
        """Time hipBLASLt's solutions for every dense weight shape of the model ON THE COMPUTE UNITS THIS PROCESS OWNS and
        route prefill-sized batches of those layers to the measured winners (csrc/dense_gemm.hip; ops.dense_gemm).
        Under an HSA_CU_MASK the library's own pick -- persistent stream-K grids sized for the whole device -- runs as
        two rounds on any partial share.  Returns the tuning table as text."""
        from semi_pd_amd import ops
        from semi_pd_amd.layers.basic import ColumnParallelLinear, RowParallelLinear
        # (SEMIPD_DG_NUM_FULL_SEARCH: how many row counts get the exhaustive search, ~20 s each per weight shape; tests use 0)
        num_full_search = int(os.environ.get("SEMIPD_DG_NUM_FULL_SEARCH", num_full_search))
        shapes = []
        for m in self.model.modules():
            if isinstance(m, (ColumnParallelLinear, RowParallelLinear)) and getattr(m, "quant_config", None) is None:
                w = m.weight
                if w.dim() == 2 and w.dtype in (torch.bfloat16, torch.float16) and w.device.type == "cuda":
                    key = (int(w.shape[0]), int(w.shape[1]), w.dtype)
                    if key not in shapes:
                        shapes.append(key)
        # start-up cache: the table is a property of (architecture, CUs of the share, library build, shapes, row counts);
        # it is kept under $SEMIPD_CACHE_DIR (default ~/.cache/semipd) and next to the model when there is a model path
        import hashlib
        lib_ver = (torch.version.hip or "?", ops.dense_gemm_library_version())   # solution indices are per library build
        arch = torch.cuda.get_device_properties(self.device).gcnArchName.split(":")[0]
        key = hashlib.sha256(repr((arch, self.num_cus, self.num_cus_owned, lib_ver, sorted((n, k, str(dt)) for n, k, dt in shapes),
                                   tuple(rows), int(num_full_search), os.environ.get("SEMIPD_DG_FINAL_US", ""))).encode()).hexdigest()[:16]
        dirs = [os.environ.get("SEMIPD_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "semipd")]
        # (a model directory may be shared and is not ours to write into or to trust: only on request)
        if os.environ.get("SEMIPD_DG_CACHE_IN_MODEL_DIR") == "1" and getattr(self, "model_path", None) \
                and os.path.isdir(self.model_path):
            dirs.insert(0, self.model_path)
        name = f"dense_gemm_{arch}_{self.num_cus_owned}cus_{key}.txt"
        if os.environ.get("SEMIPD_DG_CACHE", "1") != "0":
            for d in dirs:
                try:
                    text = open(os.path.join(d, name)).read()
                except OSError:
                    continue
                if ops.dense_gemm_import(text, shapes) > 0:
                    logger.warning("library GEMM table for %d CUs taken from %s", self.num_cus_owned, os.path.join(d, name))
                    self.time_tall_against_library()
                    return ops.dense_gemm_report()
        with torch.cuda.device(self.device):
            for n, k, dt in shapes:
                ops.dense_gemm_tune(n, k, list(rows), dt, num_full_search=num_full_search)
        self.time_tall_against_library()
        report = ops.dense_gemm_report()
        if os.environ.get("SEMIPD_DG_CACHE", "1") != "0":
            mine = "".join(ln + "\n" for ln in report.splitlines() if ln.startswith(f"cus={self.num_cus_owned} "))
            for d in dirs:
                try:
                    os.makedirs(d, exist_ok=True)
                    tmp = os.path.join(d, name + f".{os.getpid()}.tmp")
                    with open(tmp, "w") as f:
                        f.write(mine)
                    os.replace(tmp, os.path.join(d, name))
                    break
                except OSError:
                    continue
        return ops.dense_gemm_report()

    def time_tall_against_library(self, rows=(512, 1024, 1536, 2048, 3072, 4096, 8192)) -> str:
        """For every dense weight of the model: the tiled ping-pong GEMM (csrc/gemm8p.hip; with its SiLU * mul epilogue for a
        merged gate_up weight) against the library's measured winner (+ silu_and_mul) at prefill row counts, on this process's
        CUs and next to whatever runs beside it right now; where it wins by 3 % the layer takes it (ops.tall_preferred).
        A fraction of a second per start: not cached.  SEMIPD_TALL_PREFILL=0 turns it off."""
        import os
        from semi_pd_amd import ops
        from semi_pd_amd.layers.basic import ColumnParallelLinear, MergedColumnParallelLinear, RowParallelLinear
        if os.environ.get("SEMIPD_TALL_PREFILL", "1") == "0":
            return ""
        seen, lines = set(), []
        # the tiled kernel is taken below margin x the library's time.  SEMIPD_TALL_MARGIN_WIDE: the same for weights of 16 k
        # rows and more (gate_up: the library's pick there is a persistent stream-K kernel)
        margin = float(os.environ.get("SEMIPD_TALL_MARGIN", "0.97"))
        margin_wide = float(os.environ.get("SEMIPD_TALL_MARGIN_WIDE", str(margin)))

        def timed(fn, iters=3):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        with torch.cuda.device(self.device):
            for m in self.model.modules():
                if not isinstance(m, (ColumnParallelLinear, RowParallelLinear)) or getattr(m, "quant_config", None) is not None:
                    continue
                w = m.weight
                if w.dim() != 2 or w.dtype not in (torch.bfloat16, torch.float16) or w.device.type != "cuda" or m.bias is not None:
                    continue
                for silu in ((False, True) if isinstance(m, MergedColumnParallelLinear) and w.shape[0] % 2 == 0 else (False,)):
                    key = (int(w.shape[0]), int(w.shape[1]), w.dtype, silu)
                    if key in seen or not ops.dense_gemm_is_tuned(w):
                        continue
                    seen.add(key)
                    wins = []
                    for r in rows:
                        x = torch.randn(r, w.shape[1], device=self.device, dtype=torch.float32).to(w.dtype)
                        if not ops.gemm_tall_is_supported(x, w, fuse_silu_mul=silu):
                            continue
                        t_lib = timed((lambda: ops.silu_and_mul(ops.dense_gemm(x, w))) if silu else (lambda: ops.dense_gemm(x, w)))
                        t_tall = timed(lambda: ops.gemm_tall(x, w, fuse_silu_mul=silu))
                        wins.append((r, t_tall < (margin_wide if w.shape[0] >= 16384 else margin) * t_lib))
                        lines.append(f"n={w.shape[0]} k={w.shape[1]} silu={int(silu)} rows={r}: library {t_lib:.1f} us, tiled {t_tall:.1f} us"
                                     f"{'  <- tiled' if wins[-1][1] else ''}")
                        del x
                    ops.set_tall_preference(w.shape[0], w.shape[1], w.dtype, silu, wins)
        text = "\n".join(lines)
        if lines:
            logger.warning("tiled GEMM against the library's measured winners on %d CUs:\n%s", self.num_cus_owned, text)
        return text

    # ------------------------------------------------------------------------------------ backend / graphs
    def init_attention_backend(self):
        self.attn_backend = HipAttnBackend(self)

    def init_cuda_graphs(self):
        """Decode hipGraphs (cuda_graph_runner.py): only the decode instance captures
        (semi_pd_scheduler.py:409-411)."""
        if self.disable_cuda_graph:
            return
        from semi_pd_amd.model_executor.hip_graph_runner import HipGraphRunner
        if self.cu_share is not None:
            # the process's own capture stream is unmasked; replays go to whichever stream is current (cu_share.py)
            torch.cuda.synchronize(self.device)
        if self.tp_size == 1:
            self.graph_runner = HipGraphRunner(self)
            return
        # TP > 1: the graphs contain RCCL all-reduces.  If this RCCL build refuses stream capture the
        # decode instance keeps running the same HIP kernels eagerly instead of dying at start-up.
        try:
            self.graph_runner = HipGraphRunner(self)
        except Exception as e:  # pragma: no cover (needs a multi-GPU node)
            logger.warning("decode hipGraph capture failed with TP=%d (%s); running decode eagerly", self.tp_size, e)
            self.graph_runner = None
            # the runner has destroyed its capture stream and parked what lived on it (hip_graph_runner.py)
            from semi_pd_amd.model_executor.hip_graph_runner import recover_after_failed_capture
            recover_after_failed_capture(self.device)
        # graph or eager is ONE decision for the whole tensor-parallel group: a rank that replays while another launches
        # eagerly runs collectives with different payloads (padded vs raw batch) against each other and both wait for ever
        from semi_pd_amd.distributed import all_ranks_agree
        if not all_ranks_agree(self.graph_runner is not None) and self.graph_runner is not None:
            logger.warning("decode hipGraphs dropped on rank %d: another tensor-parallel rank failed to capture", self.tp_rank)
            self._unused_graph_runner = self.graph_runner      # (kept alive: its graphs own pool memory other tensors share)
            self.graph_runner = None

    # ------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, forward_batch: ForwardBatch):
        kt = getattr(self, "kernel_timing", None)
        sampled = kt.begin_step() if kt is not None else False
        try:
            if forward_batch.forward_mode.is_decode():
                # a sampled step runs eagerly so that HIP events can bracket the attention launches
                if (not sampled and self.graph_runner is not None
                        and self.graph_runner.can_run(forward_batch)):
                    return self.graph_runner.replay(forward_batch)
                return self.forward_decode(forward_batch)
            if forward_batch.forward_mode.is_extend():
                return self.forward_extend(forward_batch)
            raise ValueError(f"Invalid forward mode: {forward_batch.forward_mode}")
        finally:
            if kt is not None:
                kt.end_step()

    def forward_decode(self, forward_batch: ForwardBatch):
        self.attn_backend.init_forward_metadata(forward_batch)
        return self.model.forward(forward_batch.input_ids, forward_batch.positions, forward_batch)

    def init_step_pacer(self, board) -> None:
        """Prefill instance with a decode-step deadline: a forward pre-hook in front of every decoder layer that (1) keeps the
        host at most two layers ahead of the GPU and (2) stops launching while the decode instance's step in flight is older
        than the deadline (semi_pd/step_pacer.py)."""
        from semi_pd_amd.semi_pd.step_pacer import StepPacer
        self.step_pacer = StepPacer(board, self.step_deadline_ms, self.device, slo_ms=self.tbt_slo_ms)
        # bench.py's accounting of a prefill batch (roofline_extra.prefill_batch_ms): GPU time per layer from the hooks' events
        self.step_pacer.time_layers = getattr(self, "kernel_timing", None) is not None
        n = 0
        for name, m in self.model.named_modules():
            if isinstance(m, nn.ModuleList) and name.split(".")[-1] == "layers":
                for i, layer in enumerate(m):
                    layer.register_forward_pre_hook(lambda mod, args, _p=self.step_pacer, _i=i: _p.before_layer(_i))
                    n += 1
        if n == 0:
            raise RuntimeError("decode-step deadline: the model has no `layers` ModuleList to pace")

    def forward_extend(self, forward_batch: ForwardBatch):
        self.attn_backend.init_forward_metadata(forward_batch)
        return self.model.forward(forward_batch.input_ids, forward_batch.positions, forward_batch)

    def sample(self, logits_output, forward_batch=None) -> torch.Tensor:
        return self.sampler(logits_output, getattr(forward_batch, "sampling_info", None),
                            return_logprob=bool(getattr(forward_batch, "return_logprob", False)),
                            top_logprobs_nums=getattr(forward_batch, "top_logprobs_nums", None))
