"""Engine: process topology of Semi-PD mode and the token-level client API.

Reference: entrypoints/engine.py:540-728 `_launch_semi_pd_subprocesses` — per TP rank the decode
process is forked first (loads weights, allocates KV, exports IPCInfo through an mp.Queue), then the
prefill process (maps everything from the handles); managers/semi_pd_scheduler.py:326-432
`run_scheduler_process`.  The compute share of each process is set through the environment before
the fork (engine.py:591-593, 632-634) — here a CU mask (HSA_CU_MASK) instead of
CUDA_MPS_ACTIVE_THREAD_PERCENTAGE.

Tokenisation / HTTP are §8(f) "next" rows: this client API speaks token ids.
"""
from __future__ import annotations

import faulthandler
import logging
import multiprocessing as mp
import os
import signal
import sys
import time
import traceback
from typing import Dict, Iterable, List, Optional, Sequence

from semi_pd_amd.managers.io_struct import (BatchTokenIDOut, SamplingParams, ShutdownReq, StatsReq, SyntheticLoadReq,
                                            TokenizedGenerateReqInput)
from semi_pd_amd.distributed import get_custom_all_reduce
from semi_pd_amd.managers.transport import PullSocket, PushSocket
from semi_pd_amd.semi_pd import ttft_trace
from semi_pd_amd.semi_pd.utils import AggregatedSocket, InstanceRole
from semi_pd_amd.server_args import SemiPDPortArgs, ServerArgs

logger = logging.getLogger(__name__)


def _build_runner(server_args: ServerArgs, gpu_id: int, tp_rank: int, role: InstanceRole, nccl_port: int,
                  max_total_tokens=None, bypass_load_weight=False, cu_percent=100):
    from semi_pd_amd.model_executor.model_runner import ModelRunner
    mr = ModelRunner(
        server_args.model_config, gpu_id=gpu_id, tp_rank=tp_rank, tp_size=server_args.tp_size,
        dtype=server_args.torch_dtype, context_length=server_args.context_length,
        max_running_requests=server_args.max_running_requests,
        mem_fraction_static=server_args.mem_fraction_static, max_total_tokens=max_total_tokens,
        nccl_init_method=f"tcp://{server_args.dist_init_addr}:{nccl_port}", instance_role=role,
        dist_backend=server_args.dist_backend, model_path=server_args.model_path,
        load_format=server_args.load_format, kv_cache_dtype=server_args.kv_cache_dtype,
        bypass_load_weight=bypass_load_weight, seed=server_args.random_seed, cu_percent=cu_percent,
        disable_cuda_graph=server_args.disable_cuda_graph, cuda_graph_max_bs=server_args.cuda_graph_max_bs,
        disable_custom_all_reduce=server_args.disable_custom_all_reduce, enable_ep_moe=server_args.enable_ep_moe,
        enable_ep_all_to_all=server_args.enable_ep_all_to_all,
        disable_stream_linear=server_args.disable_stream_linear,
        num_kv_splits=server_args.triton_attention_num_kv_splits,
        dummy_lm_head_scale=server_args.dummy_lm_head_scale, k_split_by_share=server_args.k_split_by_share,
        step_deadline_ms=(server_args.decode_step_deadline_ms if server_args.enable_semi_pd else 0.0),
        tbt_slo_ms=server_args.decode_tbt_slo_ms)
    if server_args.collect_kernel_timing:
        from semi_pd_amd.model_executor.kernel_timing import KernelTiming
        mr.kernel_timing = KernelTiming()
        from semi_pd_amd.layers.basic import set_stream_linear_timing
        set_stream_linear_timing(mr.kernel_timing)
    return mr


def _init_dynamic_share(server_args: ServerArgs, port_args: SemiPDPortArgs, mr, role: InstanceRole):
    """--cu-mask-mode dynamic: the process is unmasked; it gets a masked stream over its share next to a stream over every
    CU (model_executor/cu_share.py) and meets the other instance on the share board (semi_pd/share_board.py), a file next
    to the engine's sockets."""
    if server_args.cu_mask_mode != "dynamic":
        return
    from semi_pd_amd.semi_pd.share_board import ShareBoard
    # one board per tensor-parallel rank: rank r's prefill and decode process share GPU r
    board = ShareBoard(os.path.join(os.path.dirname(port_args.tokenizer_ipc_name), f"share_board_{mr.tp_rank}"), create=True)
    percent = server_args.decode_cu_percent if role == InstanceRole.DECODE else server_args.prefill_cu_percent
    mr.init_cu_share(role, percent, board)
    if role == InstanceRole.PREFILL and server_args.decode_step_deadline_ms > 0:
        mr.init_step_pacer(board)


def run_scheduler_process(server_args: ServerArgs, port_args: SemiPDPortArgs, gpu_id: int, tp_rank: int,
                          role: InstanceRole, ipc_queue, pipe_writer, pkg_paths: Sequence[str]):
    """managers/semi_pd_scheduler.py:326-432."""
    for p in pkg_paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    faulthandler.enable()
    if os.environ.get("SEMIPD_DUMP_TRACEBACK_AFTER"):  # debugging aid: where does a start-up hang?
        faulthandler.dump_traceback_later(float(os.environ["SEMIPD_DUMP_TRACEBACK_AFTER"]), repeat=False)
    logging.basicConfig(level=os.environ.get("SEMIPD_LOGLEVEL", "WARNING"),
                        format=f"[%(asctime)s {role.name} TP{tp_rank}] %(message)s")
    parent = os.getppid()
    try:
        import torch
        if server_args.test_plugin:
            # tests reach into the scheduler processes through a plugin file executed at start-up (fault injection
            # lives in tests/, not in the serving code); an explicit ServerArgs field the tests set -- no command-line
            # flag, no environment variable: nothing in a production launch can name a file to run
            import runpy
            runpy.run_path(server_args.test_plugin, run_name="semipd_test_plugin")
        from semi_pd_amd.managers.semi_pd_decode_scheduler import SemiPDDecodeScheduler
        from semi_pd_amd.managers.semi_pd_prefill_scheduler import SemiPDPrefillScheduler
        rank0 = tp_rank == 0
        prio = server_args.decode_stream_priority if role == InstanceRole.DECODE else server_args.prefill_stream_priority
        if role == InstanceRole.DECODE and os.environ.get("SEMIPD_DECODE_OWN_STREAM") == "1" and not prio:
            # experiments: the decode instance on a created (non-blocking) stream instead of the NULL stream
            import ctypes
            from semi_pd_amd import _lib
            torch.cuda.set_device(gpu_id)
            raw = ctypes.c_void_p()
            _lib.check(_lib.load().semipd_stream_create(gpu_id, ctypes.addressof(raw)), "stream_create")
            torch.cuda.set_stream(torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", gpu_id)))
        if prio:
            # the whole instance (weights, graphs, every launch) lives on one prioritised stream: the hardware
            # scheduler serves the queue of a high-priority stream first when both instances have work ready
            # (created through the C-ABI: torch clamps positive -- LOW -- priorities to 0)
            import ctypes
            from semi_pd_amd import _lib
            torch.cuda.set_device(gpu_id)
            raw, rng = ctypes.c_void_p(), (ctypes.c_int * 2)()
            _lib.check(_lib.load().semipd_stream_create_with_priority(gpu_id, int(prio), ctypes.addressof(raw),
                                                                      ctypes.addressof(rng)), "stream_create_with_priority")
            logging.getLogger(__name__).warning("%s instance on a stream of priority %d (device range %d .. %d)", role.name,
                                                int(prio), rng[0], rng[1])
            torch.cuda.set_stream(torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", gpu_id)))
        if role == InstanceRole.DECODE:
            mr = _build_runner(server_args, gpu_id, tp_rank, role, port_args.d_nccl_port,
                               max_total_tokens=server_args.max_total_tokens,
                               cu_percent=server_args.decode_cu_percent)
            ipc_queue.put(mr.get_ipc_info())       # semi_pd_scheduler.py:388-389
            _init_dynamic_share(server_args, port_args, mr, role)
            mr.init_attention_backend()
            mr.init_cuda_graphs()                   # decode only (semi_pd_scheduler.py:409-411)
            sched = SemiPDDecodeScheduler(
                server_args, mr, tp_rank,
                recv_socket=PullSocket(port_args.d_scheduler_input_ipc_name) if rank0 else None,
                send_to_detokenizer=PushSocket(port_args.tokenizer_ipc_name) if rank0 else None,
                bridge_socket=PushSocket(port_args.bridge_ipc_name) if rank0 else None,
                send_to_p_instance=PushSocket(port_args.p_scheduler_input_ipc_name) if rank0 else None)
        else:
            ipc_info = ipc_queue.get()              # semi_pd_scheduler.py:369-370
            mr = _build_runner(server_args, gpu_id, tp_rank, role, port_args.p_nccl_port,
                               max_total_tokens=ipc_info.kvcache_info["max_total_num_tokens"],
                               bypass_load_weight=True, cu_percent=server_args.prefill_cu_percent)
            mr.share_params_from_ipc(ipc_info)      # semi_pd_scheduler.py:406-407
            _init_dynamic_share(server_args, port_args, mr, role)
            mr.init_attention_backend()
            to_d = PushSocket(port_args.d_scheduler_input_ipc_name) if rank0 else None
            tune = server_args.tune_prefill_gemm
            if tune or (tune is None and server_args.prefill_cu_percent < 100
                        and server_args.cu_mask_mode in ("env", "dynamic")):
                # the candidates are timed next to what they will run next to: the decode instance (ready and idle at
                # this point) replays a captured decode step in a loop meanwhile
                under_load = (rank0 and server_args.tp_size == 1
                              and os.environ.get("SEMIPD_TUNE_UNDER_DECODE_LOAD", "1") != "0")
                t0 = time.time()
                if under_load:
                    to_d.send_pyobj(SyntheticLoadReq(on=True))
                    time.sleep(0.1)
                    # watch every candidate for longer than the load's period (a decode step, ~6 ms)
                    os.environ.setdefault("SEMIPD_DG_FIRST_US", "1500")
                    os.environ.setdefault("SEMIPD_DG_FINAL_US", "20000")
                    os.environ.setdefault("SEMIPD_DG_FINALISTS", "10")
                try:
                    table = mr.tune_dense_gemms()
                finally:
                    if under_load:
                        to_d.send_pyobj(SyntheticLoadReq(on=False))
                logger.warning("library GEMM solutions timed on this share%s in %.1f s:\n%s",
                               " next to a running decode step" if under_load else "", time.time() - t0, table)
            sched = SemiPDPrefillScheduler(
                server_args, mr, tp_rank,
                recv_socket=PullSocket(port_args.p_scheduler_input_ipc_name) if rank0 else None,
                send_to_d_instance=to_d,
                bridge_socket=PullSocket(port_args.bridge_ipc_name) if rank0 else None,
                send_stats_to=PushSocket(port_args.tokenizer_ipc_name) if rank0 else None)
        torch.cuda.synchronize()
        pipe_writer.send({"status": "ready", "max_total_num_tokens": mr.max_total_num_tokens,
                          "max_req_input_len": sched.max_req_input_len, "role": role.name,
                          "tp_rank": tp_rank, "hsa_cu_mask": os.environ.get("HSA_CU_MASK", ""),
                          "custom_all_reduce": get_custom_all_reduce() is not None})
        sched.event_loop_normal()
        # orderly end: nothing in flight, the created streams gone before the runtime's exit handlers run
        torch.cuda.synchronize()
        if getattr(mr, "cu_share", None) is not None:
            mr.cu_share.close()
    except Exception:
        msg = traceback.format_exc()
        logger.error("scheduler hit an exception: %s", msg)
        try:
            pipe_writer.send({"status": "error", "error": msg, "role": role.name})
        except Exception:
            pass
        # the reference sends SIGQUIT to the parent (semi_pd_scheduler.py:429-432); here the launcher
        # notices the dead child (Engine.check_children) and fails fast on its own thread
        del parent
        raise


def _pkg_paths() -> List[str]:
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.dirname(os.path.dirname(here))  # .../semi-pd_amd
    return [pkg, os.path.dirname(pkg)]


class Engine:
    """Token-id client over either a Semi-PD process pair per TP rank or an in-process unified
    scheduler (enable_semi_pd=False), with the same request / streaming interface."""

    def __init__(self, server_args: ServerArgs, local_tp_ranks: Optional[Iterable[int]] = None,
                 gpu_ids: Optional[Dict[int, int]] = None, ready_timeout: float = 1800.0):
        self.server_args = server_args
        self.procs: List[mp.Process] = []
        self._rid = 0
        self._outputs: Dict[str, List[int]] = {}
        self._logprobs: Dict[str, dict] = {}
        self._finished: Dict[str, Optional[str]] = {}
        self._first_token_time: Dict[str, float] = {}
        self._token_times: Dict[str, List[float]] = {}
        self._send_time: Dict[str, float] = {}
        self._stats_inbox: List[dict] = []
        self.scheduler = None
        self.local_tp_ranks = list(local_tp_ranks) if local_tp_ranks is not None else list(range(server_args.tp_size))
        self.gpu_ids = gpu_ids or {r: server_args.base_gpu_id + r for r in self.local_tp_ranks}
        self.is_driver = 0 in self.local_tp_ranks  # the client sockets live next to TP rank 0
        if server_args.enable_semi_pd:
            self._launch_semi_pd_subprocesses(ready_timeout)
        else:
            self._launch_unified_in_process()

    # ------------------------------------------------------------------------------------ launch
    def _launch_unified_in_process(self):
        from semi_pd_amd.managers.scheduler import Scheduler
        sa = self.server_args
        assert sa.tp_size == 1 or len(self.local_tp_ranks) == 1
        tp_rank = self.local_tp_ranks[0]
        mr = _build_runner(sa, self.gpu_ids[tp_rank], tp_rank, InstanceRole.OTHER,
                           (sa.nccl_port_base or 29600) + 3, max_total_tokens=sa.max_total_tokens)
        mr.init_attention_backend()
        mr.init_cuda_graphs()
        self.model_runner = mr
        self._inbox: List = []
        engine = self

        class _Loop:  # in-process stand-ins for the two sockets
            def recv_pyobj_nowait(self_inner):
                from semi_pd_amd.managers.transport import NOTHING
                return engine._inbox.pop(0) if engine._inbox else NOTHING

            def send_pyobj(self_inner, obj):
                engine._handle_output(obj)

        loop = _Loop()
        self.scheduler = Scheduler(sa, mr, tp_rank, loop, loop)
        self.max_total_num_tokens = mr.max_total_num_tokens

    def _launch_semi_pd_subprocesses(self, ready_timeout: float):
        """entrypoints/engine.py:540-728."""
        sa = self.server_args
        from semi_pd_amd.semi_pd.utils import cu_mask_env, get_device_sm_count
        self.port_args = SemiPDPortArgs.init_new(sa)
        ctx = mp.get_context("spawn")
        if self.is_driver:
            self.recv_from_scheduler = PullSocket(self.port_args.tokenizer_ipc_name)
        queues = {r: ctx.Queue() for r in self.local_tp_ranks}
        readers = []
        paths = _pkg_paths()

        def spawn(role, tp_rank, percent, from_top):
            gpu_id = self.gpu_ids[tp_rank]
            reader, writer = ctx.Pipe(duplex=False)
            env_add = {}
            # (SEMIPD_DYN_ALSO_ENV_MASK=1, experiments: dynamic mode with the process mask on top of the masked stream)
            if (sa.cu_mask_mode == "env" or (sa.cu_mask_mode == "dynamic" and os.environ.get("SEMIPD_DYN_ALSO_ENV_MASK") == "1")) \
                    and percent < 100:
                env_add = cu_mask_env(gpu_id, self._num_cus(gpu_id), percent, from_top,
                                      library_grid=sa.library_gemm_grid)
            old = {k: os.environ.get(k) for k in env_add}
            os.environ.update(env_add)   # like engine.py:591-593: set the share, then fork
            try:
                p = ctx.Process(target=run_scheduler_process, name=f"semipd-{role.name.lower()}-tp{tp_rank}",
                                args=(sa, self.port_args, gpu_id, tp_rank, role, queues[tp_rank], writer, paths))
                p.start()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            self.procs.append(p)
            readers.append((role, tp_rank, reader, p))

        try:
            # decode instances first: they own the weights and the KV cache
            for r in self.local_tp_ranks:
                spawn(InstanceRole.DECODE, r, sa.decode_cu_percent, from_top=True)
            infos = self._wait_ready([x for x in readers if x[0] == InstanceRole.DECODE], ready_timeout)
            self.max_total_num_tokens = min(i["max_total_num_tokens"] for i in infos)
            n_d = len(readers)
            for r in self.local_tp_ranks:
                spawn(InstanceRole.PREFILL, r, sa.prefill_cu_percent, from_top=False)
            infos += self._wait_ready(readers[n_d:], ready_timeout)
        except BaseException:
            # a child that failed after queueing its IPC info cannot exit while nobody drains the queue, and the
            # interpreter waits for non-daemon children: stop them here so that the error surfaces at once
            for p in self.procs:
                if p.is_alive():
                    p.terminate()
            for p in self.procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
            raise
        self.ready_infos = infos
        if self.is_driver:
            # every request goes to D first, then to P (tokenizer_manager.py:149-160)
            self.send_to_scheduler = AggregatedSocket([
                PushSocket(self.port_args.d_scheduler_input_ipc_name),
                PushSocket(self.port_args.p_scheduler_input_ipc_name)])

    _cu_cache: Dict[int, int] = {}

    def _num_cus(self, gpu_id: int) -> int:
        if gpu_id not in Engine._cu_cache:
            env = os.environ.get("SEMIPD_NUM_CUS")
            if env:
                Engine._cu_cache[gpu_id] = int(env)
            else:
                # query in a throw-away process so the launcher itself never initialises HIP
                ctx = mp.get_context("spawn")
                q = ctx.Queue()
                p = ctx.Process(target=_query_cus, args=(gpu_id, q, _pkg_paths()))
                p.start()
                Engine._cu_cache[gpu_id] = q.get(timeout=300)
                p.join()
        return Engine._cu_cache[gpu_id]

    def _wait_ready(self, readers, timeout: float) -> List[dict]:
        infos = []
        deadline = time.monotonic() + timeout
        for role, tp_rank, reader, proc in readers:
            while True:
                if reader.poll(1.0):
                    data = reader.recv()
                    break
                if not proc.is_alive():
                    raise RuntimeError(f"{role.name} TP{tp_rank} process died during start-up "
                                       f"(exit code {proc.exitcode})")
                if time.monotonic() > deadline:
                    raise TimeoutError(f"{role.name} TP{tp_rank} did not become ready in {timeout}s")
            if data.get("status") != "ready":
                raise RuntimeError(f"{role.name} TP{tp_rank} failed to initialise:\n{data.get('error')}")
            infos.append(data)
        return infos

    # ------------------------------------------------------------------------------------ client API
    def add_request(self, input_ids: Sequence[int], sampling_params: SamplingParams,
                    rid: Optional[str] = None, return_logprob: bool = False, top_logprobs_num: int = 0) -> str:
        if rid is None:
            rid = f"r{self._rid}"
            self._rid += 1
        req = TokenizedGenerateReqInput(rid=rid, input_text=None, input_ids=list(input_ids),
                                        sampling_params=sampling_params, return_logprob=bool(return_logprob),
                                        top_logprobs_num=int(top_logprobs_num))
        self._outputs[rid] = []
        if return_logprob:
            self._logprobs[rid] = {"token": [], "top": []}
        self._finished[rid] = None
        self._token_times[rid] = []
        self._send_time[rid] = time.time()
        ttft_trace.mark("client_send", [rid])
        if self.scheduler is not None:
            self._inbox.append(req)
        else:
            self.send_to_scheduler.send_pyobj(req)
        return rid

    def abort_request(self, rid: str):
        """Stop generating for `rid` (client disconnect, stop string): both instances are told.  A queued
        request is dropped without a reply (scheduler.py:1565-1577), so the record is closed here; a running
        one still sends its last tokens with finish reason "abort" when its slots are freed."""
        from semi_pd_amd.managers.io_struct import AbortReq
        if rid in self._finished and self._finished[rid] is None:
            self._finished[rid] = "abort"
        if self.scheduler is not None:
            self._inbox.append(AbortReq(rid))
        else:
            self.send_to_scheduler.send_pyobj(AbortReq(rid))

    def _handle_output(self, obj):
        if isinstance(obj, BatchTokenIDOut):
            now = time.time()
            for i, (rid, fin, toks) in enumerate(zip(obj.rids, obj.finished_reasons, obj.output_ids)):
                if rid not in self._outputs:
                    continue
                self._outputs[rid].extend(toks)
                if obj.output_token_logprobs is not None and rid in self._logprobs \
                        and obj.output_token_logprobs[i] is not None:
                    self._logprobs[rid]["token"].extend(obj.output_token_logprobs[i])
                    self._logprobs[rid]["top"].extend(obj.output_top_logprobs[i])
                if not self._token_times[rid]:
                    ttft_trace.mark("client_first_token", [rid])
                self._token_times[rid].extend([now] * len(toks))
                if fin is not None:
                    self._finished[rid] = fin
        elif isinstance(obj, tuple) and obj and obj[0] == "stats":
            self._stats_inbox.append(obj[1])

    def poll(self, timeout: float = 0.0) -> bool:
        """Pump one message (or, in-process, one scheduler step).  Returns True if anything happened."""
        if self.scheduler is not None:
            return self.scheduler.step()
        from semi_pd_amd.managers.transport import NOTHING
        obj = self.recv_from_scheduler.recv_pyobj_nowait()
        if obj is NOTHING:
            if timeout > 0:
                try:
                    obj = self.recv_from_scheduler.recv_pyobj(timeout=timeout)
                except TimeoutError:
                    return False
            else:
                return False
        self._handle_output(obj)
        return True

    def check_children(self):
        for p in self.procs:
            if not p.is_alive():
                raise RuntimeError(f"scheduler process {p.pid} exited with code {p.exitcode}")

    def wait(self, rids: Sequence[str], timeout: float = 3600.0):
        deadline = time.monotonic() + timeout
        pending = set(rids)
        while pending:
            if not self.poll(timeout=0.05):
                self.check_children()
            pending = {r for r in pending if self._finished[r] is None}
            if time.monotonic() > deadline:
                raise TimeoutError(f"{len(pending)} requests unfinished after {timeout}s")

    def generate(self, prompts: Sequence[Sequence[int]], sampling_params: SamplingParams,
                 timeout: float = 3600.0, return_logprob: bool = False, top_logprobs_num: int = 0):
        """Blocking helper: token ids per prompt; with return_logprob also (ids, logprobs) where logprobs[i] =
        {"token": [...], "top": [[(logprob, token id), ...], ...]}."""
        import copy
        per_req = sampling_params if isinstance(sampling_params, (list, tuple)) else [sampling_params] * len(prompts)
        if len(per_req) != len(prompts):
            raise ValueError("generate: one SamplingParams per prompt (or a single one for all) expected")
        rids = [self.add_request(p, copy.deepcopy(sp), return_logprob=return_logprob,
                                 top_logprobs_num=top_logprobs_num) for p, sp in zip(prompts, per_req)]
        self.wait(rids, timeout)
        outs = [self._outputs[r] for r in rids]
        if return_logprob:
            return outs, [self._logprobs[r] for r in rids]
        return outs

    def request_record(self, rid: str) -> dict:
        return {"send": self._send_time[rid], "token_times": self._token_times[rid],
                "output_ids": self._outputs[rid], "finished": self._finished[rid]}

    def get_stats(self, reset: bool = False, expect: int = 2, timeout: float = 60.0) -> List[dict]:
        if self.scheduler is not None:
            out = dict(self.scheduler.stats)
            out["available_kv_slots"] = int(self.scheduler.token_to_kv_pool_allocator.available_size())
            out["num_running_reqs"] = len(self.scheduler.running_batch.reqs)
            out["num_waiting_reqs"] = len(self.scheduler.waiting_queue)
            kt = getattr(self.model_runner, "kernel_timing", None)
            if kt is not None:
                out["kernel_timing"] = kt.summary()
                if reset:
                    kt.reset()
            return [out]
        self._stats_inbox.clear()
        self.send_to_scheduler.send_pyobj(StatsReq(reset=reset))
        deadline = time.monotonic() + timeout
        while len(self._stats_inbox) < expect and time.monotonic() < deadline:
            self.poll(timeout=0.05)
        return list(self._stats_inbox)

    def shutdown(self):
        if self.scheduler is not None:
            self.scheduler = None
            return
        try:
            if self.is_driver:
                self.send_to_scheduler.send_pyobj(ShutdownReq())
        except Exception:
            pass
        # (a profiler attached to the scheduler processes writes its trace at exit: SEMIPD_SHUTDOWN_JOIN_S gives it time)
        join_s = float(os.environ.get("SEMIPD_SHUTDOWN_JOIN_S", "20"))
        for p in self.procs:
            p.join(timeout=join_s)
        for p in self.procs:
            if p.is_alive():
                logger.warning("scheduler process %s still alive %.0f s after the shutdown request: terminating it", p.name, join_s)
                p.terminate()
                p.join(timeout=5)
            elif p.exitcode not in (0, None):
                # (a process that dies on the way out also loses whatever a profiler attached to it would have written)
                logger.warning("scheduler process %s exited with code %s", p.name, p.exitcode)
        if self.is_driver and hasattr(self, "recv_from_scheduler"):
            self.recv_from_scheduler.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown()


def _query_cus(gpu_id, q, paths):
    for p in paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    from semi_pd_amd.semi_pd.utils import get_device_sm_count
    q.put(get_device_sm_count(gpu_id))
