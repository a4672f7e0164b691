"""Decode hipGraph runner (reference: model_executor/cuda_graph_runner.py:109-150 batch-size list,
:455-531 replay with padding).  One graph per batch-size bucket; the graph contains the metadata
kernels (kv_indptr / kv_indices), every layer and the lm_head + argmax, so a decode step is one
hipGraphLaunch.  Padded slots use seq_len = 1 and write their KV to the dummy slot 0."""
from __future__ import annotations

import bisect
import os
from typing import Dict, List, Tuple

import torch

from semi_pd_amd.model_executor.forward_batch_info import ForwardBatch, ForwardMode


def get_batch_sizes_to_capture(max_bs: int) -> List[int]:
    """cuda_graph_runner.py:109-150: [1, 2, 4] + multiples of 8 up to 160 (HIP: up to 256); above that
    (large TP groups serving many requests) coarser steps up to 1024 so that a burst does not fall off
    the graphs into eager launches."""
    bs = [1, 2, 4] + [8 * i for i in range(1, 33)] + list(range(288, 513, 32)) + list(range(576, 1025, 64))
    return [b for b in bs if b <= max_bs]


_PARKED: list = []   # see _capture_one

# fault injection for tests: a callable (bs, outputs) run inside the capture.  Installed only by a test plugin that the
# scheduler process loads at start-up (entrypoints/engine.py: ServerArgs.test_plugin); never set in production.
capture_fault_hook = None


def _destroy_stream(raw: int) -> None:
    try:
        from semi_pd_amd import _lib
        _lib.load().semipd_stream_destroy(raw)
    except Exception:  # noqa: BLE001  (interpreter shutdown)
        pass


def recover_after_failed_capture(device) -> None:
    """After HipGraphRunner raised out of a capture (its stream is destroyed, what lived on it parked): the sequence that
    brings the process back on ROCm 7.0 (tools/probe_capture_abort.py, tests/test_gpu_ops.py::
    test_abort_of_a_failed_stream_capture) -- one throw-away call takes the hipErrorInvalidValue the framework's
    bookkeeping left behind, one small clean capture on a fresh stream settles its capture state."""
    import ctypes
    from semi_pd_amd import _lib
    lib = _lib.load()
    x = None
    for _ in range(2):
        try:
            lib.semipd_clear_last_error()
            x = torch.ones(8, device=device)
            x.cpu()
            break
        except Exception:  # noqa: BLE001
            continue
    raw = ctypes.c_void_p()
    _lib.check(lib.semipd_stream_create(torch.device(device).index or 0, ctypes.addressof(raw)), "stream_create")
    st = torch.cuda.ExternalStream(raw.value, device=device)
    st.wait_stream(torch.cuda.current_stream())
    if x is None:
        x = torch.ones(8, device=device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        y = x * 2
    _PARKED.append((x, y, g, st))
    torch.cuda.synchronize()


class HipGraphRunner:
    def __init__(self, model_runner):
        self.mr = model_runner
        dev = model_runner.device
        self.capture_bs = get_batch_sizes_to_capture(min(model_runner.cuda_graph_max_bs,
                                                         model_runner.max_running_requests))
        self.max_bs = max(self.capture_bs)
        self.seq_len_fill_value = model_runner.attn_backend.get_cuda_graph_seq_len_fill_value()
        with torch.device(dev):
            self.input_ids = torch.zeros(self.max_bs, dtype=torch.int64)
            self.req_pool_indices = torch.zeros(self.max_bs, dtype=torch.int64)
            self.seq_lens = torch.full((self.max_bs,), self.seq_len_fill_value, dtype=torch.int64)
            self.out_cache_loc = torch.zeros(self.max_bs, dtype=torch.int64)
            self.positions = torch.zeros(self.max_bs, dtype=torch.int64)
        model_runner.attn_backend.init_cuda_graph_state(self.max_bs)
        self.graphs: Dict[Tuple[int, int], torch.cuda.CUDAGraph] = {}      # (CUs of the stream, batch size) -> graph
        self.outputs: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self.pool = None
        # the capture stream is OURS (not one of torch's pooled streams): a capture that fails half way can only be got rid
        # of by destroying its stream (include/semipd.h, semipd_stream_abort_capture)
        import ctypes
        from semi_pd_amd import _lib
        raw = ctypes.c_void_p()
        _lib.check(_lib.load().semipd_stream_create(torch.device(dev).index or 0, ctypes.addressof(raw)), "stream_create")
        self._raw_stream = raw.value
        self.stream = torch.cuda.ExternalStream(self._raw_stream, device=dev)
        import weakref
        self._stream_finalizer = weakref.finalize(self, _destroy_stream, self._raw_stream)
        # one set of graphs per CU count the instance may run a step on (--cu-mask-mode dynamic with a decode share below
        # 100 %: its masked stream and the whole chip; model_executor/cu_share.py): grids, K splits and split-KV counts
        # are baked into a graph at capture
        share = getattr(model_runner, "cu_share", None)
        self.variants = sorted(set(share.cus.values())) if share is not None else [model_runner.num_cus_owned]
        entered = model_runner.num_cus_owned
        try:
            for cus in self.variants:
                if len(self.variants) > 1:
                    model_runner.set_owned_cus(cus)
                self._cus = cus
                for bs in reversed(self.capture_bs):
                    self._capture_one(bs)
        finally:
            if len(self.variants) > 1:
                model_runner.set_owned_cus(entered)

    def _capture_one(self, bs: int):
        mr = self.mr
        fb = ForwardBatch(
            forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=self.input_ids[:bs],
            req_pool_indices=self.req_pool_indices[:bs], seq_lens=self.seq_lens[:bs],
            out_cache_loc=self.out_cache_loc[:bs], seq_lens_sum=bs, positions=self.positions[:bs],
            req_to_token_pool=mr.req_to_token_pool, token_to_kv_pool=mr.token_to_kv_pool,
            attn_backend=mr.attn_backend)

        def run_once():
            torch.clamp(self.seq_lens[:bs] - 1, min=0, out=self.positions[:bs])
            mr.attn_backend.init_forward_metadata_capture_cuda_graph(
                bs, bs, self.req_pool_indices[:bs], self.seq_lens[:bs], None, ForwardMode.DECODE, None)
            out = mr.model.forward(fb.input_ids, fb.positions, fb)
            ids = out.next_token_ids if out.next_token_ids is not None else mr.sampler(out)
            return out.next_token_logits, ids

        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(2):  # warm up allocator / hipBLASLt heuristics outside the capture
                run_once()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, pool=self.pool, stream=self.stream):
                out = run_once()
                if capture_fault_hook is not None:
                    capture_fault_hook(bs, out)
        except BaseException:
            # something in the step refused capture (a collective of this TP backend, say): the invalidated capture
            # poisons every later synchronising call of the process until its stream is destroyed
            from semi_pd_amd import _lib
            _lib.load().semipd_stream_abort_capture(self._raw_stream)
            self._raw_stream = None
            self._stream_finalizer.detach()   # the abort destroyed the stream
            # nothing that lives on the dead stream may be freed (the allocator would touch the stream again): the graphs,
            # their outputs and the runner itself are parked for the life of the process
            _PARKED.append((self, g, self.graphs, self.outputs, self.pool, self.stream))
            raise
        self.pool = self.pool or g.pool()
        self.graphs[(self._cus, bs)] = g
        self.outputs[(self._cus, bs)] = out

    def can_run(self, forward_batch: ForwardBatch) -> bool:
        return forward_batch.forward_mode.is_decode() and forward_batch.batch_size <= self.max_bs

    def replay(self, forward_batch: ForwardBatch):
        from semi_pd_amd.layers.basic import LogitsProcessorOutput
        raw_bs = forward_batch.batch_size
        bs = self.capture_bs[bisect.bisect_left(self.capture_bs, raw_bs)]
        if bs != raw_bs:
            self.seq_lens.fill_(self.seq_len_fill_value)
            self.out_cache_loc.zero_()
            self.req_pool_indices.zero_()
        self.input_ids[:raw_bs].copy_(forward_batch.input_ids)
        self.req_pool_indices[:raw_bs].copy_(forward_batch.req_pool_indices)
        self.seq_lens[:raw_bs].copy_(forward_batch.seq_lens)
        self.out_cache_loc[:raw_bs].copy_(forward_batch.out_cache_loc)
        # the variant of the stream the instance is on (its OWN CU count: an experiment that declares another count to the
        # kernels, SEMIPD_DECLARED_CUS_*, must not change which graph is found)
        share = getattr(self.mr, "cu_share", None)
        cus = share.cus[share.active] if (share is not None and share.active and len(self.variants) > 1) else self.variants[0]
        key = (cus, bs)
        self.graphs[key].replay()
        logits, ids = self.outputs[key]
        return LogitsProcessorOutput(logits[:raw_bs] if logits is not None else None, next_token_ids=ids[:raw_bs])
