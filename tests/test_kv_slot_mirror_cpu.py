"""The allocating scheduler keeps a host mirror of every request's row of req_to_token (Req.kv_slots), so that KV
slots are released without reading device memory (the overlapped decode loop must never wait for the GPU there).
The mirror must equal the table row at every stage, and every slot must come back."""
import torch

from semi_pd_amd.managers.io_struct import SamplingParams
from semi_pd_amd.managers.schedule_batch import ChunkCache, Req, ScheduleBatch
from semi_pd_amd.mem_cache.memory_pool import ReqToTokenPool, TokenToKVPoolAllocator


def make(size=500, max_reqs=8, ctx=128):
    r2t = ReqToTokenPool(max_reqs, ctx, "cpu")
    alloc = TokenToKVPoolAllocator(size, torch.bfloat16, "cpu", None)
    return r2t, alloc, ChunkCache(r2t, alloc)


def row(r2t, req, n):
    return r2t.req_to_token[req.req_pool_idx, :n].tolist()


def test_mirror_tracks_the_table_through_extend_decode_finish_and_retract():
    r2t, alloc, cache = make()
    reqs = [Req(f"r{i}", list(range(10 + 7 * i)), SamplingParams(max_new_tokens=4, ignore_eos=True)) for i in range(3)]
    for r in reqs:
        r.init_next_round_input()
    batch = ScheduleBatch.init_new(reqs, r2t, alloc, cache, "cpu")
    batch.prepare_for_extend()
    for r in reqs:
        assert r.kv_slots == row(r2t, r, len(r.origin_input_ids)) and len(set(r.kv_slots)) == len(r.kv_slots)
    assert alloc.available_size() == 500 - sum(len(r.origin_input_ids) for r in reqs)
    batch.output_ids = torch.tensor([1, 2, 3])
    for r, t in zip(reqs, (1, 2, 3)):
        r.output_ids.append(t)
    for step in range(2):
        batch.prepare_for_decode()
        for r in reqs:
            n = len(r.origin_input_ids) + len(r.output_ids)
            assert r.kv_slots == row(r2t, r, n) and batch.seq_lens_cpu == batch.seq_lens.tolist()
        batch.output_ids = torch.tensor([5, 6, 7])
        for r in reqs:
            r.output_ids.append(5)
    # finish one request the way the plain loop does (KV for all tokens but the last sampled one) ...
    r0 = reqs[0]
    r0.finished_reason = "length"
    cache.cache_finished_req(r0)
    assert r0.kv_slots == []
    # ... one the way the overlapped loop does: a surplus step took one more slot, released afterwards
    r1 = reqs[1]
    batch.filter_batch()
    batch.prepare_for_decode()          # r1 (and r2) run once more
    r1.finished_reason = "length"
    cache.cache_finished_req(r1)
    assert len(r1.kv_slots) == 1      # the surplus step's slot stays with the request until that step is dropped
    alloc.free(r1.kv_slots)
    r1.kv_slots = []
    # ... and retract the last one
    batch.filter_batch()
    r2 = reqs[2]
    r2.output_ids.append(5)             # the surplus step's token of r2 has been processed
    retracted, _ = batch.retract_decode(force=1) if len(batch.reqs) > 1 else ([], 0)
    if not retracted:                   # a single request is never retracted: release it as finished
        r2.finished_reason = "length"
        cache.cache_finished_req(r2)
        alloc.free(r2.kv_slots)
    assert alloc.available_size() == 500 and r2t.available_size() == r2t.size
    assert sorted(alloc.free_slots.tolist()) == list(range(1, 501))


def test_allocator_is_fifo_and_host_side():
    _, alloc, _ = make(size=20)
    a = alloc.alloc(5)
    assert a.device.type == "cpu" and a.tolist() == [1, 2, 3, 4, 5]
    alloc.free([2, 4])
    alloc.free(torch.tensor([1]))
    assert alloc.available_size() == 18
    assert alloc.alloc(15).tolist() == list(range(6, 21))
    assert alloc.alloc(3).tolist() == [2, 4, 1]      # freed chunks come back in the order they were freed
    assert alloc.alloc(1) is None
    alloc.free_group_begin()
    alloc.free([7, 8])
    alloc.free([9])
    assert alloc.available_size() == 0              # grouped frees land together
    alloc.free_group_end()
    assert alloc.available_size() == 3 and alloc.alloc(3).tolist() == [7, 8, 9]
