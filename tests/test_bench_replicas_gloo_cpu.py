"""bench.py --gpus N (N > 1) runs N independent Semi-PD replicas; what the ranks exchange is covered here with two
gloo processes on the CPU: the timed region is the slowest rank's, rank 0 reports over every replica's requests."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, port, q):
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]
        import torch.distributed as dist
        import bench
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        # replica `rank` served 3 + rank requests of 4 tokens each in (1 + rank) seconds
        records = [{"send": 10.0 * rank, "token_times": [10.0 * rank + 0.1 * (i + 1) + 0.01 * j for j in range(4)],
                    "output_ids": [1, 2, 3, 4], "finished": "length"} for i in range(3 + rank)]
        recs, elapsed = bench.combine_ranks(records, 1.0 + rank, rank, world, replicas=True)
        out = {"n": len(recs), "elapsed": elapsed}
        if rank == 0:
            out["summary"] = bench.summarize(recs, elapsed)
        # a tensor-parallel engine is driven by rank 0: only the time is combined
        recs_tp, elapsed_tp = bench.combine_ranks(records, 1.0 + rank, rank, world, replicas=False)
        out["n_tp"], out["elapsed_tp"] = len(recs_tp), elapsed_tp
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_two_replicas_are_combined_on_rank_zero():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 24000 + os.getpid() % 4000
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert "error" not in got[0] and "error" not in got[1], (got[0].get("error"), got[1].get("error"))
    assert got[0]["elapsed"] == got[1]["elapsed"] == 2.0            # MAX over ranks
    assert got[0]["n"] == 3 + 4 and got[1]["n"] == 4                # rank 0 holds every replica's requests
    s = got[0]["summary"]
    assert abs(s["output_tok_s"] - (7 * 4) / 2.0) < 1e-9            # whole-job tokens / slowest replica's time
    assert got[0]["n_tp"] == 3 and got[0]["elapsed_tp"] == 2.0
