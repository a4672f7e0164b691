"""GPU time of ONE decode step (one hipGraph replay) of a model at a fixed batch and context, no scheduler, no second
instance: the number VERDICT r03 item 3 sets bars on (Llama-3-8B B = 32, llama3-70b-tp8-rank, deepseek-v3-tp8-rank).

    python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 [--quantization fp8] [--eager]

Prints the step time, the weight bytes a step reads and the floor that implies at 8 TB/s, and -- with --kernels -- the
per-kernel launch count of one step (from torch's profiler, kernel names only)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=1100)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--quantization", default=None)
    ap.add_argument("--kernels", action="store_true")
    args = ap.parse_args()
    import bench
    from semi_pd_amd.model_executor.model_runner import ModelRunner
    cfg = bench.model_config(args.model)
    if args.quantization == "fp8":
        import dataclasses
        cfg = dataclasses.replace(cfg, quantization_config={"quant_method": "fp8", "weight_block_size": [128, 128],
                                                            "activation_scheme": "dynamic"})
    bs, ctx = args.batch, args.ctx
    mr = ModelRunner(cfg, context_length=ctx + 8, max_running_requests=bs, max_total_tokens=bs * (ctx + 8) + 64,
                     cuda_graph_max_bs=bs, mem_fraction_static=0.8)
    mr.init_attention_backend()
    t0 = time.time()
    mr.init_cuda_graphs()
    gr = mr.graph_runner
    dev = mr.device
    r2t = mr.req_to_token_pool.req_to_token
    for r in range(bs):
        r2t[r, : ctx + 1] = torch.arange(r * (ctx + 1), (r + 1) * (ctx + 1), device=dev, dtype=r2t.dtype) + 1
    n = max(b for _, b in gr.graphs if b <= bs)
    key = next(k for k in gr.graphs if k[1] == n)
    gr.req_pool_indices[:n] = torch.arange(n, device=dev)
    gr.seq_lens[:n] = ctx + 1
    gr.out_cache_loc[:n] = r2t[:n, ctx].to(torch.int64)
    gr.input_ids[:n] = torch.randint(0, cfg.vocab_size, (n,), device=dev)
    g = gr.graphs[key]
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    wbytes = sum(p.numel() * p.element_size() for n_, p in mr.model.named_parameters() if "embed_tokens" not in n_)
    moe = sum(p.numel() * p.element_size() for n_, p in mr.model.named_parameters() if "experts.w" in n_)
    print(f"{args.model}{' fp8' if args.quantization else ''} B={n} ctx={ctx}: {ms:.3f} ms per decode step (graph replay, "
          f"{args.steps} steps); weights {wbytes / 1e9:.2f} GB (of which routed experts {moe / 1e9:.2f} GB, touched only in "
          f"part) -> floor {wbytes / 8e12 * 1e3:.2f} ms at 8 TB/s; graphs captured in {time.time() - t0:.1f} s", flush=True)
    if args.kernels:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            g.replay()
            torch.cuda.synchronize()
        rows = sorted(((e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0),
                      key=lambda r: -r[2])
        total_n = sum(r[1] for r in rows)
        total_t = sum(r[2] for r in rows)
        print(f"  {total_n} kernel launches, {total_t / 1e3:.3f} ms of kernel time in one step")
        if total_n == 0:
            print("  (torch's profiler sees no kernels of a hipGraph replay on this image: run the command without --kernels under\n"
                  "   `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/decode_step_bench.py ... --steps 200`\n"
                  "   and read <dir>/**/*kernel_stats.csv with tools/stats_top.py -- profiles/r06_decode_step_*_kernel_stats.csv were made so)")
        for k, c, t in rows[:24]:
            print(f"  {c:5d} x {t / c:8.1f} us = {t / 1e3:7.3f} ms  {k[:100]}")


if __name__ == "__main__":
    main()
