#!/bin/bash
# round 3: the driver's N = 8 command shape on ONE GPU (all 8 ranks on GPU 0, gloo + the peer-memory all-reduce kernels):
# 8 prefill + 8 decode scheduler processes, per-rank IPC handles, ports, capture-abort fallback -- a functional dry run,
# no scaling claim.  llama-tiny (hidden 1024, 8 / 2 heads) so that heads and KV heads shard 8 ways... kv heads = 2 < 8:
# replicated KV heads, the models/llama.py:115-131 rule.
O=gpurun_out/r03_dryrun_n8_masked; mkdir -p $O
SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
  --master-port 29711 bench.py --gpus 8 --steps 1 --warmup 1 --model llama-tiny --num-requests 4 --request-rate 1 --fixed-load \
  --input-len 256 --output-len 32 --no-cpu-baseline --mem-fraction-static 0.05 --max-total-tokens 20000 --max-running-requests 4 \
  --rate-sweep "" --no-saturation-wave > $O/bench_gpus8_tp_dry_run_one_gpu.json 2> $O/bench_gpus8.err
echo "rc=$?"; tail -c 1500 $O/bench_gpus8_tp_dry_run_one_gpu.json; echo; tail -5 $O/bench_gpus8.err | cut -c1-300
