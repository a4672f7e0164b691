#!/bin/bash
# round 5, call 27: the N = 2 command shape on one GPU with the defaults (kernel timing on, pacer on) after the sampling fix
OUT=gpurun_out/r05_s27; mkdir -p $OUT
SEMIPD_DUMP_TRACEBACK_AFTER=200 SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --model llama-tiny --steps 1 --warmup 1 --num-requests 48 --request-rate 8 --no-cpu-baseline --mem-fraction-static 0.3 --no-prefill-gemm-tuning > $OUT/bench_tp2_one_gpu.json 2> $OUT/bench_tp2.err; echo "tp2 rc=$?"
python tools/summarize_runs.py $OUT/bench_tp2_one_gpu.json | cut -c1-300
python - <<PY
import json
d = json.loads(open("$OUT/bench_tp2_one_gpu.json").read().strip().splitlines()[-1])
print(d["config"]["workload"][:200]); print(d.get("tensor_parallel")); print("roofline", (d.get("roofline") or {}).get("launches_sampled"), d["roofline_extra"]["prefill_batch_ms"].get("step_gate"))
PY
