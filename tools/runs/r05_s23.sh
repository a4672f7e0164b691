#!/bin/bash
# round 5, call 23: the N = 2 command shape on one GPU again, without start-up GEMM tuning: where is everybody after 90 s?
OUT=gpurun_out/r05_s23; mkdir -p $OUT
SEMIPD_DUMP_TRACEBACK_AFTER=90 SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --model llama-tiny --steps 1 --warmup 1 --num-requests 48 --request-rate 8 --no-cpu-baseline --mem-fraction-static 0.3 --no-prefill-gemm-tuning > $OUT/bench_tp2_one_gpu.json 2> $OUT/bench_tp2.err; echo "tp2 rc=$?"
grep -n "File \"/root/repo\|^Thread\|Current thread" $OUT/bench_tp2.err | cut -c1-200 | head -90
python tools/summarize_runs.py $OUT/bench_tp2_one_gpu.json
