#!/bin/bash
# round 5, call 24: bisecting the N = 2 one-GPU hang by configuration: no deadline / static masks / no kernel timing
OUT=gpurun_out/r05_s24; mkdir -p $OUT
go() { name=$1; shift; SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --model llama-tiny --steps 1 --warmup 1 --num-requests 48 --request-rate 8 --no-cpu-baseline --mem-fraction-static 0.3 --no-prefill-gemm-tuning "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python tools/summarize_runs.py $OUT/$name.json | cut -c1-200; sleep 3; }
go no_deadline --decode-step-deadline-ms 0
go env_masks --cu-mask-mode env --decode-step-deadline-ms 0
go no_timing --no-kernel-timing
