#!/bin/bash
# round 3: the driver's N = 2 command on one GPU with the DEFAULT masks (62 / 38 per rank) and the prefill GEMM tuning on
O=gpurun_out/r03_tp2_final; mkdir -p $O
SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --steps 1 --warmup 1 --num-requests 24 --request-rate 4 --fixed-load --no-cpu-baseline --mem-fraction-static 0.3 > $O/bench_tp2_one_gpu_masked.json 2> $O/bench_tp2.err
echo "rc=$?"; tail -c 1200 $O/bench_tp2_one_gpu_masked.json; echo; grep -v "amdgpu.ids\|^frame #\|UserWarning\|warnings.warn\|socket.cpp\|Gloo\|library GEMM\|^dtype=" $O/bench_tp2.err | tail -8 | cut -c1-300
