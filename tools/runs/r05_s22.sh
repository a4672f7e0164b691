#!/bin/bash
# round 5, call 22: where does the N = 2 command shape on one GPU hang (traceback of every scheduler process after 120 s)?  the retract test under the
# default policy; what takes 300 s in the full-depth DeepSeek test
OUT=gpurun_out/r05_s22; mkdir -p $OUT
SEMIPD_DUMP_TRACEBACK_AFTER=120 SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --model llama-tiny --steps 1 --warmup 1 --num-requests 48 --request-rate 8 --no-cpu-baseline --mem-fraction-static 0.3 > $OUT/bench_tp2_one_gpu.json 2> $OUT/bench_tp2.err; echo "tp2 rc=$?"
grep -n "File \"\|Thread\|Current thread\|most recent call" $OUT/bench_tp2.err | grep -v "torch/distributed/elastic\|runpy" | cut -c1-200 | head -80
timeout 600 python -m pytest tests/test_gpu_cu_share.py -q -x -k retract > $OUT/pytest_retract.txt 2>&1; echo "retract rc=$?"; tail -5 $OUT/pytest_retract.txt | cut -c1-200
T0=$(date +%s); timeout 900 python -m pytest tests/test_gpu_full_depth.py -q -x -s -k deepseek > $OUT/pytest_depth.txt 2>&1; echo "depth rc=$? in $(( $(date +%s) - T0 )) s"; grep -n "WARNING\|step_gate\|equal\|passed\|failed" $OUT/pytest_depth.txt | cut -c1-220 | head -20
