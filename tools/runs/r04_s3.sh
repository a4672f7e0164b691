#!/bin/bash
# round 4, call 3: per-CU HBM probe share by share; stream-mask vs env-mask placement; new parity tests; decode step times
OUT=gpurun_out/r04_s3; mkdir -p $OUT
: > $OUT/hbm_cu_probe.txt
for sh in "8 8" "32 8" "64 8" "96 8" "128 8" "160 8" "192 8" "256 8" "96 4" "96 3" "32 1" "64 2"; do
  timeout 60 tools/hbm_cu_probe $sh >> $OUT/hbm_cu_probe.txt 2>&1 || echo "share $sh: rc=$? (timeout 60 s)" >> $OUT/hbm_cu_probe.txt
done
timeout 60 tools/hbm_cu_probe pair >> $OUT/hbm_cu_probe.txt 2>&1 || echo "pair rc=$?" >> $OUT/hbm_cu_probe.txt
timeout 300 python tools/cu_mask_check.py --streams > $OUT/cu_mask_streams.txt 2>&1; echo "mask check rc=$?"
timeout 900 python -m pytest tests/test_gpu_cu_share.py tests/test_gpu_rank_widths.py tests/test_gpu_full_width.py -x -q -s --durations=10 > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?"
for m in "llama3-8b" "llama3-70b-tp8-rank" ; do
  timeout 600 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --kernels > $OUT/step_$m.txt 2>&1; echo "step $m rc=$?"
done
timeout 900 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --kernels > $OUT/step_v3rank.txt 2>&1; echo "step v3 rc=$?"
tail -5 $OUT/pytest_new.txt; grep "ms per decode\|kernel launches" $OUT/step_*.txt; cat $OUT/cu_mask_streams.txt | tail -8
