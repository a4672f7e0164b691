#!/bin/bash
# round 4, call 10: add + norm inside the launch of the decode GEMM behind it (LaunchGate): parity with the separate launches, decode steps A / B
OUT=gpurun_out/r04_s10; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fused_prologue.py -q -x --durations=5 > $OUT/pytest_fused.txt 2>&1; echo "pytest fused rc=$?"
tail -5 $OUT/pytest_fused.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "rmsnorm or planes" > $OUT/pytest_norm.txt 2>&1; echo "pytest norm rc=$?"
tail -3 $OUT/pytest_norm.txt | cut -c1-200
for fuse in 1 0; do
  for m in llama3-8b llama3-70b-tp8-rank; do
    SEMIPD_FUSE_NORM_GEMM=$fuse timeout 300 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | sed "s/^/fuse=$fuse /"
  done
done > $OUT/steps.txt 2>&1
cut -c1-110 $OUT/steps.txt
SEMIPD_FUSE_NORM_GEMM=1 timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 20 --kernels 2>&1 | tail -16 | cut -c1-150 > $OUT/kernels_8b.txt
cat $OUT/kernels_8b.txt
timeout 900 python -m pytest tests/test_gpu_full_width.py tests/test_gpu_rank_widths.py -q -x > $OUT/pytest_engine.txt 2>&1; echo "pytest engine rc=$?"
tail -3 $OUT/pytest_engine.txt | cut -c1-200
