#!/bin/bash
# round 6, call 10: the streaming GEMM's wide form (65 .. 128 rows): tests, kbench against the tiled GEMM, the decode step at 96 / 128
OUT=gpurun_out/r06_s10; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "stream_linear" > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.txt | cut -c1-300
KBENCH_KS=1,2,4 timeout 600 python tools/kbench_wide_rows.py 2>&1 | grep -v amdgpu.ids | tee $OUT/kbench_wide_rows.txt
for B in 96 128; do
  for W in 0 1; do
    SEMIPD_SL_WIDE=$W timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch $B --ctx 1100 2>&1 | grep "ms per decode step" | cut -c1-70 | sed "s/^/wide=$W /"
  done
done
