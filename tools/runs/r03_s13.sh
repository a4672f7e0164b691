#!/bin/bash
# round 3, call 13: tile order of the ping-pong GEMM for several row tiles (XCD-contiguous, row tiles fastest) A/B; the
# streaming GEMM on the whole chip with the share-aware K split
OUT=gpurun_out/r03_s13; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "gemm_tall" 2>&1 | tail -2
{
for o in 1 0; do
  echo "## SEMIPD_G8_XCD_ORDER=$o whole chip"; SEMIPD_G8_XCD_ORDER=$o KBENCH_MS=1024,2048,4096,8192 timeout 600 python tools/kbench.py gemm_tall
  echo "## SEMIPD_G8_XCD_ORDER=$o 160 CUs"; SEMIPD_G8_XCD_ORDER=$o HSA_CU_MASK=0:0-159 KBENCH_NUM_CUS=160 KBENCH_MS=1024,2048 timeout 600 python tools/kbench.py gemm_tall
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_tall_tile_order.txt | cut -c1-150
KBENCH_NUM_CUS=256 KBENCH_MS=16,32,64 KBENCH_SL_SWEEP=0 timeout 600 python tools/kbench.py stream_linear 2>&1 | grep -v amdgpu.ids | tee $OUT/stream_linear_whole_chip.txt | cut -c1-120
