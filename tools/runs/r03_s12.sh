#!/bin/bash
# round 3, call 12: MoE tall GEMM geometry (256-row blocks / 256 x 256 tiles vs 128-row blocks / 128 x 512 tiles)
OUT=gpurun_out/r03_s12; mkdir -p $OUT
{
for bm in 256 128; do
  echo "## SEMIPD_MOE_TALL_BLOCK_M=$bm (tall from T = 1024)"
  SEMIPD_MOE_TALL_BLOCK_M=$bm SEMIPD_MOE_TALL_MIN_ROWS=4096 SEMIPD_MOE_TALL_MIN_ROWS_PER_EXPERT=64 KBENCH_MOE_TS=1024,2048,4096,8192,16384 timeout 600 python tools/kbench.py moe
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/moe_tall_geometry.txt | cut -c1-120
SEMIPD_MOE_TALL_BLOCK_M=128 timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "moe_gemm_tall or fused_experts_takes" 2>&1 | tail -3
