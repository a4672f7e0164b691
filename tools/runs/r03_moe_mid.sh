#!/bin/bash
# round 3: fused MoE between the decode-sized streaming kernel and the 256-row tiles: 128-row geometry of gemm8p vs the
# round-1 tiled kernel, whole chip and on the 128-CU share
OUT=gpurun_out/r03_moe_mid; mkdir -p $OUT
export KBENCH_MOE_TS=384,512,768,1024,1536,2048,3072,4096
for mask in "" "0:0-127"; do
  for mid in 100000 40; do
    echo "# HSA_CU_MASK=$mask SEMIPD_MOE_MID_MIN_ROWS_PER_EXPERT=$mid"
    if [ -n "$mask" ]; then export HSA_CU_MASK=$mask; export KBENCH_NUM_CUS=128; else unset HSA_CU_MASK; fi
    SEMIPD_MOE_MID_MIN_ROWS_PER_EXPERT=$mid timeout 600 python tools/kbench.py moe 2>&1 | grep "^moe"
  done
done | tee $OUT/kbench_moe_mid.txt
