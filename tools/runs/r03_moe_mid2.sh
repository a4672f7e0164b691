#!/bin/bash
# round 3: where the 128-row geometry hands over to the 256-row one (rows per expert), 128-CU share and whole chip
OUT=gpurun_out/r03_moe_mid; mkdir -p $OUT
export KBENCH_MOE_TS=1024,1152,1280,1408,1536,1792,2048
for mask in "0:0-127" ""; do
  if [ -n "$mask" ]; then export HSA_CU_MASK=$mask; export KBENCH_NUM_CUS=128; else unset HSA_CU_MASK; unset KBENCH_NUM_CUS; fi
  echo "# HSA_CU_MASK=$mask 128-row geometry below the 256-row bounds (12288 rows, 192 per expert)"
  timeout 600 python tools/kbench.py moe 2>&1 | grep "^moe"
  echo "# HSA_CU_MASK=$mask 256-row geometry from 100 rows per expert"
  SEMIPD_MOE_TALL_MIN_ROWS=0 SEMIPD_MOE_TALL_MIN_ROWS_PER_EXPERT=100 timeout 600 python tools/kbench.py moe 2>&1 | grep "^moe"
done | tee $OUT/kbench_moe_mid_handover.txt
