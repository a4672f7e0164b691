#!/bin/bash
# round 3: the two forms of the shared-KV extend kernel on the 160-CU prefill share (1 and 2 requests of 1024 tokens)
OUT=gpurun_out/r03_extend_forms; mkdir -p $OUT
export HSA_CU_MASK=0:0-159
for form in 2 1; do
  echo "# HSA_CU_MASK=$HSA_CU_MASK SEMIPD_EXTEND_KV_FORM=$form"
  SEMIPD_EXTEND_KV_FORM=$form timeout 300 python tools/kbench.py extend 2>&1 | grep "^extend" | grep "Hq=32"
done | tee $OUT/kbench_extend_forms_160cu.txt
