#!/bin/bash
# round 4, call 28: gemm8p (gemm_tall) against the library at prefill row counts on a 208-CU share (the default prefill share) and on the whole chip
OUT=gpurun_out/r04_s28; mkdir -p $OUT
HSA_CU_MASK=0:0-207 KBENCH_NUM_CUS=208 KBENCH_MS=1024,1408,2048 timeout 600 python tools/kbench.py gemm_tall 2>&1 | grep -v Warning | tee $OUT/gemm_tall_208cus.txt
KBENCH_MS=1024,2048 timeout 600 python tools/kbench.py gemm_tall 2>&1 | grep -v Warning | tee $OUT/gemm_tall_whole_chip.txt
