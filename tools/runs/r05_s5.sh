#!/bin/bash
# round 5, call 5: what caps the decode instance's 48 private CUs at 1.36 TB/s?  the probe on that CU set, on other sets of 48, on the whole
# chip: contiguous streams, the GEMM's loop, the GEMM's row-pitched access pattern
OUT=gpurun_out/r05_s5; mkdir -p $OUT
for r in "208 255" "0 47" "104 151" "0 23 232 255" "160 255" "0 255"; do
  timeout 120 tools/hbm_cu_probe range $r 2>&1 | grep -v "^$"
done | tee $OUT/hbm_probe_ranges.txt
