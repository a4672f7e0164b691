#!/bin/bash
# round 6, call 5: the tiled-GEMM tests after the form rule (K partition independent of the form), then the 4-wave form where it
# counts: the headline wave with SEMIPD_G8_FORM = 8 (round-5 kernels only) / 0 (by epilogue) alternating on one box
OUT=gpurun_out/r06_s5; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm_tall or four_wave or fused_experts or moe_gemm" > $OUT/pytest_auto.txt 2>&1; echo "pytest auto rc=$?"; tail -2 $OUT/pytest_auto.txt | cut -c1-300
bash tools/ab_in_situ.sh SEMIPD_G8_FORM 8 0 8 0 $OUT
grep -h "tiled" $OUT/run*.err | sort | uniq -c | sort -rn | head -30
