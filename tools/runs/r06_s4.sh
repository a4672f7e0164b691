#!/bin/bash
# round 6, call 4: the grouped (fused-MoE) form of the tiled GEMM on 8 / 4 waves; MoE parity tests in both forms
OUT=gpurun_out/r06_s4; mkdir -p $OUT
SEMIPD_G8_FORM=4 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "moe_gemm_tall or fused_experts or grouped_kernels_race" > $OUT/pytest_moe_form4.txt 2>&1; echo "pytest form4 rc=$?"; tail -2 $OUT/pytest_moe_form4.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm_tall or four_wave or fused_experts" > $OUT/pytest_auto.txt 2>&1; echo "pytest auto rc=$?"; tail -2 $OUT/pytest_auto.txt | cut -c1-300
timeout 600 python tools/kbench_moe_forms.py > $OUT/kbench_moe_forms.txt 2>&1; echo "kbench rc=$?"; cat $OUT/kbench_moe_forms.txt
