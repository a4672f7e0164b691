#!/bin/bash
# round 4, call 21: MLA decode (16 heads) with two tiles in flight in registers: per split count, against the previous build; parity tests
OUT=gpurun_out/r04_s21; mkdir -p $OUT
echo "== previous build (one tile in flight)" | tee $OUT/kbench_mla16.txt
SEMIPD_HIP_LIB=$PWD/semi-pd_amd/lib/libsemipd_hip_prev.so timeout 300 python tools/kbench_mla16.py 2>&1 | grep -v Warning | tee -a $OUT/kbench_mla16.txt
echo "== this build (two tiles in flight)" | tee -a $OUT/kbench_mla16.txt
timeout 300 python tools/kbench_mla16.py 2>&1 | grep -v Warning | tee -a $OUT/kbench_mla16.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py -q -x -k "mla or decode" > $OUT/pytest_mla.txt 2>&1; echo "pytest mla rc=$?"
tail -3 $OUT/pytest_mla.txt | cut -c1-200
