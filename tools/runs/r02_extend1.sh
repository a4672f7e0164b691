#!/bin/bash
# round 2: extend attention FAST path (buffer loads for the new tokens, folded scale): parity, kbench, PMC
OUT=gpurun_out/r02_extend1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_gemm.py tests/test_gpu_fp8_kv.py -x -q -m gpu -k "extend or input_to_float8 or bmm_fp8" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 600 python tools/kbench.py extend > $OUT/kbench_extend.txt 2>&1; grep -v amdgpu.ids $OUT/kbench_extend.txt
bash tools/runs/r02_pmc_extend.sh > $OUT/pmc_extend.txt 2>&1; tail -6 $OUT/pmc_extend.txt
