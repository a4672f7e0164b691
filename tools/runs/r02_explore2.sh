#!/bin/bash
# round 2, GPU exploration 2: can the library's persistent stream-K GEMM (SK3 MT256x256x64) be made to follow a 192-CU mask?
OUT=gpurun_out/r02_explore2; mkdir -p $OUT
export KBENCH_MS=1024,4096
probe() { ( for kv in "$@"; do export "$kv"; done; timeout 200 python tools/kbench.py linear_prefill ) >> $OUT/linear_prefill_env.txt 2>&1; }
probe HSA_CU_MASK=0:0-191 TENSILE_STREAMK_FIXED_GRID=192
probe HSA_CU_MASK=0:0-191 TENSILE_STREAMK_DYNAMIC_GRID=0
probe HSA_CU_MASK=0:0-191 TENSILE_STREAMK_DYNAMIC_GRID=1 TENSILE_STREAMK_MAX_CUS=192
probe HSA_CU_MASK=0:0-191 TENSILE_STREAMK_DYNAMIC_GRID=3 TENSILE_STREAMK_MAX_CUS=192
probe HSA_CU_MASK=0:0-191 TENSILE_STREAMK_GRID_MULTIPLIER=3
probe HSA_CU_MASK=0:0-191 KBENCH_BLAS=rocblas
probe HSA_CU_MASK=0:0-127 KBENCH_BLAS=rocblas
probe KBENCH_BLAS=rocblas
grep -v amdgpu.ids $OUT/linear_prefill_env.txt
