#!/bin/bash
# round 5, call 18: the MoE tests again (no k-block rotation in the grouped streaming GEMM), DeepSeek engine tests, then the PMC passes of call 17
OUT=gpurun_out/r05_s18; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_deepseek.py -q -k "moe or deepseek" --durations=5 > $OUT/pytest_moe.txt 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest_moe.txt | cut -c1-200
bash tools/runs/r05_s17.sh
