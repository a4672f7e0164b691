#!/bin/bash
# round 2: bmm_fp8 + DeepSeek tests, default bench with the streaming-GEMM roofline, PMC breakdown of extend attention
OUT=gpurun_out/r02_full2; mkdir -p $OUT
( time timeout 1800 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_deepseek.py -x -q -m gpu ) > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
bash tools/runs/r02_pmc_extend.sh > $OUT/pmc_extend.txt 2>&1; tail -8 $OUT/pmc_extend.txt
