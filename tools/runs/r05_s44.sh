#!/bin/bash
# round 5, call 44: the planes form of the tiled GEMM again (test expectations corrected), then the engines that run prefill batches through it
OUT=gpurun_out/r05_s44; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "tall_planes or defers" > $OUT/pytest_tall.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_tall.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_engine.py -q -x --durations=4 -k "full_depth or 32_layers or matches_unified or chunked or tuning or tuned" > $OUT/pytest_engine.txt 2>&1; echo "pytest engine rc=$?"; tail -8 $OUT/pytest_engine.txt | cut -c1-200
