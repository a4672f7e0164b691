#!/bin/bash
# round 5, call 49: custom masks at the absorbed-MLA row shape (576 / 512: the one-wave kernel), golden + random
OUT=gpurun_out/r05_s49; mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_ops.py -q -k "custom_mask" > $OUT/pytest_mask.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_mask.txt | cut -c1-200
