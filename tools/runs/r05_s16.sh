#!/bin/bash
# round 5, call 16: the whole GPU suite on the tree with the new defaults
OUT=gpurun_out/r05_s16; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -40 $OUT/pytest_gpu.txt | cut -c1-200
