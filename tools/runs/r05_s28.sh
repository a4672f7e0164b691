#!/bin/bash
# round 5, call 28: the whole GPU suite on the final tree, then smoke()
OUT=gpurun_out/r05_s28; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -22 $OUT/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
