#!/bin/bash
# round 5, call 41: the whole GPU suite + smoke() on the final tree (custom masks in extend attention, full-depth CPU baseline)
OUT=gpurun_out/r05_s41; mkdir -p $OUT
T0=$(date +%s)
timeout 1300 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -16 $OUT/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
