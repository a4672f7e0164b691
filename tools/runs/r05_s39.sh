#!/bin/bash
# round 5, call 39 (re-entry after the container was re-created; call 38's output was lost with it): the whole GPU suite on the
# committed tree (6aa5f6b), then smoke()
OUT=gpurun_out/r05_s39; mkdir -p $OUT
T0=$(date +%s)
timeout 1300 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -16 $OUT/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
