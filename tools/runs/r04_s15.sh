#!/bin/bash
# round 4, call 15: the whole GPU suite with its clock (the driver's step limit is 1200 s), smoke
OUT=gpurun_out/r04_s15; mkdir -p $OUT
SECONDS=0
timeout 1700 python -m pytest tests/ -q -x -m gpu --durations=40 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? wall=${SECONDS}s"
tail -60 $OUT/pytest_gpu.txt | cut -c1-180
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
