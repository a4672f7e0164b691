#!/bin/bash
# round 2: full GPU suite, the default bench line (as the driver runs it), PMC breakdown of extend attention
OUT=gpurun_out/r02_full1; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 2500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
bash tools/runs/r02_pmc_extend.sh > $OUT/pmc_extend.txt 2>&1; tail -8 $OUT/pmc_extend.txt
