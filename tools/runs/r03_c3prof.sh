#!/bin/bash
OUT=gpurun_out/r03_c3prof; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python $R/bench.py --model deepseek-v2-lite --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 1 --warmup 1 > $R/$OUT/bench_c3.json 2> $R/$OUT/bench_c3.err )
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do python tools/stats_top.py $f | head -40; done
