#!/bin/bash
OUT=gpurun_out/r02_c3prof; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python $R/bench.py --model deepseek-v2-lite --no-cpu-baseline --no-static-split-wave --no-saturation-wave > $R/$OUT/bench.json 2> $R/$OUT/bench.err )
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do echo "== $f"; python tools/stats_top.py $f | head -22; done
