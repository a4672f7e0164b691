#!/bin/bash
# round 2, final tree: the default bench command under rocprofv3 (kernel stats of both processes)
OUT=gpurun_out/r02_final_prof; rm -rf $OUT; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
find $OUT/bench_prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
tail -c 300 $OUT/bench_under_rocprof.json; echo
for f in $(find $OUT/bench_prof -name "*kernel_stats.csv"); do echo "== $f"; python tools/stats_top.py $f | head -14; done
