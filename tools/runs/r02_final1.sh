#!/bin/bash
# round 2, final set: the default bench under rocprofv3 (kernel stats of both processes), config 3 line, default line
OUT=gpurun_out/r02_final1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
find $OUT/bench_prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
for f in $(find $OUT/bench_prof -name "*kernel_stats.csv"); do echo "== $f"; python tools/stats_top.py $f | head -12; done
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline > $OUT/bench_config3_deepseek_v2_lite.json 2> $OUT/c3.err; tail -c 600 $OUT/bench_config3_deepseek_v2_lite.json; echo
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json; echo
