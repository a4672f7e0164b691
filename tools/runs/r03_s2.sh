#!/bin/bash
# round 3, call 2: (1) which hipBLASLt solutions win under prefill masks (first heuristic / best of 64 heuristics / best of all),
# (2) rocprofv3 kernel stats of both instances under P62/D38
OUT=gpurun_out/r03_s2; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/blaslt_probe.cpp -o /tmp/blaslt_probe -lhipblaslt 2>/dev/null
SHAPES=""
for M in 256 512 1024 2048 4096; do SHAPES="$SHAPES $M 28672 4096 $M 4096 14336 $M 6144 4096 $M 4096 4096"; done
{
for mask in none 0:0-127 0:0-159 0:0-191 0:0-223; do
  if [ "$mask" = none ]; then env -u HSA_CU_MASK timeout 900 /tmp/blaslt_probe $SHAPES; else HSA_CU_MASK=$mask timeout 900 /tmp/blaslt_probe $SHAPES; fi
done
} > $OUT/blaslt_probe.txt 2>&1
tail -5 $OUT/blaslt_probe.txt
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench_prof -- python $R/bench.py --prefill-cu 62 --decode-cu 38 --no-cpu-baseline --no-static-split-wave --no-saturation-wave > $R/$OUT/bench_p62_d38_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
find $OUT/bench_prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
tail -c 300 $OUT/bench_p62_d38_under_rocprof.json; echo
for f in $(find $OUT/bench_prof -name "*kernel_stats.csv"); do python tools/stats_top.py $f | head -22; done
