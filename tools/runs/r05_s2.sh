#!/bin/bash
# round 5, call 2: why the prefill process leaves no rocprofv3 output (exit codes at shutdown), then the traces of both instances
OUT=gpurun_out/r05_s2; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp SEMIPD_SHUTDOWN_JOIN_S=180 SEMIPD_LOGLEVEL=WARNING && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof0 -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-side-configs --rate-sweep "" --num-requests 32 --warmup 0 > $R/$OUT/small.json 2> $R/$OUT/small.err )
echo "small rc=$?"; grep -n "exited with code\|still alive\|Opened result file" $OUT/small.err | cut -c1-200
find /tmp/bench_prof0 -name "*.csv" | xargs ls -la
