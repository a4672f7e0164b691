#!/bin/bash
# round 5, call 10: kernel traces of both instances with the deadline gate on a 224-CU prefill share: what happens during a hold?
OUT=gpurun_out/r05_s10; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-saturation-wave --no-side-configs --rate-sweep "" --num-requests 16 --no-kernel-timing --prefill-cu 88 > /dev/null 2> $OUT/warm.err; echo "warm rc=$?"
( cd /tmp && export TMPDIR=/tmp SEMIPD_SHUTDOWN_JOIN_S=180 && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-saturation-wave --no-side-configs --rate-sweep "" --warmup 0 --steps 1 --num-requests 128 --prefill-cu 88 --decode-step-deadline-ms 8 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
echo "rocprof bench rc=$?"
PT=""; DT=""
for f in $(find /tmp/bench_prof -name "*kernel_stats.csv"); do n=$(grep -c "extend_attn" $f); m=$(grep -c "decode_mfma" $f); t=${f/kernel_stats/kernel_trace}; if [ "$n" -gt 0 ]; then PT=$t; elif [ "$m" -gt 0 ]; then DT=$t; fi; done
ls -la $PT $DT
python tools/summarize_runs.py $OUT/bench_under_rocprof.json
python tools/gate_trace.py $PT $DT 2>&1 | cut -c1-220 | tee $OUT/gate_trace.txt
