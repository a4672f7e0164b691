#!/bin/bash
# round 4: rocprofv3 kernel stats of the default bench command (headline wave only) with the library GEMM table already cached
# (a first short run tunes it: otherwise the decode process's statistics hold 28 s of synthetic decode replays next to the tuner)
OUT=gpurun_out/r04_final_prof; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-side-configs --rate-sweep "" --num-requests 16 --no-kernel-timing > /dev/null 2> $OUT/warm.err; echo "warm rc=$?"
( cd /tmp && export TMPDIR=/tmp SEMIPD_SHUTDOWN_JOIN_S=180 && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-side-configs --rate-sweep "" > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
find /tmp/bench_prof -name "*kernel_stats.csv" | xargs wc -l
rm -f $OUT/*_kernel_stats.csv
for f in $(find /tmp/bench_prof -name "*kernel_stats.csv"); do n=$(grep -c "extend_attn" $f); m=$(grep -c "decode_mfma" $f); if [ "$n" -gt 0 ]; then cp $f $OUT/prefill_process_kernel_stats.csv; elif [ "$m" -gt 0 ]; then cp $f $OUT/decode_process_kernel_stats.csv; fi; done
python - <<PY
import json
d = json.loads(open("$OUT/bench_under_rocprof.json").read().strip().splitlines()[-1])
print("under rocprof:", d["value"], "TTFT", d["p50_ttft_ms"], "TBT", d["p50_tbt_ms"], d["p99_tbt_ms"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "avg_launch_us_minus_event_overhead", "launches_sampled")})
PY
for f in $OUT/*_kernel_stats.csv; do python tools/stats_top.py $f | head -14 | cut -c1-150; done
