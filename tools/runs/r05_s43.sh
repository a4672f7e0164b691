#!/bin/bash
# round 5, call 43: the default bench line on the final tree (5 timed waves, every side wave), then rocprofv3 kernel stats + traces of BOTH instances
# of the default command's headline wave, the decode kernels split by whether prefill work was running (tools/trace_overlap.py)
OUT=gpurun_out/r05_s43; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
T0=$(date +%s)
timeout 1500 python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python tools/summarize_runs.py $OUT/bench_default.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
for k in ("static_split_50_50", "unified_same_load", "saturation", "config1_opt_125m", "config3_deepseek_v2_lite", "cpu_baseline"):
    v = d.get(k) or {}
    print(k, {kk: v.get(kk) for kk in ("timed_waves", "output_tok_s", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "value", "error") if kk in v})
print("qps_sweep", [(s["request_rate"], s["output_tok_s"], s["p50_ttft_ms"], s["p50_tbt_ms"], s["p99_tbt_ms"]) for s in d.get("qps_sweep", [])])
print("roofline", d["roofline"]); print("decode_attention", d["roofline_extra"].get("decode_attention")); print("gate", d["roofline_extra"]["prefill_batch_ms"].get("step_gate"))
PY
( cd /tmp && export TMPDIR=/tmp SEMIPD_SHUTDOWN_JOIN_S=180 && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-saturation-wave --no-side-configs --rate-sweep "" > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
echo "rocprof bench rc=$?"; grep -n "exited with code\|still alive\|SIGSEGV" $OUT/bench_under_rocprof.err | cut -c1-200
PT=""; DT=""
for f in $(find /tmp/bench_prof -name "*kernel_stats.csv"); do n=$(grep -c "extend_attn" $f); m=$(grep -c "decode_mfma" $f); t=${f/kernel_stats/kernel_trace}; if [ "$n" -gt 0 ]; then cp $f $OUT/prefill_process_kernel_stats.csv; PT=$t; elif [ "$m" -gt 0 ]; then cp $f $OUT/decode_process_kernel_stats.csv; DT=$t; fi; done
python tools/summarize_runs.py $OUT/bench_under_rocprof.json
if [ -n "$PT" ] && [ -n "$DT" ]; then
  python tools/trace_overlap.py $DT $PT 2>&1 | cut -c1-330 > $OUT/decode_kernels_by_overlap.txt
  python tools/trace_overlap.py $PT $DT 2>&1 | cut -c1-330 > $OUT/prefill_kernels_by_overlap.txt
  head -8 $OUT/decode_kernels_by_overlap.txt | cut -c1-300
fi
for f in $OUT/*_kernel_stats.csv; do python tools/stats_top.py $f | head -12 | cut -c1-150; done
