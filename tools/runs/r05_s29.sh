#!/bin/bash
# round 5, call 29: the two engine tests that time the library's GEMMs at start-up, then the default bench line on the final tree (5 timed waves)
OUT=gpurun_out/r05_s29; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "test_semi_pd_matches_unified or test_launch_server_semi_pd_http" --durations=3 > $OUT/pytest_two.txt 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_two.txt | cut -c1-200
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
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_estimated", "avg_launch_us", "launches_sampled")})
print("gate", d["roofline_extra"]["prefill_batch_ms"].get("step_gate"))
PY
