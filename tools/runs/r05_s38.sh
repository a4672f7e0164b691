#!/bin/bash
# round 5, call 38: the whole GPU suite on the final tree, smoke(), then the default bench line (5 timed waves) with its wall time
OUT=gpurun_out/r05_s38; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -14 $OUT/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s)
timeout 1500 python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python tools/summarize_runs.py $OUT/bench_default.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
for k in ("static_split_50_50", "unified_same_load", "saturation", "config1_opt_125m", "config3_deepseek_v2_lite"):
    v = d.get(k) or {}
    print(k, {kk: v.get(kk) for kk in ("timed_waves", "output_tok_s", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "value", "error") if kk in v})
print("qps_sweep", [(s["request_rate"], s["output_tok_s"], s["p50_ttft_ms"], s["p50_tbt_ms"], s["p99_tbt_ms"]) for s in d.get("qps_sweep", [])])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_estimated", "avg_launch_us", "launches_sampled")})
print("gate", d["roofline_extra"]["prefill_batch_ms"].get("step_gate"))
PY
