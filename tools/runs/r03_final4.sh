#!/bin/bash
# round 3, final checks on the final tree: smoke(), the whole GPU suite, the default bench line
OUT=gpurun_out/r03_final4; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt; grep -n "^E \|^FAILED" $OUT/pytest_gpu_full.txt | head -20
( time timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt
tail -3 $OUT/bench_time.txt
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["p50_ttft_ms"], d["p99_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("static", {k: d["static_split_50_50"][k] for k in ("output_tok_s","p50_ttft_ms","p50_tbt_ms","p99_tbt_ms")}); print("sat", d["saturation"]["output_tok_s"]); print("sweep", [(s["request_rate"], s["output_tok_s"], s["p50_ttft_ms"], s["p50_tbt_ms"], s["p99_tbt_ms"]) for s in d["qps_sweep"]])
PY
