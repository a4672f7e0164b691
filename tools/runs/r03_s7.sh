#!/bin/bash
# round 3, call 7: the default bench line as the driver runs it (short: --steps 3), then the whole GPU suite
OUT=gpurun_out/r03_s7; mkdir -p $OUT
( time timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt
tail -3 $OUT/bench_time.txt
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["p50_ttft_ms"], d["p99_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"])
print(d["config"]["workload"])
print("roofline", d["roofline"])
print("extra", {k: v for k, v in d["roofline_extra"].items()})
print("static", d.get("static_split_50_50")); print("sat", d.get("saturation")); print("sweep", d.get("qps_sweep")); print("cpu", d.get("cpu_baseline"))
PY
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $OUT/pytest_gpu_full.txt
