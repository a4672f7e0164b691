#!/bin/bash
# round 4, final tree: the whole GPU suite with its clock, smoke, the default bench command as the driver runs it
OUT=gpurun_out/r04_final; mkdir -p $OUT
SECONDS=0
timeout 1700 python -m pytest tests/ -q -x -m gpu --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? wall=${SECONDS}s"
tail -22 $OUT/pytest_gpu.txt | cut -c1-180
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
SECONDS=0
timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
s = d.get("saturation") or {}
print("default", d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), "sat", s.get("output_tok_s"), "frac", d["roofline"]["frac"])
for k in ("config1_opt_125m", "config3_deepseek_v2_lite", "static_split_50_50"):
    v = d.get(k)
    print(k, {kk: v[kk] for kk in v if kk.startswith("p") or kk == "output_tok_s"} if v else None)
print("qps_sweep", json.dumps(d.get("qps_sweep"))[:700])
PY
