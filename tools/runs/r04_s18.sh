#!/bin/bash
# round 4, call 18: merged q_a | kv_a block-fp8 GEMM (DeepSeek-V3 rank shapes): parity, step time A / B; then the default bench line
OUT=gpurun_out/r04_s18; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_rank_widths.py tests/test_gpu_deepseek.py -q -x -k "deepseek or v3" > $OUT/pytest_v3.txt 2>&1; echo "pytest v3 rc=$?"
tail -3 $OUT/pytest_v3.txt | cut -c1-220
for m in 1 0; do
  SEMIPD_MLA_MERGED_QKV_A=$m timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | sed "s/^/merged=$m /" | cut -c1-120
done | tee $OUT/steps_v3.txt
SECONDS=0
timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
s = d.get("saturation") or {}
print("default", d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), "sat", s.get("output_tok_s"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
for k in ("config1_opt_125m", "config3_deepseek_v2_lite", "static_split_50_50"):
    v = d.get(k) or d.get("config", {}).get(k)
    print(k, json.dumps(v)[:600] if v else None)
print("cpu_baseline", d.get("cpu_baseline"))
print(sorted(d.keys()))
PY
