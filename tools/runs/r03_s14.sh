#!/bin/bash
# round 3, call 14: the round-aware split-KV rule: decode tests, default bench line
OUT=gpurun_out/r03_s14; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fp8_kv.py -q -m gpu -k "decode or engine or semi_pd or unified" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 1500 python bench.py --steps 3 --warmup 1 --no-static-split-wave > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["p50_ttft_ms"], d["p99_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"], d["roofline"]["frac"], d["roofline_extra"]["decode_attention"]["avg_launch_us"])
print("sat", d["saturation"]["output_tok_s"], d["saturation"]["p50_tbt_ms"]); print("sweep", [(s["request_rate"], s["output_tok_s"], s["p50_ttft_ms"], s["p50_tbt_ms"], s["p99_tbt_ms"]) for s in d["qps_sweep"]])
PY
