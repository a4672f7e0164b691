#!/bin/bash
# round 3: per-rank shapes of DeepSeek-V3 block-fp8 TP = 8 (all 61 layers, widths / 8) on one GPU
OUT=gpurun_out/r03_v3rank; mkdir -p $OUT
timeout 1700 python bench.py --model deepseek-v3-tp8-rank --quantization fp8 --num-requests 48 --request-rate 4 --output-len 64 --no-cpu-baseline --rate-sweep "" --no-static-split-wave --steps 1 --warmup 1 --max-running-requests 64 --mem-fraction-static 0.6 > $OUT/bench_v3_rank.json 2> $OUT/bench_v3_rank.err
echo "rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_v3_rank.json").read().strip().splitlines()[-1])
    print("v3-tp8-rank", d["config"]["workload"][:120]); print(d["value"], d["p50_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"], d.get("saturation",{}).get("output_tok_s"), d.get("saturation",{}).get("p50_tbt_ms"), d["roofline_extra"].get("decode_step_ms"), d["roofline_extra"].get("prefill_batch_ms"))
except Exception as e:
    print("failed", e)
PY
grep -v "amdgpu.ids\|library GEMM\|^dtype=" $OUT/bench_v3_rank.err | tail -6 | cut -c1-300
