#!/bin/bash
# round 3, call 4: tall-GEMM parity, lm_head through the streaming kernel, policy sweep with the heuristic-list tuner
OUT=gpurun_out/r03_s4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_tall or dense_gemm or lm_head or logits" 2>&1 | tail -6 | tee $OUT/pytest_new_ops.txt
for pd in "62 38" "56 44" "50 50" "69 38" "75 38" "62 44"; do
  set -- $pd
  timeout 900 python bench.py --prefill-cu $1 --decode-cu $2 --no-saturation-wave --no-cpu-baseline --rate-sweep "" \
      --steps 1 --warmup 1 > $OUT/bench_p$1_d$2.json 2> $OUT/bench_p$1_d$2.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_p$1_d$2.json").read().strip().splitlines()[-1])
    print("P$1/D$2", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"), d["roofline_extra"].get("decode_step_ms"))
except Exception as e:
    print("P$1/D$2 failed", e)
PY
  grep -A30 "library GEMM solutions timed" $OUT/bench_p$1_d$2.err | head -34 > $OUT/tuning_table_p$1.txt
done 2>&1 | tee $OUT/policy_sweep.txt
head -2 $OUT/tuning_table_p62.txt | cut -c1-200
