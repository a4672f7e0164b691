#!/bin/bash
# round 3, call 3: new tests; streaming GEMM with the 4-deep ring and the share-aware K split on small shares; the tall
# GEMM; policy sweep with library solutions timed on the prefill share
OUT=gpurun_out/r03_s3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_tall or dense_gemm or stream_linear or planes" 2>&1 | tail -5 | tee $OUT/pytest_new_ops.txt
for m in 0:192-255/64 0:160-255/96 0:128-255/128; do
  mask=${m%/*}; n=${m#*/}
  for ring in 4 3; do
    echo "## ring=$ring"; SEMIPD_SL_RING=$ring HSA_CU_MASK=$mask KBENCH_NUM_CUS=$n KBENCH_MS=16,32 KBENCH_SL_SWEEP=0 timeout 300 python tools/kbench.py stream_linear
  done
done 2>&1 | grep -v amdgpu.ids > $OUT/stream_linear_shares_ring4.txt
cat $OUT/stream_linear_shares_ring4.txt | cut -c1-150
KBENCH_MS=96,128,192,256,1024,4096 timeout 900 python tools/kbench.py gemm_tall 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_tall.txt | cut -c1-150
timeout 1500 python -m pytest tests/test_gpu_full_width.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_engine.txt
for pd in "62 38" "70 30" "50 50" "75 25"; do
  set -- $pd
  timeout 900 python bench.py --prefill-cu $1 --decode-cu $2 --no-saturation-wave --no-cpu-baseline --rate-sweep "" \
      --steps 1 --warmup 1 > $OUT/bench_p$1_d$2.json 2> $OUT/bench_p$1_d$2.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_p$1_d$2.json").read().strip().splitlines()[-1])
    print("P$1/D$2 tuned", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"))
except Exception as e:
    print("P$1/D$2 failed", e)
PY
  grep -A30 "library GEMM solutions timed" $OUT/bench_p$1_d$2.err | head -34 > $OUT/tuning_table_p$1.txt
done 2>&1 | tee $OUT/policy_sweep.txt
timeout 900 python bench.py --prefill-cu 62 --decode-cu 38 --no-prefill-gemm-tuning --no-saturation-wave --no-cpu-baseline --rate-sweep "" --steps 1 --warmup 1 > $OUT/bench_p62_d38_untuned.json 2> $OUT/bench_p62_d38_untuned.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_p62_d38_untuned.json").read().strip().splitlines()[-1])
print("P62/D38 untuned", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"))
PY
