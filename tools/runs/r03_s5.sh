#!/bin/bash
# round 3, call 5: parity of the tall / grouped tiled GEMM; its speed under prefill masks and for MoE; policy repeatability
OUT=gpurun_out/r03_s5; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "gemm_tall or dense_gemm or lm_head or logits or silu or stream_linear or moe or fused" 2>&1 | tail -8 | tee $OUT/pytest_new_ops.txt
{
KBENCH_MS=96,128,256 timeout 600 python tools/kbench.py gemm_tall
HSA_CU_MASK=0:128-255 KBENCH_NUM_CUS=128 KBENCH_MS=96,128,256 timeout 600 python tools/kbench.py gemm_tall
HSA_CU_MASK=0:160-255 KBENCH_NUM_CUS=96 KBENCH_MS=96,128,256 timeout 600 python tools/kbench.py gemm_tall
HSA_CU_MASK=0:0-127 KBENCH_NUM_CUS=128 KBENCH_MS=1024,2048 timeout 600 python tools/kbench.py gemm_tall
HSA_CU_MASK=0:0-159 KBENCH_NUM_CUS=160 KBENCH_MS=1024,2048 timeout 600 python tools/kbench.py gemm_tall
} 2>&1 | grep -v amdgpu.ids > $OUT/gemm_tall_v2.txt
cut -c1-150 $OUT/gemm_tall_v2.txt
{ echo "## tall (256-row blocks) from T = 4096"; timeout 600 python tools/kbench.py moe; echo "## 128-row blocks only"; SEMIPD_MOE_TALL_MIN_ROWS=999999999 timeout 600 python tools/kbench.py moe; echo "## tall from T = 1024"; SEMIPD_MOE_TALL_MIN_ROWS=4096 SEMIPD_MOE_TALL_MIN_ROWS_PER_EXPERT=64 timeout 600 python tools/kbench.py moe; } 2>&1 | grep -v amdgpu.ids | tee $OUT/moe_tall.txt | cut -c1-150
for pd in "50 50" "62 38" "50 50" "62 38"; do
  set -- $pd
  timeout 900 python bench.py --prefill-cu $1 --decode-cu $2 --no-saturation-wave --no-cpu-baseline --rate-sweep "" \
      --steps 2 --warmup 1 > $OUT/bench_p$1_d$2.json 2> $OUT/bench_p$1_d$2.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_p$1_d$2.json").read().strip().splitlines()[-1])
    print("P$1/D$2", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"))
except Exception as e:
    print("P$1/D$2 failed", e)
PY
done 2>&1 | tee $OUT/policy_sweep.txt
grep -A30 "library GEMM solutions timed" $OUT/bench_p62_d38.err | head -34 > $OUT/tuning_table_p62.txt; head -1 $OUT/tuning_table_p62.txt | cut -c1-150
