#!/bin/bash
# Round 3, first GPU call: where does a masked policy stand with the round-2 kernels?
#  (1) library GEMMs at the serving shapes (M = 1024 / 2048) under prefill masks, plain / TunableOp / fixed stream-K grid
#  (2) the streaming GEMM on small decode shares
#  (3) bench.py under asymmetric disjoint and nested masks
OUT=gpurun_out/r03_s1
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
run() { echo "## $*"; "$@"; }

{
for mask in none 0:0-191 0:0-207 0:0-223; do
  if [ "$mask" = none ]; then M=(env -u HSA_CU_MASK); else M=(env HSA_CU_MASK=$mask); fi
  KBENCH_MS=1024,2048 "${M[@]}" timeout 300 python tools/kbench.py linear_prefill
  KBENCH_MS=1024,2048 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0 \
      PYTORCH_TUNABLEOP_FILENAME=$OUT/tunable_${mask//[:]/_}.csv "${M[@]}" timeout 600 python tools/kbench.py linear_prefill | sed 's/^/tunableop /'
done
KBENCH_MS=1024,2048 TENSILE_STREAMK_FIXED_GRID=192 HSA_CU_MASK=0:0-191 timeout 300 python tools/kbench.py linear_prefill
KBENCH_MS=1024,2048 TENSILE_STREAMK_FIXED_GRID=208 HSA_CU_MASK=0:0-207 timeout 300 python tools/kbench.py linear_prefill
} > $OUT/library_gemm_under_masks.txt 2>&1

{
HSA_CU_MASK=0:192-255 KBENCH_NUM_CUS=64 KBENCH_MS=16,32,64 timeout 600 python tools/kbench.py stream_linear
HSA_CU_MASK=0:160-255 KBENCH_NUM_CUS=96 KBENCH_MS=16,32,64 timeout 600 python tools/kbench.py stream_linear
HSA_CU_MASK=0:208-255 KBENCH_NUM_CUS=48 KBENCH_MS=16,32 timeout 600 python tools/kbench.py stream_linear
} > $OUT/stream_linear_small_shares.txt 2>&1

for pd in "75 25" "62 38" "80 100" "80 20" "88 12" "70 30"; do
  set -- $pd
  timeout 600 python bench.py --prefill-cu $1 --decode-cu $2 --no-static-split-wave --no-saturation-wave --no-cpu-baseline \
      --steps 1 --warmup 1 > $OUT/bench_p$1_d$2.json 2> $OUT/bench_p$1_d$2.err
  tail -c 600 $OUT/bench_p$1_d$2.err | tail -3
  python - <<EOF
import json
try:
    d = json.loads(open("$OUT/bench_p$1_d$2.json").read().strip().splitlines()[-1])
    print("P$1/D$2", d["value"], d["p50_ttft_ms"], d["p99_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"], (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("P$1/D$2 failed", e)
EOF
done 2>&1 | tee $OUT/policy_sweep.txt
