#!/bin/bash
# round 3, call 10: the grouped streaming kernel above 64 tokens (blocks of 64): where does it stop paying?  then config 3
OUT=gpurun_out/r03_s10; mkdir -p $OUT
{
for m in none 0:160-255 0:128-255; do
  for sd in 1 0; do
    echo "## SEMIPD_MOE_STREAM_DECODE=$sd HSA_CU_MASK=$m"
    if [ "$m" = none ]; then KBENCH_MOE_TS=16,48,64,96,128,192,256,384 SEMIPD_MOE_STREAM_MAX_TOKENS=512 SEMIPD_MOE_STREAM_DECODE=$sd timeout 300 python tools/kbench.py moe; else KBENCH_MOE_TS=16,48,64,96,128,192,256,384 SEMIPD_MOE_STREAM_MAX_TOKENS=512 SEMIPD_MOE_STREAM_DECODE=$sd HSA_CU_MASK=$m timeout 300 python tools/kbench.py moe; fi
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/moe_decode_stream_taller.txt | cut -c1-120
for pd in "62 38" "50 50"; do
  set -- $pd
  timeout 900 python bench.py --model deepseek-v2-lite --prefill-cu $1 --decode-cu $2 --no-cpu-baseline --rate-sweep "" --no-static-split-wave --steps 2 --warmup 1 > $OUT/bench_config3_p$1.json 2> $OUT/bench_config3_p$1.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_config3_p$1.json").read().strip().splitlines()[-1])
print("config3 P$1/D$2", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), d.get("saturation",{}).get("output_tok_s"), d["roofline_extra"].get("decode_step_ms"))
PY
done 2>&1 | tee $OUT/bench_config3.txt
