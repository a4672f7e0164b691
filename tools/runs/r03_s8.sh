#!/bin/bash
# round 3, call 8: grouped LDS-DMA streaming kernel for MoE decode: parity, speed on shares, config 3 under the default masks
OUT=gpurun_out/r03_s8; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_deepseek.py -q -m gpu -k "moe or fused or deepseek" 2>&1 | tail -6 | tee $OUT/pytest_moe.txt
{
for m in none 0:160-255 0:128-255; do
  for sd in 1 0; do
    echo "## SEMIPD_MOE_STREAM_DECODE=$sd HSA_CU_MASK=$m"
    if [ "$m" = none ]; then SEMIPD_MOE_STREAM_DECODE=$sd timeout 300 python tools/kbench.py moe | head -5; else SEMIPD_MOE_STREAM_DECODE=$sd HSA_CU_MASK=$m timeout 300 python tools/kbench.py moe | head -5; fi
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/moe_decode_stream.txt | cut -c1-150
for sd in 1 0; do
  SEMIPD_MOE_STREAM_DECODE=$sd timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline --rate-sweep "" --no-static-split-wave --steps 2 --warmup 1 > $OUT/bench_config3_sd$sd.json 2> $OUT/bench_config3_sd$sd.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_config3_sd$sd.json").read().strip().splitlines()[-1])
print("config3 P62/D38 stream_decode=$sd", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), d.get("saturation",{}).get("output_tok_s"))
PY
done 2>&1 | tee $OUT/bench_config3.txt
