#!/bin/bash
# round 3, call 6: decode attention with K through LDS (A/B under decode masks), parity, then the serving effect
OUT=gpurun_out/r03_s6; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py -q -m gpu -k "decode" 2>&1 | tail -4 | tee $OUT/pytest_decode.txt
{
for m in none 0:160-255 0:128-255; do
  for kl in 1 0; do
    echo "## SEMIPD_DECODE_K_LDS=$kl"
    if [ "$m" = none ]; then SEMIPD_DECODE_K_LDS=$kl timeout 300 python tools/kbench.py decode_small; else SEMIPD_DECODE_K_LDS=$kl HSA_CU_MASK=$m timeout 300 python tools/kbench.py decode_small; fi
  done
done
} 2>&1 | grep -v amdgpu.ids > $OUT/decode_k_through_lds.txt
cut -c1-230 $OUT/decode_k_through_lds.txt
SEMIPD_DECODE_K_LDS=1 timeout 600 python tools/kbench.py decode 2>&1 | grep -v amdgpu.ids | grep "Hq=32" > $OUT/decode_big_kl1.txt
SEMIPD_DECODE_K_LDS=0 timeout 600 python tools/kbench.py decode 2>&1 | grep -v amdgpu.ids | grep "Hq=32" > $OUT/decode_big_kl0.txt
paste -d'\n' $OUT/decode_big_kl1.txt $OUT/decode_big_kl0.txt | cut -c1-120
for kl in 1 0; do
  SEMIPD_DECODE_K_LDS=$kl timeout 900 python bench.py --prefill-cu 50 --decode-cu 50 --no-saturation-wave --no-cpu-baseline --rate-sweep "" --steps 2 --warmup 1 > $OUT/bench_p50_d50_kl$kl.json 2> $OUT/bench_p50_d50_kl$kl.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_p50_d50_kl$kl.json").read().strip().splitlines()[-1])
print("P50/D50 K_LDS=$kl", d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("decode_attention",{}).get("avg_launch_us"))
PY
done 2>&1 | tee $OUT/bench_ab.txt
