#!/bin/bash
# round 3: new MoE thresholds: tests, kbench sweep with the defaults, config 3 line
OUT=gpurun_out/r03_moe_mid; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "moe or fused_experts" -x > $OUT/pytest_moe.txt 2>&1; tail -3 $OUT/pytest_moe.txt
timeout 1500 python -m pytest tests/test_gpu_deepseek.py tests/test_gpu_engine.py -q -m gpu -x > $OUT/pytest_engine.txt 2>&1; tail -3 $OUT/pytest_engine.txt
export KBENCH_MOE_TS=384,512,768,1024,1280,1536,2048,3072,4096,8192
for mask in "0:0-127" ""; do
  if [ -n "$mask" ]; then export HSA_CU_MASK=$mask; export KBENCH_NUM_CUS=128; else unset HSA_CU_MASK; unset KBENCH_NUM_CUS; fi
  echo "# HSA_CU_MASK=$mask defaults"; timeout 600 python tools/kbench.py moe 2>&1 | grep "^moe"
done | tee $OUT/kbench_moe_defaults.txt
unset HSA_CU_MASK; unset KBENCH_NUM_CUS
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline --no-saturation-wave --rate-sweep "" --steps 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python -c "
import json; d=json.loads(open('$OUT/bench_c3.json').read().strip().splitlines()[-1]); print('dsv2lite', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'])"
