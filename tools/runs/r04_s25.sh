#!/bin/bash
# round 4, call 25: the shared experts' down_proj planes summed by the moe_sum launch (route first): parity, engine tests, step A / B
OUT=gpurun_out/r04_s25; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py tests/test_gpu_ops.py -q -x -k "moe_sum or mla or topk" > $OUT/pytest_ops.txt 2>&1; echo "pytest ops rc=$?"
tail -4 $OUT/pytest_ops.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_deepseek.py tests/test_gpu_fp8_kv.py tests/test_gpu_full_width.py tests/test_gpu_rank_widths.py -q -x -k "deepseek or v3" > $OUT/pytest_engines.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_engines.txt | cut -c1-220
for m in 1 0; do
  SEMIPD_MOE_SHARED_PLANES=$m timeout 300 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | sed "s/^/shared_planes=$m /" | cut -c1-120
done | tee $OUT/steps.txt
