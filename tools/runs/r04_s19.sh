#!/bin/bash
# round 4, call 19: merged q_a | kv_a block-fp8 GEMM (DeepSeek-V3 rank shapes): parity, step time A / B
OUT=gpurun_out/r04_s19; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_rank_widths.py tests/test_gpu_deepseek.py tests/test_gpu_fp8_kv.py -q -x -k "deepseek or v3" > $OUT/pytest_v3.txt 2>&1; echo "pytest v3 rc=$?"
tail -3 $OUT/pytest_v3.txt | cut -c1-220
for m in 1 0; do
  SEMIPD_MLA_MERGED_QKV_A=$m timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | sed "s/^/merged=$m /" | cut -c1-120
done | tee $OUT/steps_v3.txt
