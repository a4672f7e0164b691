#!/bin/bash
# round 4, call 22: mla_decode_prep_rows (the q_lora / block-fp8 decode path): parity, engine tests, DeepSeek-V3 rank step A / B
OUT=gpurun_out/r04_s22; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py -q > $OUT/pytest_mla_prep.txt 2>&1; echo "pytest mla prep rc=$?"
tail -4 $OUT/pytest_mla_prep.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_rank_widths.py tests/test_gpu_deepseek.py tests/test_gpu_fp8_kv.py tests/test_gpu_full_width.py -q -x -k "deepseek or v3" > $OUT/pytest_v3.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_v3.txt | cut -c1-220
for m in 1 0; do
  SEMIPD_MLA_PREP_ROWS=$m timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | sed "s/^/prep_rows=$m /" | cut -c1-120
done | tee $OUT/steps_v3.txt
