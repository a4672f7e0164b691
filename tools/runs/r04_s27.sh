#!/bin/bash
# round 4, call 27: the W_kc absorption inside the MLA prep launch: parity, engine tests, V2-Lite step A / B
OUT=gpurun_out/r04_s27; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py tests/test_gpu_ops.py -q -x -k "mla or bmm" > $OUT/pytest_mla.txt 2>&1; echo "pytest mla rc=$?"
tail -4 $OUT/pytest_mla.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_deepseek.py tests/test_gpu_fp8_kv.py tests/test_gpu_full_width.py -q -x -k "deepseek" > $OUT/pytest_engines.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_engines.txt | cut -c1-220
for m in 1 0; do
  SEMIPD_MLA_ABSORB_IN_PREP=$m timeout 300 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | sed "s/^/absorb_in_prep=$m /" | cut -c1-120
done | tee $OUT/steps.txt
