#!/bin/bash
# round 4, call 24: router GEMM as planes into the routing kernel: parity, goldens, engine tests, DeepSeek steps A / B
OUT=gpurun_out/r04_s24; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py tests/test_gpu_ops.py tests/test_gpu_torch_ops.py -q -x -k "topk or mla or moe" > $OUT/pytest_topk.txt 2>&1; echo "pytest topk rc=$?"
tail -4 $OUT/pytest_topk.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_rank_widths.py tests/test_gpu_deepseek.py tests/test_gpu_fp8_kv.py tests/test_gpu_full_width.py -q -x -k "deepseek or v3" > $OUT/pytest_v3.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_v3.txt | cut -c1-220
for m in 1 0; do
  SEMIPD_MOE_GATE_PLANES=$m timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | sed "s/^/gate_planes=$m /" | cut -c1-120
  SEMIPD_MOE_GATE_PLANES=$m timeout 300 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | sed "s/^/gate_planes=$m /" | cut -c1-120
done | tee $OUT/steps.txt
