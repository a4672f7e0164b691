#!/bin/bash
# round 4, call 20: moe_align's scan by a wave instead of one thread: golden / parity tests, DeepSeek decode steps
OUT=gpurun_out/r04_s20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_torch_ops.py -q -x -k "moe or align" > $OUT/pytest_moe.txt 2>&1; echo "pytest moe rc=$?"
tail -3 $OUT/pytest_moe.txt | cut -c1-220
timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | cut -c1-120 | tee $OUT/steps.txt
timeout 300 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | cut -c1-120 | tee -a $OUT/steps.txt
