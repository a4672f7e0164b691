#!/bin/bash
# round 3: DeepSeek decode layer with fewer launches (moe_sum + scale + shared in one, planes into the norm, batched GEMMs
# written in place): the op test, the DeepSeek tests, config 3 line
OUT=gpurun_out/r03_c3launch; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "moe_sum_scale_add or fused_moe or fused_experts" -x > $OUT/pytest_ops.txt 2>&1; tail -3 $OUT/pytest_ops.txt
timeout 1500 python -m pytest tests/test_gpu_deepseek.py -q -m gpu -x > $OUT/pytest_deepseek.txt 2>&1; tail -3 $OUT/pytest_deepseek.txt
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline --no-saturation-wave --rate-sweep "" > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python -c "
import json; d=json.loads(open('$OUT/bench_c3.json').read().strip().splitlines()[-1]); print('dsv2lite', d['value'], d['p50_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'], d.get('static_split_50_50'))"
