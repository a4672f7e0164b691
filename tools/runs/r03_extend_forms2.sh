#!/bin/bash
# round 3: share-aware choice of the extend kernel's form: tests, then the default line without the extra waves
OUT=gpurun_out/r03_extend_forms; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "extend" -x > $OUT/pytest_extend.txt 2>&1; tail -2 $OUT/pytest_extend.txt
timeout 900 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('line', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'], d['roofline_extra'].get('extend_attention'))"
