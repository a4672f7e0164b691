#!/bin/bash
# round 5, call 40: extend attention under a custom mask (golden, random tree masks on every kernel route, properties) and every other
# extend-attention test (the kernel file changed: unmasked instantiations must be what they were)
OUT=gpurun_out/r05_s40; mkdir -p $OUT
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py -q -k "extend" --durations=5 > $OUT/pytest_extend.txt 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -12 $OUT/pytest_extend.txt | cut -c1-220
timeout 300 python tools/kbench.py extend 2>&1 | tail -12 | tee $OUT/kbench_extend.txt
