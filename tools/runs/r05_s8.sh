#!/bin/bash
# round 5, call 8: the deadline gate with RELAXED polling loads: prefill share 224 / 192 CUs x deadline 6 / 8 / 10 ms
OUT=gpurun_out/r05_s8; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step_clock.py -q -x > $OUT/pytest_gate.txt 2>&1; echo "pytest gate rc=$?"
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave --prefill-cu 88
run p88_d6 --prefill-cu 88 --decode-step-deadline-ms 6
run p88_d8 --prefill-cu 88 --decode-step-deadline-ms 8
run p88_d10 --prefill-cu 88 --decode-step-deadline-ms 10
run p88_off --prefill-cu 88
run p75_d8 --prefill-cu 75 --decode-step-deadline-ms 8
run p100_d8 --prefill-cu 100 --decode-step-deadline-ms 8
python tools/summarize_runs.py $OUT/p*.json | tee $OUT/summary.txt
