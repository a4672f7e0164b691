#!/bin/bash
# round 5, call 13: the decode-step deadline on the HOST (semi_pd/step_pacer.py): engine test, then prefill share 224 / 192 CUs x deadline
OUT=gpurun_out/r05_s13; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_cu_share.py -q -x -s > $OUT/pytest_share.txt 2>&1; echo "pytest share rc=$?"
grep -E "passed|failed|^E  |step pacer" $OUT/pytest_share.txt | head -8 | cut -c1-220
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave --prefill-cu 88
run p88_d6 --prefill-cu 88 --decode-step-deadline-ms 6
run p88_d8 --prefill-cu 88 --decode-step-deadline-ms 8
run p88_d10 --prefill-cu 88 --decode-step-deadline-ms 10
run p88_d1000 --prefill-cu 88 --decode-step-deadline-ms 1000
run p100_d8 --prefill-cu 100 --decode-step-deadline-ms 8
run p75_d8 --prefill-cu 75 --decode-step-deadline-ms 8
python tools/summarize_runs.py $OUT/p*.json | tee $OUT/summary.txt
