#!/bin/bash
# round 5, call 6: the deadline gate's kernels and engine test, shader-engine balance of the shares (probe at 40 / 56 / 64 / 208 / 224 CUs),
# and the policy sweep: prefill share 192 / 224 / 256 CUs x decode-step deadline off / 6 / 8 / 10 ms
OUT=gpurun_out/r05_s6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_step_clock.py tests/test_gpu_cu_share.py -q -x -s > $OUT/pytest_gate.txt 2>&1; echo "pytest gate rc=$?"
grep -E "passed|failed|^E  |step gate" $OUT/pytest_gate.txt | head -12 | cut -c1-220
for r in "0 39" "0 55" "0 63" "0 191" "0 207" "0 223"; do
  timeout 120 tools/hbm_cu_probe range $r 2>&1 | grep -E "^# CUs|lds-nt   \| 8|anat"
done | tee $OUT/hbm_probe_se_balance.txt
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave --prefill-cu 88
run p75_off --prefill-cu 75
run p88_off --prefill-cu 88
run p88_d6 --prefill-cu 88 --decode-step-deadline-ms 6
run p88_d8 --prefill-cu 88 --decode-step-deadline-ms 8
run p88_d10 --prefill-cu 88 --decode-step-deadline-ms 10
run p100_d6 --prefill-cu 100 --decode-step-deadline-ms 6
run p100_d8 --prefill-cu 100 --decode-step-deadline-ms 8
run p75_d8 --prefill-cu 75 --decode-step-deadline-ms 8
python tools/summarize_runs.py $OUT/p*.json | tee $OUT/summary.txt
