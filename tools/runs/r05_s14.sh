#!/bin/bash
# round 5, call 14: step pacer -- run-ahead 1 / 2 layers x deadline 7 / 8 / 9 ms on the 224-CU prefill share; holds off under the backlog rule
OUT=gpurun_out/r05_s14; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave --prefill-cu 88
SEMIPD_PACER_RUN_AHEAD=1 run p88_ra1_d7 --prefill-cu 88 --decode-step-deadline-ms 7
SEMIPD_PACER_RUN_AHEAD=1 run p88_ra1_d8 --prefill-cu 88 --decode-step-deadline-ms 8
SEMIPD_PACER_RUN_AHEAD=1 run p88_ra1_d9 --prefill-cu 88 --decode-step-deadline-ms 9
SEMIPD_PACER_RUN_AHEAD=2 run p88_ra2_d9 --prefill-cu 88 --decode-step-deadline-ms 9
SEMIPD_PACER_RUN_AHEAD=2 run p88_ra2_d8 --prefill-cu 88 --decode-step-deadline-ms 8
SEMIPD_PACER_RUN_AHEAD=1 run p88_ra1_d8_again --prefill-cu 88 --decode-step-deadline-ms 8
python tools/summarize_runs.py $OUT/p*.json | tee $OUT/summary.txt
