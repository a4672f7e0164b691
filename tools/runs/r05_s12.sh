#!/bin/bash
# round 5, call 12: is the collapse with the deadline gate a feedback through the backlog rule (prefill takes every CU from 8192 waiting
# tokens -> decode steps of 15-20 ms -> every gate holds)?  the gate with the backlog rule off, and at half the load
OUT=gpurun_out/r05_s12; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs --no-saturation-wave "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --prefill-cu 88
run p88_d8_nobacklog --prefill-cu 88 --decode-step-deadline-ms 8 --prefill-backlog-full-tokens 0
run p88_d8_rate16 --prefill-cu 88 --decode-step-deadline-ms 8 --request-rate 16
run p88_d12_nobacklog --prefill-cu 88 --decode-step-deadline-ms 12 --prefill-backlog-full-tokens 0
python tools/summarize_runs.py $OUT/p*.json | tee $OUT/summary.txt
