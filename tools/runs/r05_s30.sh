#!/bin/bash
# round 5, call 30: A / B of the pacer's waits servicing the previous batch's result (two runs each, same box), then SURVEY 8d.2's sweep for in = 128 / out = 64
OUT=gpurun_out/r05_s30; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 3 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs --no-saturation-wave "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --steps 1
SEMIPD_PACER_NO_WAIT_HOOK=1 run ab_off_1
run ab_on_1
SEMIPD_PACER_NO_WAIT_HOOK=1 run ab_off_2
run ab_on_2
python tools/summarize_runs.py $OUT/ab_*.json | tee $OUT/ab_summary.txt
bash tools/runs/r05_s19.sh b 2>&1 | tail -14
