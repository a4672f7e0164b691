#!/bin/bash
# round 5, call 15: the deadline following a 12 ms objective for the 99th percentile of the token gaps (start 9 ms), run-ahead 1 / 2, four timed waves
OUT=gpurun_out/r05_s15; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python bench.py --steps 4 --warmup 1 --rate-sweep "" --no-static-split-wave --no-unified-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave --prefill-cu 88 --steps 1
SEMIPD_PACER_RUN_AHEAD=1 run p88_ra1_d9_slo12 --prefill-cu 88 --decode-step-deadline-ms 9 --decode-tbt-slo-ms 12
SEMIPD_PACER_RUN_AHEAD=2 run p88_ra2_d9_slo12 --prefill-cu 88 --decode-step-deadline-ms 9 --decode-tbt-slo-ms 12
SEMIPD_PACER_RUN_AHEAD=1 run p88_ra1_d8p5 --prefill-cu 88 --decode-step-deadline-ms 8.5
SEMIPD_PACER_RUN_AHEAD=2 run p88_ra2_d8_slo11p5 --prefill-cu 88 --decode-step-deadline-ms 8 --decode-tbt-slo-ms 11.5
python tools/summarize_runs.py $OUT/p*.json | tee $OUT/summary.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/p*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["roofline_extra"]["prefill_batch_ms"].get("step_gate"))
PY
