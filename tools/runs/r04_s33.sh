#!/bin/bash
# round 4, call 33: the tiled GEMM's XCD-contiguous tile order (row tiles of one W tile next to each other on one XCD) for layer-sized weights in situ
OUT=gpurun_out/r04_s33; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave
SEMIPD_G8_XCD_ORDER=1 run order1
SEMIPD_G8_XCD_ORDER=0 run order0
grep -h "rows=1024\|rows=1536\|rows=2048" $OUT/order1.err | grep silu | cut -c1-140
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/order*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); pb = d["roofline_extra"].get("prefill_batch_ms") or {}
    print(f.split("/")[-1], "TTFT", round(d["p50_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), "P", pb.get("forward_and_sync"))
PY
