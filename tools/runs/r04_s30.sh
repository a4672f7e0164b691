#!/bin/bash
# round 4, call 30: the tiled GEMM on a 224-CU prefill share (P88): 1024 rows x gate_up = 224 tiles of 256 x 256 = ONE round there; serving with / without
OUT=gpurun_out/r04_s30; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave --prefill-cu 88
SEMIPD_TALL_PREFILL=1 run p88_tall_on --prefill-cu 88
SEMIPD_TALL_PREFILL=0 run p88_tall_off --prefill-cu 88
SEMIPD_TALL_PREFILL=1 run p80_tall_on
grep -h "rows=" $OUT/p88_tall_on.err | grep "silu\|tiled" | cut -c1-140 | head -40 > $OUT/tall_table_p88.txt; cat $OUT/tall_table_p88.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/p8*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        e = d["roofline_extra"]; pb = e.get("prefill_batch_ms") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), "frac", (d.get("roofline") or {}).get("frac"), "P", pb.get("batches"), pb.get("avg_tokens"), pb.get("forward_and_sync"))
    except Exception as e:
        print(f, "failed", e)
PY
