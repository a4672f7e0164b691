#!/bin/bash
# round 4, call 31: the prefill share between 80 % and 100 % with this tree (tiled GEMM where it wins, shorter prefill batches): p50s against the TBT tail
OUT=gpurun_out/r04_s31; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
for p in 84 88 91 94 80 88; do run dyn_p${p}_d100_$RANDOM --prefill-cu $p; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/dyn_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        e = d["roofline_extra"]; pb = e.get("prefill_batch_ms") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), s.get("p50_tbt_ms"), "frac", (d.get("roofline") or {}).get("frac"), "P", pb.get("batches"), pb.get("avg_tokens"), pb.get("forward_and_sync"))
    except Exception as e:
        print(f, "failed", e)
PY
