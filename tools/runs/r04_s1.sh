#!/bin/bash
# round 4, call 1: nested CU policies re-measured with the share-tuned prefill GEMMs (existing HSA_CU_MASK mechanism)
OUT=gpurun_out/r04_s1; mkdir -p $OUT
for pol in "62 38" "62 100" "75 100" "88 100" "100 100"; do
  set -- $pol
  timeout 600 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline \
     --prefill-cu $1 --decode-cu $2 > $OUT/p$1_d$2.json 2> $OUT/p$1_d$2.err
  echo "P$1/D$2 rc=$?"
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), s.get("p50_tbt_ms"), "frac", (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"), d["roofline_extra"].get("decode_step_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
