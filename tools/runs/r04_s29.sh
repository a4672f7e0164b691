#!/bin/bash
# round 4, call 29: the tiled GEMM timed against the library's winners at start-up and taken where it wins (prefill-sized batches on the share): serving A / B, engine tests
OUT=gpurun_out/r04_s29; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave
SEMIPD_TALL_PREFILL=1 run tall_on
SEMIPD_TALL_PREFILL=0 run tall_off
SEMIPD_TALL_PREFILL=1 run tall_on_2
SEMIPD_TALL_PREFILL=0 run tall_off_2
grep -h "rows=" $OUT/tall_on.err | cut -c1-140 | head -60 > $OUT/tall_table.txt; cat $OUT/tall_table.txt | head -40
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/tall_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        e = d["roofline_extra"]; pb = e.get("prefill_batch_ms") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), "frac", (d.get("roofline") or {}).get("frac"), "P", pb.get("batches"), pb.get("avg_tokens"), pb.get("forward_and_sync"))
    except Exception as e:
        print(f, "failed", e)
PY
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_width.py tests/test_gpu_cu_share.py -q -x > $OUT/pytest_engines.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_engines.txt | cut -c1-220
