#!/bin/bash
# round 4, call 34: how eagerly should the prefill instance take the tiled GEMM?  (it is the friendlier neighbour: TBT p50 fell 10 % when down_proj moved to it)
OUT=gpurun_out/r04_s34; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave
SEMIPD_TALL_MARGIN=0.97 run margin_097
SEMIPD_TALL_MARGIN=1.08 run margin_108
SEMIPD_TALL_MARGIN=1.25 run margin_125
SEMIPD_TALL_MARGIN=0.97 run margin_097_2
SEMIPD_TALL_MARGIN=1.08 run margin_108_2
for m in 108 125; do echo "== margin $m"; grep -h "<- tiled" $OUT/margin_$m.err | cut -c1-120; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/margin*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); pb = d["roofline_extra"].get("prefill_batch_ms") or {}; s = d.get("saturation") or {}
    print(f.split("/")[-1], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), "sat", s.get("output_tok_s"), "frac", d["roofline"]["frac"], "P", pb.get("forward_and_sync"))
PY
