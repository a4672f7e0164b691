#!/bin/bash
# round 6, call 2: disjoint work-conserving shares next to the default (P88 / D100 + deadline): the decode instance on its OWN
# CUs while the prefill instance is busy (whole chip otherwise).  Shares are whole groups of 32 CUs: 75 / 25 = 192 / 64,
# 62 / 38 = 160 / 96, 75 / 38 = 192 / 96 (32 shared).  Headline wave (1 warm-up + 3 timed), the 40 and 48 req/s points, saturation.
OUT=gpurun_out/r06_s2; mkdir -p $OUT
for pol in "75 25" "62 38" "75 38" "88 100"; do
  set -- $pol
  T0=$(date +%s)
  timeout 500 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-side-configs --rate-sweep "32,40,48" \
      --prefill-cu $1 --decode-cu $2 --steps 3 --warmup 1 > $OUT/p$1_d$2.json 2> $OUT/p$1_d$2.err
  echo "P$1/D$2 rc=$? in $(( $(date +%s) - T0 )) s"
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/p$1_d$2.json") if l.startswith("{")][-1])
pb = d["roofline_extra"]["prefill_batch_ms"]
print("  headline", {k: round(d[k], 2) for k in ("p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms")}, "sat", d["saturation"]["output_tok_s"], "goodput", d["goodput"]["semi_pd"])
print("  prefill", {k: pb.get(k) for k in ("avg_tokens", "gpu_owned", "layers_without_hold", "held")}, "holds", pb.get("step_gate", {}).get("holds"), "decode", d["roofline_extra"]["decode_step_ms"], "frac", d["roofline"]["frac"])
for r in d["qps_sweep"]:
    print("  ", {k: r[k] for k in ("request_rate", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms", "meets_slo_itl")})
PY
done
