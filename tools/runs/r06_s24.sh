#!/bin/bash
# round 6, call 24: SEMIPD_TALL_MARGIN 0.97 (the tiled GEMM where it beat the library's winner at start-up) against 1.5 (the tiled GEMM
# unless the library is 1.5x faster) at the 40 / 48 req/s points of the goodput grid and at saturation, alternating on one box
OUT=gpurun_out/r06_s24; mkdir -p $OUT
i=0
for v in 0.97 1.5 0.97 1.5; do
  i=$((i + 1))
  SEMIPD_TALL_MARGIN=$v timeout 500 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-side-configs --no-token-check \
      --rate-sweep 40,48 --steps 1 --warmup 1 > $OUT/run${i}_$v.json 2> $OUT/run${i}_$v.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/run${i}_$v.json") if l.startswith("{")][-1])
print("margin $v: 32 req/s TTFT %.1f/%.1f TBT %.2f/%.2f sat %.0f" % (d["p50_ttft_ms"], d["p99_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"], d["saturation"]["output_tok_s"]))
for r in d["qps_sweep"]:
    if r["request_rate"] >= 40:
        print("   ", {k: r[k] for k in ("request_rate", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms", "meets_slo_itl", "meets_slo_tpot")})
PY
done
