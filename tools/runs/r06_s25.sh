#!/bin/bash
# round 6, call 25: tiled GEMM preferred for qkv / o / down only (SEMIPD_TALL_MARGIN=1.5 SEMIPD_TALL_MARGIN_WIDE=0.97: gate_up keeps the
# library's stream-K kernel unless the tiled one beats it) against both uniform margins, headline + 40 / 48 req/s + saturation
OUT=gpurun_out/r06_s25; mkdir -p $OUT
i=0
for v in "0.97 0.97" "1.5 0.97" "1.5 1.5" "0.97 0.97" "1.5 0.97" "1.5 1.5"; do
  set -- $v; i=$((i + 1))
  SEMIPD_TALL_MARGIN=$1 SEMIPD_TALL_MARGIN_WIDE=$2 timeout 500 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-side-configs --no-token-check \
      --rate-sweep 40,48 --steps 2 --warmup 1 > $OUT/run${i}.json 2> $OUT/run${i}.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/run${i}.json") if l.startswith("{")][-1])
pb = d["roofline_extra"]["prefill_batch_ms"]
print("margins $1 / $2: 32 req/s TTFT %.1f/%.1f TBT %.2f/%.2f sat %.0f  P batch %.1f (layers %.1f held %.2f) gemm %.0f TF/s holds %s" % (d["p50_ttft_ms"], d["p99_ttft_ms"], d["p50_tbt_ms"], d["p99_tbt_ms"], d["saturation"]["output_tok_s"], pb["forward_and_sync"], pb["layers_without_hold"], pb["held_gpu_idle"], d["roofline_extra"]["prefill_gemm"]["achieved"], pb["step_gate"]["holds"]))
for r in d["qps_sweep"]:
    if r["request_rate"] >= 40:
        print("   ", {k: r[k] for k in ("request_rate", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms", "meets_slo_itl", "meets_slo_tpot")})
PY
done
