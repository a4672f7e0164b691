#!/bin/bash
# round 6, call 1: the GPU tests the round's first changes touch (masked extend NaN convention, dense-GEMM import read-back, the
# pacer's timed layer events, full-depth parity with the divergence gaps printed + the 27-layer DeepSeek oracle check), then the
# default bench line with the goodput grid of both engines, the token check and the prefill accounting (3 timed steps)
OUT=gpurun_out/r06_s1; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "mask or dense_gemm" > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -1 $OUT/pytest_ops.txt | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_cu_share.py -q -x > $OUT/pytest_cu_share.txt 2>&1; echo "cu_share rc=$?"; tail -1 $OUT/pytest_cu_share.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_full_depth.py -q -x -s > $OUT/pytest_full_depth.txt 2>&1; echo "full_depth rc=$?"; grep -h "token-for-token\|near-tie\|oracle\|passed\|failed" $OUT/pytest_full_depth.txt | cut -c1-260
T0=$(date +%s)
timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
for k in ("value", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "goodput_req_s", "goodput", "token_check"):
    print(k, json.dumps(d.get(k))[:600])
print("prefill_batch_ms", json.dumps(d["roofline_extra"]["prefill_batch_ms"])[:1500])
for key in ("qps_sweep", "qps_sweep_unified"):
    for r in d.get(key, []):
        print(key, {k: r[k] for k in ("request_rate", "num_requests", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms", "output_tok_s", "meets_slo_itl", "meets_slo_tpot")})
print("saturation", d.get("saturation", {}).get("output_tok_s"), "unified", (d.get("unified_same_load") or {}).get("saturation", {}).get("output_tok_s"))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
