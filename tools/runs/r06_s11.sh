#!/bin/bash
# round 6, call 11: the whole GPU suite on the tree with the fused decode launch and the wide streaming GEMM, then the default bench line
OUT=gpurun_out/r06_s11; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "gpu suite rc=$? in $(( $(date +%s) - T0 )) s"; tail -3 $OUT/pytest_gpu.txt | cut -c1-300
T0=$(date +%s)
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
for k in ("value", "steps", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "goodput_req_s", "goodput", "token_check"):
    print(k, json.dumps(d.get(k))[:600])
print("prefill_batch_ms", json.dumps(d["roofline_extra"]["prefill_batch_ms"])[:1200])
for key in ("qps_sweep", "qps_sweep_unified"):
    for r in d.get(key, []):
        print(key, {k: r[k] for k in ("request_rate", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms", "output_tok_s", "meets_slo_itl", "meets_slo_tpot")})
print("saturation", d.get("saturation", {}).get("output_tok_s"), "unified", (d.get("unified_same_load") or {}).get("saturation", {}).get("output_tok_s"))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "decode_attention", json.dumps(d["roofline_extra"].get("decode_attention"))[:400])
print("config3", json.dumps(d.get("config3_deepseek_v2_lite"))[:400])
PY
