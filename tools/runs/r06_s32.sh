#!/bin/bash
# round 6, call 32: the final tree (four-wave workgroups up to 64 rows, one-group wide form): smoke, the whole GPU suite, the default bench line
OUT=gpurun_out/r06_s32; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/smoke.txt
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "gpu suite rc=$? in $(( $(date +%s) - T0 )) s"; tail -3 $OUT/pytest_gpu.txt | cut -c1-300
T0=$(date +%s)
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
for k in ("value", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "goodput_req_s", "goodput"):
    print(k, d.get(k))
for e in d["token_check"]["engines"]:
    print(e["engine"], "equal", e["equal_requests"], "near", [(x["rank_in_reference"], x["logprob_gap"]) for x in e["near_tie_divergences"]], "errors", e["errors"])
for r in d["qps_sweep"]:
    print({k: r[k] for k in ("request_rate", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "p99_tpot_ms", "meets_slo_itl", "meets_slo_tpot")})
pb = d["roofline_extra"]["prefill_batch_ms"]
print("roofline", d["roofline"]["frac"], "prefill_gemm", d["roofline_extra"]["prefill_gemm"]["frac"], "sat", d["saturation"]["output_tok_s"], "prefill batch", pb["forward_and_sync"], pb["avg_tokens"])
PY
