O=gpurun_out/r02_lmhead; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for m in 64 100000; do
  SEMIPD_LM_HEAD_FUSED_MAX_ROWS=$m timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave > $O/bench_$m.json 2> $O/err_$m.txt
  python - "$O/bench_$m.json" "$m" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["saturation"]
print("fused up to", sys.argv[2], "rows: tok/s", d["value"], "tbt", round(d["p50_tbt_ms"], 2), round(d["p99_tbt_ms"], 2), "| saturation", s["output_tok_s"], "tbt", s["p50_tbt_ms"], s["p99_tbt_ms"])
PY
done
