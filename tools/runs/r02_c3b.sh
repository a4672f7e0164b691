O=gpurun_out/r02_c3b; mkdir -p $O
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline > $O/bench_config3.json 2> $O/err.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_c3b/bench_config3.json").read().strip().splitlines()[-1])
print(d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],1), d.get("saturation",{}).get("output_tok_s"), d.get("static_split_50_50"))
PY
