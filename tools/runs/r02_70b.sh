O=gpurun_out/r02_70b; mkdir -p $O
timeout 1500 python bench.py --model llama3-70b --num-requests 64 --request-rate 4 --no-cpu-baseline > $O/bench_llama3_70b_tp1.json 2> $O/err.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_70b/bench_llama3_70b_tp1.json").read().strip().splitlines()[-1])
print(d["config"]["workload"]); print(d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],1), d["roofline"]["achieved"], d["roofline"]["frac"], d.get("saturation",{}).get("output_tok_s"), d.get("static_split_50_50"))
PY
timeout 900 python bench.py --model deepseek-v3-slice --quantization fp8 --num-requests 96 --request-rate 8 --no-cpu-baseline --no-static-split-wave > $O/bench_deepseek_v3_slice_fp8.json 2> $O/err2.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_70b/bench_deepseek_v3_slice_fp8.json").read().strip().splitlines()[-1])
print(d["config"]["workload"]); print(d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],1), d.get("saturation",{}).get("output_tok_s"))
PY
tail -3 $O/err.txt $O/err2.txt | cut -c1-300
