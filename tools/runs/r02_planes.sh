O=gpurun_out/r02_planes; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "qkv_planes" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fp8_kv.py -q -x 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_planes/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],1), d["roofline"]["frac"], d["roofline"]["avg_launch_us"], (d.get("saturation") or {}).get("output_tok_s"))
PY
