O=gpurun_out/r02_v3slice3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_deepseek.py -q -x -k "decode_attention or mla or deepseek" 2>&1 | tail -3
timeout 900 python bench.py --model deepseek-v3-slice --quantization fp8 --num-requests 96 --request-rate 8 --no-cpu-baseline --no-static-split-wave > $O/bench_deepseek_v3_slice_fp8.json 2> $O/err.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_v3slice3/bench_deepseek_v3_slice_fp8.json").read().strip().splitlines()[-1])
print(d["value"], d.get("p50_ttft_ms"), d.get("p50_tbt_ms"), d.get("p99_tbt_ms")); print(d["roofline"]); print(d["saturation"])
PY
