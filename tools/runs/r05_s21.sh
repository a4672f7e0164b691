#!/bin/bash
# round 5, call 21: validation of the final tree -- the whole GPU suite, smoke(), the N = 2 command shape on one GPU (TP = 2 over gloo + the
# peer-memory kernels, pacer and share agreement on), then the default bench line (3 timed waves)
OUT=gpurun_out/r05_s21; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"
tail -22 $OUT/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --model llama-tiny --steps 1 --warmup 1 --num-requests 48 --request-rate 8 --no-cpu-baseline --mem-fraction-static 0.3 > $OUT/bench_tp2_one_gpu.json 2> $OUT/bench_tp2.err; echo "tp2 rc=$?"
python tools/summarize_runs.py $OUT/bench_tp2_one_gpu.json
T0=$(date +%s)
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python tools/summarize_runs.py $OUT/bench_default.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
for k in ("static_split_50_50", "unified_same_load", "saturation", "config1_opt_125m", "config3_deepseek_v2_lite"):
    v = d.get(k) or {}
    print(k, {kk: v.get(kk) for kk in ("timed_waves", "output_tok_s", "p50_ttft_ms", "p99_ttft_ms", "p50_tbt_ms", "p99_tbt_ms", "error") if kk in v})
print("gate", d["roofline_extra"]["prefill_batch_ms"].get("step_gate"))
PY
