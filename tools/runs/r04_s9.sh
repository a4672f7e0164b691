#!/bin/bash
# round 4, call 9: the shortened small kernels (plane sums, stage 2) + fused rope attention: parity tests, decode steps, the line
OUT=gpurun_out/r04_s9; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ep_all_to_all.py tests/test_gpu_fp8_kv.py "tests/test_gpu_rank_widths.py" tests/test_gpu_full_width.py -q -x --durations=15 > $OUT/pytest_kernels.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^E  " $OUT/pytest_kernels.txt | head -12
for m in llama3-8b llama3-70b-tp8-rank; do
  timeout 300 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode"
done > $OUT/steps.txt 2>&1
timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" >> $OUT/steps.txt
cut -c1-100 $OUT/steps.txt
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_8b -- python $R/tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 50 > $R/$OUT/prof_8b.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -size +1M -delete
python tools/stats_top.py $(find $OUT/prof_8b -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -16 | cut -c1-140
timeout 900 python bench.py --steps 2 --warmup 1 --no-side-configs --no-static-split-wave --rate-sweep "" > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    s = d.get("saturation") or {}
    print("default", d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), "sat", s.get("output_tok_s"), "frac", d["roofline"]["frac"], d["roofline_extra"].get("decode_step_ms"), d["roofline_extra"].get("prefill_batch_ms"))
except Exception as e:
    print("bench parse failed", e)
PY
