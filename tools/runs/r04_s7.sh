#!/bin/bash
# round 4, call 7: EP all-to-all tests; runtime knobs on the decode step; the new default bench line end to end; the whole GPU
# suite with durations
OUT=gpurun_out/r04_s7; mkdir -p $OUT
timeout 700 python -m pytest tests/test_gpu_ep_all_to_all.py "tests/test_gpu_deepseek.py::test_deepseek_tp2_expert_all_to_all_on_one_gpu" tests/test_gpu_cu_share.py -x -q -s --durations=5 > $OUT/pytest_ep.txt 2>&1; echo "pytest ep rc=$?"
grep -E "passed|failed|^E  " $OUT/pytest_ep.txt | head -12
for kv in "BASE=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=8" "HSA_NO_SCRATCH_RECLAIM=1" "AMD_DIRECT_DISPATCH=0" "HIP_LAUNCH_BLOCKING=0 DEBUG_HIP_GRAPH_DOT_PRINT=0"; do
  echo "== $kv"; env $kv timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode"
done > $OUT/step_env_knobs.txt 2>&1
cat $OUT/step_env_knobs.txt | cut -c1-120
timeout 1500 python bench.py --steps 2 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    s = d.get("saturation") or {}
    print("default", d["value"], "TTFT", d["p50_ttft_ms"], d["p99_ttft_ms"], "TBT", d["p50_tbt_ms"], d["p99_tbt_ms"], "sat", s.get("output_tok_s"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
    print(" static", d.get("static_split_50_50")); print(" c1", d.get("config1_opt_125m")); print(" c3", d.get("config3_deepseek_v2_lite"))
    print(" sweep", d.get("qps_sweep")); print(" cpu", d.get("cpu_baseline")); print(" prefill", d["roofline_extra"].get("prefill_batch_ms")); print(" decode", d["roofline_extra"].get("decode_step_ms"))
    print(" workload:", d["config"]["workload"])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $OUT/bench_default.err | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --durations=120 -x > $OUT/pytest_gpu_full.txt 2>&1; echo "pytest full rc=$?"
tail -140 $OUT/pytest_gpu_full.txt | cut -c1-160
