#!/bin/bash
# round 4, call 6: bisect "dynamic worse than env" (stream kind of the decode instance vs what the prefill instance launches
# outside its masked stream); EP all-to-all tests; the fixed rank-width test
OUT=gpurun_out/r04_s6; mkdir -p $OUT
run() { name=$1; shift; timeout 600 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
SEMIPD_DECODE_OWN_STREAM=1 run env_p81_d100_decode_own_stream --prefill-cu 81 --decode-cu 100
SEMIPD_DYN_ALSO_ENV_MASK=1 run dyn_p81_d100_also_env --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 100
SEMIPD_DYN_FULL_ON_NULL_STREAM=1 run dyn_p81_d100_null_full --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 100
run dyn_p81_d100 --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 100
run env_p81_d100 --prefill-cu 81 --decode-cu 100
timeout 900 python -m pytest tests/test_gpu_ep_all_to_all.py "tests/test_gpu_deepseek.py::test_deepseek_tp2_expert_all_to_all_on_one_gpu" "tests/test_gpu_rank_widths.py::test_deepseek_v3_tp8_rank_width_block_fp8_unified_and_semi_pd_match_the_oracle" -q -s --durations=5 > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "frac", (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms",{}).get("forward_and_sync"), d["roofline_extra"].get("decode_step_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
grep -E "passed|failed|^E  " $OUT/pytest_new.txt | head -20
