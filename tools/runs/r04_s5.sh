#!/bin/bash
# round 4, call 5: why is a masked STREAM slower than a masked PROCESS for the prefill instance? (prefill-only load under
# rocprofv3, both mechanisms); whole-chip prefill with tiled (non stream-K) library GEMMs; declared CU counts of the decode grids
OUT=gpurun_out/r04_s5; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
COMMON="--steps 1 --warmup 1 --rate-sweep '' --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave"
( cd /tmp && export TMPDIR=/tmp
  for mode in env dynamic; do
    SEMIPD_CU_SHARE_FORCE=share timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python $R/bench.py --steps 1 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave --no-kernel-timing --num-requests 64 --request-rate 0 --output-len 1 --cu-mask-mode $mode --prefill-cu 81 --decode-cu 100 > $R/$OUT/ponly_$mode.json 2> $R/$OUT/ponly_$mode.err
    for f in $(find /tmp/prof_$mode -name "*kernel_stats.csv"); do n=$(python $R/tools/stats_top.py $f | grep -c "extend_attn"); if [ "$n" -gt 0 ]; then cp $f $R/$OUT/ponly_${mode}_prefill_kernel_stats.csv; fi; done
  done )
for mode in env dynamic; do echo "== prefill-only, $mode"; python tools/stats_top.py $OUT/ponly_${mode}_prefill_kernel_stats.csv | head -16; python - <<PY
import json
d=json.loads(open("$OUT/ponly_$mode.json").read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"], d["roofline_extra"].get("prefill_batch_ms"))
PY
done > $OUT/ponly_summary.txt 2>&1
run() { name=$1; shift; timeout 600 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
SEMIPD_DG_EXCLUDE=_SK3 run p100_d100_tiled --prefill-cu 100 --decode-cu 100 --tune-prefill-gemm
run p100_d100_tuned --prefill-cu 100 --decode-cu 100 --tune-prefill-gemm
SEMIPD_DECLARED_CUS_DECODE=128 run env_p81_d100_decl128 --prefill-cu 81 --decode-cu 100
SEMIPD_DECLARED_CUS_DECODE=512 run env_p81_d100_decl512 --prefill-cu 81 --decode-cu 100
SEMIPD_DG_EXCLUDE=_SK3 run env_p88_d100_tiled --prefill-cu 88 --decode-cu 100
timeout 600 python -m pytest "tests/test_gpu_rank_widths.py::test_deepseek_v3_tp8_rank_width_block_fp8_unified_and_semi_pd_match_the_oracle" "tests/test_gpu_ops.py::test_dense_gemm_with_measured_library_solution_matches_fp32" -q -s > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    if "ponly" in f: continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), s.get("p50_tbt_ms"), "frac", (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"), d["roofline_extra"].get("decode_step_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
cat $OUT/ponly_summary.txt; tail -3 $OUT/pytest_new.txt
