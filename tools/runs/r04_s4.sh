#!/bin/bash
# round 4, call 4: GEMM-loop anatomy on 96 / 256 CUs; stream-mask vs process-mask as a neighbour; new parity tests;
# kernel breakdown of the rank-shape decode steps; GEMM solution names
OUT=gpurun_out/r04_s4; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
: > $OUT/hbm_anatomy.txt
for sh in "96 8" "128 8" "256 8"; do
  timeout 90 tools/hbm_cu_probe $sh 2>&1 | grep -v "^lds\|^regs" >> $OUT/hbm_anatomy.txt || echo "share $sh: rc=$?" >> $OUT/hbm_anatomy.txt
done
timeout 300 python tools/mask_equiv_probe.py > $OUT/mask_equiv.txt 2>&1; echo "mask equiv rc=$?"
timeout 900 python -m pytest tests/test_gpu_rank_widths.py tests/test_gpu_full_width.py "tests/test_gpu_ops.py::test_dense_gemm_with_measured_library_solution_matches_fp32" -q -s --durations=10 > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?"
( cd /tmp && export TMPDIR=/tmp
  for m in llama3-8b llama3-70b-tp8-rank; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$m -- python $R/tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 20 > $R/$OUT/prof_$m.log 2>&1
  done
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_v3rank -- python $R/tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 20 > $R/$OUT/prof_v3rank.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -size +1M -delete
for d in $OUT/prof_*/; do echo "== $d"; python tools/stats_top.py $(find $d -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -24; done > $OUT/step_kernel_stats.txt
timeout 600 python bench.py --steps 1 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave --prefill-cu 81 --decode-cu 100 > $OUT/env_p81_names.json 2> $OUT/env_p81_names.err
grep "kernel=" $OUT/env_p81_names.err | sed 's/us=.*kernel=/ /' | cut -c1-220 | head -60 > $OUT/solution_names.txt
tail -4 $OUT/pytest_new.txt; cat $OUT/mask_equiv.txt | grep -v amdgpu; cat $OUT/hbm_anatomy.txt
