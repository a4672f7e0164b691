#!/bin/bash
# round 4, call 16: the MLA decode step in fewer launches (merged q | kv_a GEMM, mla_decode_prep, bmm_nk): parity, engine tests, step time A / B
OUT=gpurun_out/r04_s16; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py -q -x > $OUT/pytest_mla_prep.txt 2>&1; echo "pytest mla prep rc=$?"
tail -4 $OUT/pytest_mla_prep.txt | cut -c1-220
for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
  SEMIPD_MLA_MERGED_QKV_A=$1 SEMIPD_MLA_OWN_BMM=$2 timeout 300 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | sed "s/^/merged=$1 own_bmm=$2 /" | cut -c1-120
done | tee $OUT/steps.txt
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v2l -- python $R/tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 > $R/$OUT/prof_v2l.log 2>&1 )
f=$(find /tmp/prof_v2l -name "*kernel_stats.csv" | head -1); cp $f $OUT/v2lite_b32_decode_step_kernel_stats_fused.csv
python tools/stats_top.py $f | head -24 | cut -c1-150
timeout 1200 python -m pytest tests/test_gpu_deepseek.py -q -x > $OUT/pytest_deepseek.txt 2>&1; echo "pytest deepseek rc=$?"
tail -4 $OUT/pytest_deepseek.txt | cut -c1-220
