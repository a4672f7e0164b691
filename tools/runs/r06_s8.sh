#!/bin/bash
# round 6, call 8: fused decode launch v2 with the merge's roundings pinned (fma in stage 2 and in the fused merge): bits, engine tests,
# rocprofv3 per-kernel stats of the decode step alone with the fused launch off / on
OUT=gpurun_out/r06_s8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "decode or rope" > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.txt | cut -c1-300
export TMPDIR=/tmp
for F in 0 1; do
  ( cd /tmp && SEMIPD_FUSED_DECODE_ATTN=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$F -o step -- python $GRAFT_REPO_ROOT/tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 200 > /tmp/prof_f$F.log 2>&1 )
  tail -3 /tmp/prof_f$F.log | cut -c1-200
  S=$(find /tmp/prof_f$F -name "*kernel_stats.csv" | head -1)
  [ -z "$S" ] && find /tmp/prof_f$F | head
  cp "$S" $OUT/decode_step_b32_fused${F}_kernel_stats.csv
  python tools/stats_top.py $OUT/decode_step_b32_fused${F}_kernel_stats.csv | head -16
done
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_depth.py -q -x > $OUT/pytest_engine.txt 2>&1; echo "engine rc=$?"; tail -3 $OUT/pytest_engine.txt | cut -c1-300
