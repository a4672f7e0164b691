#!/bin/bash
# round 6, call 7: fused decode launch v2 (the wave that owns the new token stores its rows, raw barrier on the q LDS writes only,
# K rows of tile 1 asked for before the rotation): bits, the decode step alone off / on, rocprofv3 per-kernel stats of both
OUT=gpurun_out/r06_s7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "decode or rope" > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.txt | cut -c1-300
for B in 32 64 24; do
  for F in 0 1; do
    SEMIPD_FUSED_DECODE_ATTN=$F timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch $B --ctx 1100 > $OUT/step_b${B}_fused$F.txt 2>&1
    echo "B=$B fused=$F: $(grep 'ms per decode step' $OUT/step_b${B}_fused$F.txt | cut -c1-120)"
  done
done
export TMPDIR=/tmp
for F in 0 1; do
  ( cd /tmp && SEMIPD_FUSED_DECODE_ATTN=$F timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f$F -o step -- python $GRAFT_REPO_ROOT/tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 200 > /tmp/prof_f$F.log 2>&1 )
  S=$(find /tmp/prof_f$F -name "*kernel_stats.csv" | head -1)
  cp "$S" $OUT/decode_step_b32_fused${F}_kernel_stats.csv
  python tools/stats_top.py $OUT/decode_step_b32_fused${F}_kernel_stats.csv | head -16
done
