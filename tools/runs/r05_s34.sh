#!/bin/bash
# round 5, call 34: per-kernel table of DeepSeek-V2-Lite's decode step alone (config 3's model; graph replay, B = 32, ctx 1100)
OUT=gpurun_out/r05_s34; mkdir -p $OUT
R=$(pwd)
export TMPDIR=/tmp
for m in deepseek-v2-lite; do
  timeout 400 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 50 --kernels 2>&1 | grep -v Warning | tee -a $OUT/decode_step.txt
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python $R/tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 20 > $R/$OUT/prof_$m.log 2>&1 )
  grep "ms per decode step" $OUT/prof_$m.log | sed 's/^/under rocprof: /' | tee -a $OUT/decode_step.txt
  f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${m}_kernel_stats.csv && python tools/stats_top.py $OUT/${m}_kernel_stats.csv 2>&1 | tee -a $OUT/decode_step.txt
done
