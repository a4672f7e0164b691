#!/bin/bash
# round 5, call 33: per-kernel table of the decode step alone (graph replay, B = 32, ctx 1100): Llama-3-8B, the Llama-3-70B TP = 8 rank shapes, the DeepSeek-V3 TP = 8 rank shapes
# (verdict r04 item 7; before = profiles/r04_decode_step_rank_shapes.txt)
OUT=gpurun_out/r05_s33; mkdir -p $OUT
R=$(pwd)
export TMPDIR=/tmp
for m in llama3-8b llama3-70b-tp8-rank deepseek-v3-tp8-rank; do
  q=""; [ $m = deepseek-v3-tp8-rank ] && q="--quantization fp8"
  timeout 400 python tools/decode_step_bench.py --model $m $q --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | tee -a $OUT/decode_step.txt
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python $R/tools/decode_step_bench.py --model $m $q --batch 32 --ctx 1100 --steps 20 > $R/$OUT/prof_$m.log 2>&1 )
  grep "ms per decode step" $OUT/prof_$m.log | sed 's/^/under rocprof: /' | tee -a $OUT/decode_step.txt
  f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${m}_kernel_stats.csv && python tools/stats_top.py $OUT/${m}_kernel_stats.csv 2>&1 | tee -a $OUT/decode_step.txt
done
