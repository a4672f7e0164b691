#!/bin/bash
# round 4, call 14: where a DeepSeek-V2-Lite decode step goes, per kernel (config 3): tools/decode_step_bench.py under rocprofv3
OUT=gpurun_out/r04_s14; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v2l -- python $R/tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 > $R/$OUT/prof_v2l.log 2>&1 )
grep "ms per decode" $OUT/prof_v2l.log | cut -c1-120
f=$(find /tmp/prof_v2l -name "*kernel_stats.csv" | head -1); cp $f $OUT/v2lite_b32_decode_step_kernel_stats.csv
python tools/stats_top.py $f | head -40 | cut -c1-150
