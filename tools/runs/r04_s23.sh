#!/bin/bash
# round 4, call 23: DeepSeek-V3 rank shapes, one decode step per kernel (rocprofv3)
OUT=gpurun_out/r04_s23; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v3 -- python $R/tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 60 > $R/$OUT/prof_v3.log 2>&1 )
grep "ms per decode" $OUT/prof_v3.log | cut -c1-120
f=$(find /tmp/prof_v3 -name "*kernel_stats.csv" | head -1); cp $f $OUT/v3rank_b32_decode_step_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/v3rank_b32_decode_step_kernel_stats.csv")))
# steps: the argmax kernel runs once per step
steps=[int(r['Calls']) for r in rows if 'argmax_kernel' in r['Name']][0]
print("steps", steps)
tot=0
for r in rows:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])/1e3
    if n>=steps-5 and t/steps>=20:
        print(f"{n/steps:6.1f}/step x {t/n:7.1f} us = {t/steps/1e3:6.3f} ms/step  {r['Name'][:110]}"); tot+=t/steps/1e3
print("listed", round(tot,2))
PY
