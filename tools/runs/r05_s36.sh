#!/bin/bash
# round 5, call 36: moe_topk_kernel with its plane loads in flight (4 experts x 8 planes per lane): parity, then the decode step alone of the two MoE models
OUT=gpurun_out/r05_s36; mkdir -p $OUT
R=$(pwd)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mla_prep.py tests/test_gpu_ops.py -q -k "topk" > $OUT/pytest_topk.txt 2>&1; echo "pytest topk rc=$?"; tail -3 $OUT/pytest_topk.txt | cut -c1-200
timeout 400 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | cut -c1-100 | tee -a $OUT/decode_step.txt
m=deepseek-v3-tp8-rank
timeout 400 python tools/decode_step_bench.py --model $m --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | cut -c1-100 | tee -a $OUT/decode_step.txt
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python $R/tools/decode_step_bench.py --model $m --quantization fp8 --batch 32 --ctx 1100 --steps 20 > $R/$OUT/prof_$m.log 2>&1 )
f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${m}_kernel_stats.csv && python - <<PY | tee -a $OUT/decode_step.txt
import csv
rows = list(csv.DictReader(open("$OUT/${m}_kernel_stats.csv")))
for r in rows[:40]:
    n = r["Name"].replace("void semipd::", "").split("(")[0]
    if n.startswith("void at::") or "rocclr" in n: continue
    print(f"  {n[:64]:64s} calls={int(r['Calls']):6d} ({int(r['Calls'])/39:6.1f} per replay) avg={float(r['AverageNs'])/1e3:7.1f}us total={float(r['TotalDurationNs'])/1e6:8.1f}ms")
PY
