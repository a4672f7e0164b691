#!/bin/bash
# round 5, call 37: moe_topk_kernel in registers: golden + random + planes parity, the DeepSeek engine tests, then the decode step alone of the two MoE models
OUT=gpurun_out/r05_s37; mkdir -p $OUT
R=$(pwd)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mla_prep.py tests/test_gpu_ops.py -q -k "topk or moe" > $OUT/pytest_topk.txt 2>&1; echo "pytest topk rc=$?"; tail -5 $OUT/pytest_topk.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_deepseek.py -q -x > $OUT/pytest_deepseek.txt 2>&1; echo "pytest deepseek rc=$?"; tail -3 $OUT/pytest_deepseek.txt | cut -c1-200
timeout 400 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | cut -c1-100 | tee -a $OUT/decode_step.txt
m=deepseek-v3-tp8-rank
timeout 400 python tools/decode_step_bench.py --model $m --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | cut -c1-100 | tee -a $OUT/decode_step.txt
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python $R/tools/decode_step_bench.py --model $m --quantization fp8 --batch 32 --ctx 1100 --steps 20 > $R/$OUT/prof_$m.log 2>&1 )
f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${m}_kernel_stats.csv && grep -E "moe_topk|moe_align|moe_sum" $OUT/${m}_kernel_stats.csv | awk -F'","' '{print substr($1,1,60), $2, $4}' | tee -a $OUT/decode_step.txt
