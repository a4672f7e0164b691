#!/bin/bash
# round 4, call 26: routing kernel with a DPP argmax, no group stage for one group; q_a_layernorm + quantisation in one kernel: goldens, parity, DeepSeek steps
OUT=gpurun_out/r04_s26; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py tests/test_gpu_ops.py tests/test_gpu_torch_ops.py -q -x -k "topk or moe or rmsnorm or mla" > $OUT/pytest_topk.txt 2>&1; echo "pytest topk rc=$?"
tail -4 $OUT/pytest_topk.txt | cut -c1-220
timeout 1200 python -m pytest tests/test_gpu_deepseek.py tests/test_gpu_rank_widths.py -q -x -k "deepseek or v3" > $OUT/pytest_engines.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_engines.txt | cut -c1-220
timeout 600 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | cut -c1-120 | tee $OUT/steps.txt
timeout 300 python tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | cut -c1-120 | tee -a $OUT/steps.txt
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v2l -- python $R/tools/decode_step_bench.py --model deepseek-v2-lite --batch 32 --ctx 1100 --steps 100 > $R/$OUT/prof_v2l.log 2>&1 )
f=$(find /tmp/prof_v2l -name "*kernel_stats.csv" | head -1); cp $f $OUT/v2lite_b32_decode_step_kernel_stats_final.csv
python tools/stats_top.py $f | grep -E "topk|align|moe_sum|mla_decode_prep|bmm_nk|rmsnorm|splitk" | cut -c1-150
