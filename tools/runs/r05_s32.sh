#!/bin/bash
# round 5, call 32: rope_planes_kernel over (tokens, parts) workgroups: parity tests, then the decode step alone (graph replay, B = 32, ctx 1100) of Llama-3-8B and of the
# Llama-3-70B TP = 8 rank shapes, with rocprofv3 kernel stats of each (verdict r04 item 7's per-kernel table; before = profiles/r04_decode_step_rank_shapes.txt)
OUT=gpurun_out/r05_s32; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py -q -k "rope or planes or kv" > $OUT/pytest_rope.txt 2>&1; echo "pytest rope rc=$?"; tail -3 $OUT/pytest_rope.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_depth.py -q -k "unified or depth" > $OUT/pytest_engine.txt 2>&1; echo "pytest engine rc=$?"; tail -3 $OUT/pytest_engine.txt | cut -c1-200
for m in llama3-8b llama3-70b-tp8-rank; do
  timeout 300 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | tee -a $OUT/decode_step.txt
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$m -o run -- python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 20 > $OUT/prof_$m.log 2>&1
  grep "ms per decode step" $OUT/prof_$m.log | sed 's/^/under rocprof: /' | tee -a $OUT/decode_step.txt
  python tools/stats_top.py $(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1) 2>&1 | tee -a $OUT/decode_step.txt
  find $OUT/prof_$m -type f ! -name "*kernel_stats.csv" -delete
done
