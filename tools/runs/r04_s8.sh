#!/bin/bash
# round 4, call 8: the fused rope + attention decode step; the whole GPU suite with durations
OUT=gpurun_out/r04_s8; mkdir -p $OUT
for m in llama3-8b llama3-70b-tp8-rank; do
  for f in 1 0; do
    SEMIPD_FUSE_ROPE_DECODE=$f timeout 300 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | sed "s/^/fuse=$f /"
  done
done > $OUT/step_fused_rope.txt 2>&1
cat $OUT/step_fused_rope.txt | cut -c1-110
timeout 2400 python -m pytest tests -m gpu -q --durations=150 > $OUT/pytest_gpu_full.txt 2>&1; echo "pytest full rc=$?"
grep -E "passed|failed" $OUT/pytest_gpu_full.txt | tail -3
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_full.txt | head -20
grep -E "^[0-9.]+s (call|setup|teardown)" $OUT/pytest_gpu_full.txt | head -90 | cut -c1-150
