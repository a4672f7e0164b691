#!/bin/bash
# round 6, call 6: the fused decode-step launch (plane sum + RoPE + KV store + attention + split merge, csrc/decode_attention_fused.hip):
# its bits against the three launches it replaces, the decode / rope tests the shared walk header touches, the Llama engine tests,
# then the decode step alone with the fused launch off / on at 8 .. 128 requests
OUT=gpurun_out/r06_s6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "decode or rope" > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x > $OUT/pytest_engine.txt 2>&1; echo "engine rc=$?"; tail -3 $OUT/pytest_engine.txt | cut -c1-300
for B in 32 8 16 64 128; do
  for F in 0 1; do
    K=""; [ $B = 32 ] && K="--kernels"
    SEMIPD_FUSED_DECODE_ATTN=$F timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch $B --ctx 1100 $K > $OUT/step_b${B}_fused$F.txt 2>&1
    echo "B=$B fused=$F: $(grep 'ms per decode step' $OUT/step_b${B}_fused$F.txt | cut -c1-120)"
  done
done
grep -h " x " $OUT/step_b32_fused0.txt | head -14
echo ---
grep -h " x " $OUT/step_b32_fused1.txt | head -14
