#!/bin/bash
# round 6, call 19: the fused decode launch with several workgroups per (request, kv head) for small batches (two launches instead of
# three): op tests, the restructured full-depth parity test, engine tests; decode step alone at small batches, SEMIPD_FUSED_DECODE_ATTN=4
# (only where one workgroup per pair fills the chip) against the default
OUT=gpurun_out/r06_s19; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "decode_rope or decode_attention" > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_full_depth.py -q -x -s > $OUT/pytest_full_depth.txt 2>&1; echo "full depth rc=$?"; grep -h "token-for-token\|near-tie\|oracle\|passed\|failed" $OUT/pytest_full_depth.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_cu_share.py -q -x > $OUT/pytest_engine.txt 2>&1; echo "engine rc=$?"; tail -3 $OUT/pytest_engine.txt | cut -c1-300
for M in llama3-8b llama3-70b-tp8-rank; do
  for B in 1 4 8 32; do
    [ $M = llama3-8b ] && [ $B = 32 ] && continue
    for F in 4 1 4 1; do
      SEMIPD_FUSED_DECODE_ATTN=$F timeout 300 python tools/decode_step_bench.py --model $M --batch $B --ctx 1100 2>&1 | grep "ms per decode step" | cut -c1-75 | sed "s/^/FUSED=$F /" | tee -a $OUT/steps.txt
    done
  done
done
