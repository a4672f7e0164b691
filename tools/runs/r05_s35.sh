#!/bin/bash
# round 5, call 35: RMSNorm with one vector per lane up to hidden 8192 (rms_threads): parity tests, then A/B against the old widths (SEMIPD_RMS_WIDE=0):
# the prefill-sized calls alone and the decode step alone (Llama-3-8B, Llama-3-70B TP = 8 rank shapes, DeepSeek-V3 rank shapes)
OUT=gpurun_out/r05_s35; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_gemm.py tests/test_gpu_mla_prep.py -q -k "norm or planes or quant" > $OUT/pytest_norm.txt 2>&1; echo "pytest norm rc=$?"; tail -3 $OUT/pytest_norm.txt | cut -c1-200
for w in 0 1 0 1; do
  export SEMIPD_RMS_WIDE=$w
  echo "== SEMIPD_RMS_WIDE=$w" | tee -a $OUT/ab.txt
  timeout 300 python tools/kbench_small_prefill_ops.py 2>&1 | grep "T=" | tee -a $OUT/ab.txt
  for m in llama3-8b llama3-70b-tp8-rank; do
    timeout 400 python tools/decode_step_bench.py --model $m --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | cut -c1-90 | tee -a $OUT/ab.txt
  done
done
for w in 0 1; do
  export SEMIPD_RMS_WIDE=$w
  echo "== SEMIPD_RMS_WIDE=$w" | tee -a $OUT/ab.txt
  timeout 400 python tools/decode_step_bench.py --model deepseek-v3-tp8-rank --quantization fp8 --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode step" | cut -c1-90 | tee -a $OUT/ab.txt
done
