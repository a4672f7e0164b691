#!/bin/bash
OUT=gpurun_out/r02_dsfp8; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_deepseek.py -x -q -m gpu -k block_fp8_unified > $OUT/pytest.txt 2>&1; grep -E "Error|assert|engine token|oracle logit|FAILED|passed|failed" $OUT/pytest.txt | head -30
SEMIPD_MLA_ABSORB_BF16=1 timeout 1200 python -m pytest tests/test_gpu_deepseek.py -x -q -m gpu -k block_fp8_unified > $OUT/pytest_bf16absorb.txt 2>&1; tail -3 $OUT/pytest_bf16absorb.txt
