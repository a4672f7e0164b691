#!/bin/bash
OUT=gpurun_out/r03_dbg4; mkdir -p $OUT
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/tune_reject_probe.txt
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "semi-pd_amd")]
import torch
from semi_pd_amd import ops
for (n, k) in ((512, 512), (1024, 512), (2048, 512), (512, 1024), (4096, 4096), (28672, 4096)):
    ops.dense_gemm_tune(n, k, [1024, 2048, 128, 256, 512], torch.bfloat16, num_full_search=2)
print(ops.dense_gemm_report())
PY
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "planes or dense_gemm or logits_processor_above" 2>&1 | tail -3
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "overlapped_decode_loop or semi_pd_matches_unified" 2>&1 | grep -v "amdgpu.ids" > $OUT/loop_$i.txt; if grep -q "failed" $OUT/loop_$i.txt; then echo "run $i FAILED"; grep -n "^E " $OUT/loop_$i.txt | head -5; else echo "run $i ok"; fi; grep -h "wrong_results_rejected=[1-9]" $OUT/loop_$i.txt | head -3; done
