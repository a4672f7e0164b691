#!/bin/bash
# round 5, call 31: the one-pass RoPE + KV-store kernel: parity tests, the prefill-side small kernels alone, then SURVEY 8d.2's sweep on the 50 / 50 split for in = 128 / out = 64
OUT=gpurun_out/r05_s31; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py tests/test_gpu_torch_ops.py -q -k "rope or kv or rotary" > $OUT/pytest_rope.txt 2>&1; echo "pytest rope rc=$?"; tail -3 $OUT/pytest_rope.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_full_depth.py tests/test_gpu_deepseek.py -q -k "unified or depth or deepseek_semi_pd or chunked" > $OUT/pytest_engine.txt 2>&1; echo "pytest engine rc=$?"; tail -3 $OUT/pytest_engine.txt | cut -c1-200
timeout 300 python tools/kbench_small_prefill_ops.py 2>&1 | grep "T=" | tee $OUT/small_prefill_ops.txt
bash tools/runs/r05_s19.sh d 2>&1 | tail -12
