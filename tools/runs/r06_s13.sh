#!/bin/bash
# round 6, call 13: the streaming GEMM with four-wave workgroups (two per CU) at <= 32 rows against the eight-wave form, by K split;
# the new engine test (fused launch forced at every batch size, 80 requests at once); stream_linear tests with the narrow form forced
OUT=gpurun_out/r06_s13; mkdir -p $OUT
timeout 900 python tools/kbench_narrow.py 2>&1 | grep -v amdgpu.ids | tee $OUT/kbench_narrow.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x -k "fused_decode_launch or unified_llama_matches" > $OUT/pytest_engine.txt 2>&1; echo "engine rc=$?"; tail -3 $OUT/pytest_engine.txt | cut -c1-300
SEMIPD_SL_NW=4 timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "stream_linear or rope_and_store_kv_from or decode_rope" > $OUT/pytest_ops_nw4.txt 2>&1; echo "ops (narrow forced) rc=$?"; tail -3 $OUT/pytest_ops_nw4.txt | cut -c1-300
