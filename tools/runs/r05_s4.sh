#!/bin/bash
# round 5, call 4: k-block rotation in the streaming GEMM (workgroups no longer walk K in lock-step): A / B on the 48 private CUs and
# the whole chip, correctness of everything that calls the kernel, the decode step
OUT=gpurun_out/r05_s4; mkdir -p $OUT
for rot in 0 1; do
  echo "== SEMIPD_SL_ROT=$rot"
  SEMIPD_SL_ROT=$rot HSA_CU_MASK=0:208-255 KBENCH_NUM_CUS=256 KBENCH_MS=32 timeout 400 python tools/kbench.py stream_planes_graph 2>&1 | grep -v "Warning\|amdgpu.ids"
  SEMIPD_SL_ROT=$rot KBENCH_NUM_CUS=256 KBENCH_MS=32 timeout 400 python tools/kbench.py stream_planes_graph 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee $OUT/stream_planes_graph_rot_ab.txt
for rot in 0 1; do
  SEMIPD_SL_ROT=$rot timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | cut -c1-140
  SEMIPD_SL_ROT=$rot HSA_CU_MASK=0:208-255 timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 50 2>&1 | grep "ms per decode" | cut -c1-140
done | tee $OUT/steps.txt
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_mla_prep.py -q -x -k "stream or linear or planes or lm_head or moe or mla" > $OUT/pytest_stream.txt 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest_stream.txt | cut -c1-220
