#!/bin/bash
# round 2: first GPU run of the persistent streaming linear: parity tests, then knob sweep on the full chip and on the 128-CU share
OUT=gpurun_out/r02_stream1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stream_linear or golden" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
export KBENCH_MS=16,48
( export HSA_CU_MASK=0:0-127 KBENCH_NUM_CUS=128; timeout 900 python tools/kbench.py stream_linear ) > $OUT/sweep_half.txt 2>&1
( export KBENCH_NUM_CUS=256; timeout 900 python tools/kbench.py stream_linear ) > $OUT/sweep_full.txt 2>&1
grep -v amdgpu.ids $OUT/sweep_half.txt; grep -v amdgpu.ids $OUT/sweep_full.txt
