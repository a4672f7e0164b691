#!/bin/bash
# round 5, call 42: the tiled GEMM's K-slice planes summed by the fused add + norm (semipd_gemm_tall_planes): parity (bits of the reducing
# form), then the pair alone, whole chip and on the prefill share's CU count
OUT=gpurun_out/r05_s42; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "tall_planes or defers or gemm_tall" > $OUT/pytest_tall.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_tall.txt | cut -c1-200
timeout 300 python tools/kbench_tall_planes.py 2>&1 | tail -7 | tee $OUT/kbench_tall_planes.txt
MASK=$(python -c "
import sys; sys.path[:0]=['semi-pd_amd']
from semi_pd_amd.semi_pd.utils import cu_mask_env
print(cu_mask_env(0, 256, 88, False)['HSA_CU_MASK'])" 2>/dev/null)
if [ -n "$MASK" ]; then HSA_CU_MASK=$MASK KBENCH_CUS=224 timeout 300 python tools/kbench_tall_planes.py 2>&1 | tail -7 | cut -c1-250 | tee -a $OUT/kbench_tall_planes.txt; fi
