#!/bin/bash
# round 6, call 3: the 4-wave form of the tiled GEMM -- parity tests (fp32 products, the bits of the 8-wave form, SiLU epilogue,
# race screen), then the kernel bench of the Llama-3-8B prefill layers: library | 8 waves | 4 waves, whole chip and under the
# prefill share's mask (224 CUs)
OUT=gpurun_out/r06_s3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "four_wave or gemm_tall" > $OUT/pytest_gemm.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gemm.txt | cut -c1-300
timeout 600 python tools/kbench_gemm_forms.py > $OUT/kbench_whole_chip.txt 2>&1; echo "kbench rc=$?"; cat $OUT/kbench_whole_chip.txt
MASK=$(python - <<PY
import sys; sys.path.insert(0, "semi-pd_amd")
from semi_pd_amd.semi_pd.utils import cu_mask_env
print(cu_mask_env(0, 256, 88, False)["HSA_CU_MASK"])
PY
)
echo "mask: $MASK"
HSA_CU_MASK=$MASK KBENCH_CUS=224 KBENCH_ROWS=1024,1411,2048 timeout 600 python tools/kbench_gemm_forms.py > $OUT/kbench_224cus.txt 2>&1; echo "kbench224 rc=$?"; cat $OUT/kbench_224cus.txt
