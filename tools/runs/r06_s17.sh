#!/bin/bash
# round 6, call 17: smoke() with the one-launch decode form and the wide streaming GEMM, then the whole GPU suite on the final tree
OUT=gpurun_out/r06_s17; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/smoke.txt
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "gpu suite rc=$? in $(( $(date +%s) - T0 )) s"; tail -3 $OUT/pytest_gpu.txt | cut -c1-300
