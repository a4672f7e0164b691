O=gpurun_out/r02_moe; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "moe or fused" -x 2>&1 | tail -8
timeout 600 python tools/kbench.py moe 2>&1 | grep -v amdgpu | tee $O/kbench_moe_tiled.txt
SEMIPD_MOE_TILED=0 timeout 600 python tools/kbench.py moe 2>&1 | grep -v amdgpu | tee $O/kbench_moe_streaming.txt
