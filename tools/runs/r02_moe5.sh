timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "moe or fused_experts" 2>&1 | tail -3
timeout 300 python tools/kbench_moe_stages.py 2>&1 | grep "T="
