timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "moe or fused_experts" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_deepseek.py -q -x 2>&1 | tail -3
timeout 300 python tools/kbench.py moe 2>&1 | grep "^moe"
