timeout 600 python -m pytest tests/test_gpu_torch_ops.py -q -x 2>&1 | tail -25
