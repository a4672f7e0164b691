timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "abort_of_a_failed" 2>&1 | tail -12
