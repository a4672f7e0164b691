timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py -q -x -k "decode" 2>&1 | tail -4
