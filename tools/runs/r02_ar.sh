timeout 600 python -m pytest tests/test_gpu_all_reduce.py -q -x -k "op_set" 2>&1 | tail -40
