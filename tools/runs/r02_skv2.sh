timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "extend" -x 2>&1 | tail -5
timeout 300 python tools/kbench_ext_quick.py 2>&1 | grep ABL
