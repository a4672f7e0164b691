for f in 1 2 0; do echo "== form $f"; SEMIPD_EXTEND_KV_FORM=$f timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "extend" -x 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_gpu_all_reduce.py -q -x 2>&1 | tail -5
