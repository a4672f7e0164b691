for f in 1 2 0; do SEMIPD_EXTEND_KV_FORM=$f timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "random_batches" 2>&1 | tail -3; done
