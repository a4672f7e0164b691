timeout 600 python -m pytest tests/test_gpu_engine.py -q -x -k "falls_back_to_eager or tp2 or two_tp" 2>&1 | tail -15
