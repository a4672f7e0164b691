#!/bin/bash
OUT=gpurun_out/r02_mla; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "decode_attention or mla" 2>&1 | tail -5 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 300 python tools/kbench_mla_quick.py > $OUT/kbench_shared.txt 2>&1; cat $OUT/kbench_shared.txt
timeout 100 python tools/dbg_mla_l2.py
