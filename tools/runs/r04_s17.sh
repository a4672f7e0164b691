#!/bin/bash
# round 4, call 17: rotary pairs with pinned roundings (one shared function): the fused MLA prep against its launches, every rope test, engine tests
OUT=gpurun_out/r04_s17; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mla_prep.py -q > $OUT/pytest_mla_prep.txt 2>&1; echo "pytest mla prep rc=$?"
tail -6 $OUT/pytest_mla_prep.txt | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8_kv.py tests/test_gpu_torch_ops.py -q -x -k "rope or rotary or planes" > $OUT/pytest_rope.txt 2>&1; echo "pytest rope rc=$?"
tail -3 $OUT/pytest_rope.txt | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_deepseek.py tests/test_gpu_full_width.py tests/test_gpu_engine.py -q -x > $OUT/pytest_engines.txt 2>&1; echo "pytest engines rc=$?"
tail -3 $OUT/pytest_engines.txt | cut -c1-220
