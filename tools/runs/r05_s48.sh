#!/bin/bash
# round 5, call 48: the tree as committed (planes form off by default): the tiled-GEMM tests, the layer routing test, the Semi-PD == unified engine test, smoke()
OUT=gpurun_out/r05_s48; mkdir -p $OUT
timeout 150 python -m pytest tests/test_gpu_ops.py -q -k "gemm_tall or defers or tall_planes" > $OUT/pytest_tall.txt 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_tall.txt | cut -c1-200
timeout 120 python -m pytest tests/test_gpu_engine.py -q -x -k "chunked" > $OUT/pytest_engine.txt 2>&1; echo "pytest engine rc=$?"; tail -2 $OUT/pytest_engine.txt | cut -c1-200
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
