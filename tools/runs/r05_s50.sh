#!/bin/bash
# round 5, call 50: dense_linear's tiled-GEMM choice through the shared predicate (_takes_tiled_gemm): the layer tests and one engine on prefill batches of 65+ rows
OUT=gpurun_out/r05_s50; mkdir -p $OUT
timeout 60 python -m pytest tests/test_gpu_ops.py -q -k "defers or tall_planes_must" > $OUT/pytest_a.txt 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest_a.txt | cut -c1-200
timeout 100 python -m pytest tests/test_gpu_engine.py -q -x -k "chunked or logprobs" > $OUT/pytest_b.txt 2>&1; echo "pytest engine rc=$?"; tail -1 $OUT/pytest_b.txt | cut -c1-200
