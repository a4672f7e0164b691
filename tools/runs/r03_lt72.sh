#!/bin/bash
# round 3: the ROCm installation's hipBLASLt loaded beside PyTorch's copy (SEMIPD_HIPBLASLT_LIB) for the prefill GEMMs
OUT=gpurun_out/r03_lt72; mkdir -p $OUT
timeout 600 env SEMIPD_HIPBLASLT_LIB=/opt/rocm/lib/libhipblaslt.so.1 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "dense_gemm" -x > $OUT/pytest_dense_gemm_lt72.txt 2>&1; tail -3 $OUT/pytest_dense_gemm_lt72.txt
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 3 --warmup 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'])"
  grep -A45 "library GEMM solutions timed" $OUT/bench_$name.err | grep -E "timed on|library:|rows=1024|rows=2048" | head -12
  grep -i "semipd dense_gemm\|error\|Traceback" $OUT/bench_$name.err | head -5
}
run lt72 SEMIPD_HIPBLASLT_LIB=/opt/rocm/lib/libhipblaslt.so.1
run bundled SEMIPD_HIPBLASLT_LIB=0
