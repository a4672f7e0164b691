#!/bin/bash
# round 3: prefill gate_up GEMM with the SiLU epilogue (gemm8p) instead of tuned library GEMM + silu_and_mul, in situ
OUT=gpurun_out/r03_fused_gate_up; mkdir -p $OUT
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 2 --warmup 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'])"
}
run lib SEMIPD_GATE_UP_FUSED_ROWS=0
run fused256 SEMIPD_GATE_UP_FUSED_ROWS=4096
run fused128x512 SEMIPD_GATE_UP_FUSED_ROWS=4096 SEMIPD_G8_XH=64
