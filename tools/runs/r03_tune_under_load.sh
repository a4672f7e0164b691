#!/bin/bash
# round 3: prefill GEMM solutions timed next to a replaying decode step (SEMIPD_TUNE_UNDER_DECODE_LOAD) vs alone
OUT=gpurun_out/r03_tune_under_load; mkdir -p $OUT
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 3 --warmup 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'])"
  grep -A45 "library GEMM solutions timed" $OUT/bench_$name.err | grep -E "timed on|rows=1024|rows=2048" | head -12
}
run alone SEMIPD_TUNE_UNDER_DECODE_LOAD=0
run under_load SEMIPD_TUNE_UNDER_DECODE_LOAD=1
