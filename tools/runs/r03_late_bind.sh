#!/bin/bash
# round 3: late-binding prefill pipeline: off / lead 2 / 3 / 5 ms at 32 req/s (default masks), with the hop trace
OUT=gpurun_out/r03_late_bind; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  rm -rf /tmp/ttft_$name
  env "$@" SEMIPD_TTFT_TRACE=/tmp/ttft_$name timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 2 --warmup 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'], d['config'].get('prefill_batches'))"
  python tools/ttft_trace.py /tmp/ttft_$name > $OUT/hops_$name.txt; tail -4 $OUT/hops_$name.txt
}
run off SEMIPD_PREFILL_LATE_BIND=0
run lead3 SEMIPD_PREFILL_LEAD_MS=3
run lead5 SEMIPD_PREFILL_LEAD_MS=5
run lead2 SEMIPD_PREFILL_LEAD_MS=2
