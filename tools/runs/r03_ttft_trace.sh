#!/bin/bash
# round 3: where the time to the first token goes at 8 and 32 req/s (default masks)
OUT=gpurun_out/r03_ttft_trace; mkdir -p $OUT
for rate in 8 32; do
  rm -rf /tmp/ttft_$rate
  SEMIPD_TTFT_TRACE=/tmp/ttft_$rate timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --request-rate $rate --steps 1 --warmup 1 > $OUT/bench_rate$rate.json 2> $OUT/bench_rate$rate.err
  python -c "
import json; d=json.loads(open('$OUT/bench_rate$rate.json').read().strip().splitlines()[-1]); print('rate $rate', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'])"
  python tools/ttft_trace.py /tmp/ttft_$rate | tee $OUT/hops_rate$rate.txt
done
