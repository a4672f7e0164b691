#!/bin/bash
# round 5, call 19: BASELINE config 2's Poisson QPS sweep as SURVEY 8d.2 spells it (bench_serving.py:1413-1438): lambda 2 .. 16 req/s, 512 requests per
# point, in = 1024 / out = 256 and in = 128 / out = 64, Llama-3-8B bf16 TP = 1, (a) the default policy, (b) the literal 50 / 50 split
OUT=gpurun_out/r05_s19; mkdir -p $OUT
COMMON="--steps 0 --warmup 0 --rate-sweep 2,4,6,8,10,12,14,16 --sweep-num-requests 512 --no-saturation-wave --no-static-split-wave --no-unified-wave --no-side-configs --no-cpu-baseline --no-kernel-timing"
case "$1" in
  a) timeout 1500 python bench.py $COMMON --input-len 1024 --output-len 256 --sweep-output-len 256 > $OUT/sweep_default_in1024_out256.json 2> $OUT/a.err; echo "a rc=$?";;
  b) timeout 1500 python bench.py $COMMON --input-len 128 --output-len 64 --sweep-output-len 64 > $OUT/sweep_default_in128_out64.json 2> $OUT/b.err; echo "b rc=$?";;
  c) timeout 1500 python bench.py $COMMON --input-len 1024 --output-len 256 --sweep-output-len 256 --prefill-cu 50 --decode-cu 50 --cu-mask-mode env --decode-step-deadline-ms 0 > $OUT/sweep_split5050_in1024_out256.json 2> $OUT/c.err; echo "c rc=$?";;
  d) timeout 1500 python bench.py $COMMON --input-len 128 --output-len 64 --sweep-output-len 64 --prefill-cu 50 --decode-cu 50 --cu-mask-mode env --decode-step-deadline-ms 0 > $OUT/sweep_split5050_in128_out64.json 2> $OUT/d.err; echo "d rc=$?";;
esac
python tools/summarize_sweep.py $OUT/*.json 2>/dev/null | tail -40
