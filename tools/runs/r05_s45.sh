#!/bin/bash
# round 5, call 45: the headline wave alone (1 warm-up + 3 timed waves, no side engines), the tiled GEMM's planes into the norm off / on / off, one box
OUT=gpurun_out/r05_s45; mkdir -p $OUT
for v in 0 1 0; do
  T0=$(date +%s)
  SEMIPD_TALL_PLANES=$v timeout 400 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-saturation-wave --no-side-configs --rate-sweep "" --steps 3 --warmup 1 > $OUT/bench_planes_$v.json 2> $OUT/bench_planes_$v.err
  echo "SEMIPD_TALL_PLANES=$v rc=$? in $(( $(date +%s) - T0 )) s"
  python tools/summarize_runs.py $OUT/bench_planes_$v.json
  cp $OUT/bench_planes_$v.json $OUT/bench_planes_${v}_$T0.json
done
