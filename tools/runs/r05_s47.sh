#!/bin/bash
# round 5, call 47: is the wide RMSNorm (one 16-byte vector per lane; call 35 measured it on decode steps ALONE) also right next to the other
# instance?  The headline wave alone (1 warm-up + 3 timed waves), SEMIPD_RMS_WIDE = 1 / 0 / 1 in both instances, one box.
OUT=gpurun_out/r05_s47; mkdir -p $OUT
for v in 1 0 1; do
  T0=$(date +%s)
  SEMIPD_RMS_WIDE=$v timeout 300 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-saturation-wave --no-side-configs --rate-sweep "" --steps 3 --warmup 1 > $OUT/bench_wide_$v.json 2> $OUT/bench_wide_$v.err
  echo "SEMIPD_RMS_WIDE=$v rc=$? in $(( $(date +%s) - T0 )) s"
  python tools/summarize_runs.py $OUT/bench_wide_$v.json
  cp $OUT/bench_wide_$v.json $OUT/bench_wide_${v}_$T0.json
done
