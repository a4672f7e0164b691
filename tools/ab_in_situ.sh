#!/bin/bash
# A/B of an environment knob where it counts: the headline wave of bench.py (Llama-3-8B, 256 requests in 1024 / out 128 at
# 32 req/s, default policy; 1 warm-up + 3 timed waves, no side engines) with both instances running, alternating values on
# ONE box.  Stand-alone kernel timings do not predict the serving run (the planes form of the tiled GEMM is
# 4-12 us per layer shorter alone and costs 1.5 ms of TTFT p50 in situ), so a kernel change is judged by this.
#   usage (on the GPU box):  bash tools/ab_in_situ.sh SEMIPD_RMS_WIDE 1 0 1   [OUT_DIR]
# ~72 s for the first run (GEMM tuning), ~42 s for each further one.  Reads: TTFT p50 / p99, TBT p50 / p99, the stream-GEMM
# fraction, the prefill batch time and the deadline holds of each run (tools/summarize_runs.py).
KNOB=$1; shift
VALS=()
while [ $# -gt 0 ] && [[ "$1" != */* ]]; do VALS+=("$1"); shift; done
OUT=${1:-gpurun_out/ab_$KNOB}; mkdir -p $OUT
i=0
for v in "${VALS[@]}"; do
  i=$((i + 1)); T0=$(date +%s)
  env "$KNOB=$v" timeout 400 python bench.py --no-cpu-baseline --no-static-split-wave --no-unified-wave --no-saturation-wave \
      --no-side-configs --rate-sweep "" --steps 3 --warmup 1 > $OUT/run${i}_$v.json 2> $OUT/run${i}_$v.err
  echo "$KNOB=$v rc=$? in $(( $(date +%s) - T0 )) s"
  python tools/summarize_runs.py $OUT/run${i}_$v.json
done
