#!/bin/bash
# round 4, call 12: where does the TBT tail of whole-chip prefill (P100 / D100: TTFT 25 ms, TBT p50 5.0, p99 27-30 ms) come from?
# (a) host marks of every decode step against the prefill batches; (b) kernel traces of both processes on a short run
OUT=gpurun_out/r04_s12; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
COMMON="--steps 1 --warmup 1 --rate-sweep  --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave --no-kernel-timing"
for pol in "none 100 100" "dynamic 80 100"; do set -- $pol
  rm -rf /tmp/marks_$1; SEMIPD_TTFT_TRACE=/tmp/marks_$1 timeout 600 python bench.py --steps 1 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave --no-kernel-timing --cu-mask-mode $1 --prefill-cu $2 --decode-cu $3 > $OUT/marks_$1.json 2> $OUT/marks_$1.err
  echo "== $pol"; python tools/tbt_tail.py --trace /tmp/marks_$1 2>&1 | tee $OUT/marks_$1.txt | cut -c1-230 | head -30
done
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python $R/bench.py --steps 1 --warmup 0 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs --no-saturation-wave --no-kernel-timing --cu-mask-mode none --prefill-cu 100 --decode-cu 100 --num-requests 96 > $R/$OUT/ktrace.json 2> $R/$OUT/ktrace.err )
python tools/tbt_tail.py --reduce /tmp/ktrace $OUT/ktrace_p100_d100.npz
python tools/tbt_tail.py --kernels $OUT/ktrace_p100_d100.npz --top 10 2>&1 | tee $OUT/ktrace_report.txt | cut -c1-260
