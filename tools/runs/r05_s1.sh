#!/bin/bash
# round 5, call 1: where the in-situ time of the decode kernels goes (kernel traces of BOTH instances, durations split by whether the
# other instance was busy), the decode GEMM kernel alone per K split / ring depth, the decode step as it stands
OUT=gpurun_out/r05_s1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-side-configs --rate-sweep "" --num-requests 16 --no-kernel-timing > /dev/null 2> $OUT/warm.err; echo "warm rc=$?"
( cd /tmp && export TMPDIR=/tmp SEMIPD_SHUTDOWN_JOIN_S=180 && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-side-configs --rate-sweep "" > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
echo "rocprof bench rc=$?"
for f in $(find /tmp/bench_prof -name "*kernel_stats.csv"); do n=$(grep -c "extend_attn" $f); m=$(grep -c "decode_mfma" $f); t=${f/kernel_stats/kernel_trace}; if [ "$n" -gt 0 ]; then cp $f $OUT/prefill_process_kernel_stats.csv; PT=$t; elif [ "$m" -gt 0 ]; then cp $f $OUT/decode_process_kernel_stats.csv; DT=$t; fi; done
ls -la $PT $DT
python tools/trace_overlap.py $DT $PT 2>&1 | cut -c1-330 | tee $OUT/decode_kernels_by_overlap.txt
python tools/trace_overlap.py $PT $DT 2>&1 | cut -c1-330 | tee $OUT/prefill_kernels_by_overlap.txt
python - <<PY
import json
d = json.loads(open("$OUT/bench_under_rocprof.json").read().strip().splitlines()[-1])
print("under rocprof:", d["value"], "TTFT", d["p50_ttft_ms"], "TBT", d["p50_tbt_ms"], d["p99_tbt_ms"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "avg_launch_us_minus_event_overhead", "launches_sampled")})
PY
for ring in 3 4; do
  SEMIPD_SL_RING=$ring KBENCH_NUM_CUS=256 KBENCH_MS=32 timeout 400 python tools/kbench.py stream_planes 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee $OUT/stream_planes_whole_chip.txt
HSA_CU_MASK=0:208-255 SEMIPD_SL_RING=3 KBENCH_NUM_CUS=256 KBENCH_MS=32 timeout 400 python tools/kbench.py stream_planes 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/stream_planes_48cus_declared256.txt
timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | cut -c1-140 | tee $OUT/steps.txt
timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 24 --ctx 1100 --steps 100 2>&1 | grep "ms per decode" | cut -c1-140 | tee -a $OUT/steps.txt
