#!/bin/bash
# round 6, call 12: rocprofv3 --kernel-trace --stats of the DEFAULT bench command (every process), then the PMC passes of the fused decode
# launch (FETCH_SIZE, WRITE_SIZE in separate passes + a timing pass) on tools/pmc_target.py decode32_fused / decode32
OUT=gpurun_out/r06_s12; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
( cd /tmp && timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> /tmp/bench_under_rocprof.err )
echo "bench under rocprofv3 rc=$? in $(( $(date +%s) - T0 )) s"; tail -2 /tmp/bench_under_rocprof.err | cut -c1-200
python - <<PY
import csv, glob, os, shutil
files = glob.glob("/tmp/prof_bench/**/*kernel_stats.csv", recursive=True)
rows = []
for f in files:
    r = list(csv.DictReader(open(f)))
    if not r:
        continue
    calls = sum(int(x["Calls"]) for x in r)
    tot = sum(float(x["TotalDurationNs"]) for x in r)
    top = r[0]["Name"][:60]
    kind = "decode" if any("stream_gemm_glds" in x["Name"] for x in r[:2]) else ("prefill" if any(("Cijk" in x["Name"] or "gemm8p" in x["Name"] or "gemm4w" in x["Name"]) for x in r[:3]) else "other")
    rows.append((tot, calls, kind, top, f))
rows.sort(reverse=True)
for tot, calls, kind, top, f in rows:
    print(f"{tot/1e9:8.2f} s {calls:9d} launches {kind:8s} {top}  {os.path.basename(f)}")
for kind in ("decode", "prefill"):
    best = [r for r in rows if r[2] == kind]
    if best:
        shutil.copy(best[0][4], f"$OUT/bench_n1_{kind}_process_kernel_stats.csv")
PY
python tools/stats_top.py $OUT/bench_n1_decode_process_kernel_stats.csv | head -14
python tools/stats_top.py $OUT/bench_n1_prefill_process_kernel_stats.csv | head -22
for T in decode32_fused decode32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${T}_$C -o t -- python $GRAFT_REPO_ROOT/tools/pmc_target.py $T > /tmp/pmc_${T}_$C.log 2>&1 )
    echo "== $T $C: $(grep algorithmic /tmp/pmc_${T}_$C.log)" | tee -a $OUT/pmc_decode_fused.txt
    python tools/pmc_summary.py /tmp/pmc_${T}_$C decode_rope decode_mfma decode_stage2 rope_planes | tee -a $OUT/pmc_decode_fused.txt
  done
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_${T}_t -o t -- python $GRAFT_REPO_ROOT/tools/pmc_target.py $T > /dev/null 2>&1 )
  python tools/stats_top.py $(find /tmp/pmc_${T}_t -name "*kernel_stats.csv" | head -1) | grep -E "decode_|rope_planes" | tee -a $OUT/pmc_decode_fused.txt
done
