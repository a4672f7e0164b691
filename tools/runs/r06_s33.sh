#!/bin/bash
# round 6, calls 15 / 16: rocprofv3 --kernel-trace --stats of the DEFAULT bench command, one stats file per process (no -o: the files carry
# the pid); the headline engine's two processes are the ones with the most launches of their kind.  Then the line itself, 3 timed steps.
OUT=gpurun_out/r06_s33; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
( cd /tmp && timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> /tmp/bench_under_rocprof.err )
echo "bench under rocprofv3 rc=$? in $(( $(date +%s) - T0 )) s"
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
    t = {x["Name"]: float(x["TotalDurationNs"]) for x in r}
    stream = sum(v for k, v in t.items() if "stream_gemm_glds" in k)
    big = sum(v for k, v in t.items() if "Cijk" in k or "gemm8p" in k or "gemm4w" in k)
    kind = "decode" if stream > big and stream > 0 else ("prefill" if big > 0 else "other")
    if stream > 0 and big > 0 and min(stream, big) > 0.25 * max(stream, big):
        kind = "unified"
    rows.append((calls, tot, kind, r[0]["Name"][:50], f))
rows.sort(key=lambda r: -r[1])      # by GPU time: the headline engine serves the most waves
for calls, tot, kind, top, f in rows:
    print(f"{tot/1e9:8.2f} s {calls:9d} launches {kind:8s} {top}  {os.path.relpath(f, '/tmp/prof_bench')}")
for kind in ("decode", "prefill"):
    best = [r for r in rows if r[2] == kind]
    if best:
        shutil.copy(best[0][4], f"$OUT/bench_n1_{kind}_process_kernel_stats.csv")
PY
python tools/stats_top.py $OUT/bench_n1_decode_process_kernel_stats.csv | head -14
python tools/stats_top.py $OUT/bench_n1_prefill_process_kernel_stats.csv | head -24
