#!/bin/bash
# round 6, call 22: the headline wave alone under rocprofv3 --kernel-trace (one trace per process), then tools/trace_overlap.py both
# ways: each instance's kernels split by whether the other instance was running (the table of profiles/r05_*_kernels_by_overlap_*)
OUT=gpurun_out/r06_s22; mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 \
    --no-static-split-wave --no-unified-wave --no-side-configs --no-saturation-wave --rate-sweep "" --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_headline_only.json 2> /tmp/ov.err )
echo "bench rc=$?"
python - <<PY
import glob, os, subprocess, sys
files = sorted(glob.glob("/tmp/prof_ov/**/*kernel_trace.csv", recursive=True), key=os.path.getsize, reverse=True)
kinds = {}
for f in files:
    fh = open(f, errors="ignore")
    fh.seek(os.path.getsize(f) * 2 // 3)            # the serving phase, not the start-up (graph capture, GEMM timing)
    head = fh.read(8_000_000)
    n_stream, n_big = head.count("stream_gemm_glds"), head.count("Cijk") + head.count("gemm8p") + head.count("gemm4w")
    kind = "decode" if n_stream > 4 * n_big else ("prefill" if n_big > 0 else "other")
    print(os.path.getsize(f) >> 20, "MB", kind, n_stream, n_big, f)
    kinds.setdefault(kind, f)
d, p = kinds.get("decode"), kinds.get("prefill")
if d and p:
    for a, b, name in ((d, p, "decode"), (p, d, "prefill")):
        out = subprocess.run([sys.executable, "tools/trace_overlap.py", a, b], capture_output=True, text=True).stdout
        open(f"$OUT/{name}_kernels_by_overlap.txt", "w").write(out)
        print("\n".join(l[:230] for l in out.splitlines()[:16]))
PY
