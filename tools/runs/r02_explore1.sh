#!/bin/bash
# round 2, GPU exploration 1: does TENSILE_STREAMK_MAX_CUS make hipBLASLt follow a CU mask, and which P/D split pays
OUT=gpurun_out/r02_explore1; mkdir -p $OUT
for cfg in "-:-" "0:0-127:-" "0:0-127:128" "0:0-191:-" "0:0-191:192" "0:0-63:64"; do
  mask=${cfg%:*}; mc=${cfg##*:}
  ( [ "$mask" != "-" ] && export HSA_CU_MASK=$mask; [ "$mc" != "-" ] && export TENSILE_STREAMK_MAX_CUS=$mc;
    timeout 300 python tools/kbench.py linear_prefill ) >> $OUT/linear_prefill.txt 2>&1
done
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; tail -c 600 $OUT/bench_$name.json | head -c 10 >/dev/null; }
run p50_d50
run p50_d50_lib --library-gemm-grid
run p75_d25_lib --prefill-cu 75 --decode-cu 25 --library-gemm-grid
run p62_d38_lib --prefill-cu 62 --decode-cu 38 --library-gemm-grid
run p75_d100_lib --prefill-cu 75 --decode-cu 100 --library-gemm-grid
run p88_d100_lib --prefill-cu 88 --decode-cu 100 --library-gemm-grid
run p100_d100 --prefill-cu 100 --decode-cu 100
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_explore1/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1][:-5].ljust(16), d["value"], "ttft", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "tbt", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2))
    except Exception as e:
        print(f, "ERR", e)
PY
