#!/bin/bash
# round 2: serving bench with the LDS-DMA streaming linear in the decode path (A/B against hipBLASLt on the same box)
OUT=gpurun_out/r02_bench1; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -m gpu -k "stream_linear or engine or semi_pd or unified or opt" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run p50_d50
run p50_d50_blaslt --disable-stream-linear
run p100_d100 --prefill-cu 100 --decode-cu 100
run p100_d50 --prefill-cu 100 --decode-cu 50
run unified --mode unified
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_bench1/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1][:-5].ljust(16), d["value"], "ttft", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "tbt", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), d["roofline_extra"].get("decode_step_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
