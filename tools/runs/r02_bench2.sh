#!/bin/bash
# round 2: isolation policy with the streaming linear + planes fusion + servicing wait: priorities, chunk size, masks
OUT=gpurun_out/r02_bench2; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_deepseek.py -x -q -m gpu -k "stream_linear or decode_attention or deepseek" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 600 python tools/kbench.py mla > $OUT/kbench_mla.txt 2>&1; grep -v amdgpu.ids $OUT/kbench_mla.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-saturation-wave --steps 1 --warmup 1 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run p100_d100 --prefill-cu 100 --decode-cu 100
run p100_d100_dprio --prefill-cu 100 --decode-cu 100 --decode-priority -1
run p100_d100_pprio --prefill-cu 100 --decode-cu 100 --prefill-priority -1
run p100_d100_chunk2k --prefill-cu 100 --decode-cu 100 --chunked-prefill-size 2048
run p50_d50
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_bench2/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1][:-5].ljust(20), d["value"], "ttft", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "tbt", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), d["roofline_extra"].get("prefill_batch_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
