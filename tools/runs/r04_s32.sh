#!/bin/bash
# round 4, call 32: tail split in the tiled GEMM (gemm8p): correctness, kbench on the 208-CU share, the start-up table and serving A / B
OUT=gpurun_out/r04_s32; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm_tall" > $OUT/pytest_tall.txt 2>&1; echo "pytest tall rc=$?"
tail -4 $OUT/pytest_tall.txt | cut -c1-220
for t in 1 0; do
  echo "== SEMIPD_G8_TAIL_SPLIT=$t"
  SEMIPD_G8_TAIL_SPLIT=$t HSA_CU_MASK=0:0-207 KBENCH_NUM_CUS=208 KBENCH_MS=1024,1536,2048 timeout 600 python tools/kbench.py gemm_tall 2>&1 | grep -v "Warning\|amdgpu.ids" | grep -v "128256"
done | tee $OUT/gemm_tall_208cus_tail.txt
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run warm --num-requests 16 --no-saturation-wave
SEMIPD_G8_TAIL_SPLIT=1 run tail_on
SEMIPD_G8_TAIL_SPLIT=0 run tail_off
SEMIPD_G8_TAIL_SPLIT=1 run tail_on_2
grep -h "rows=" $OUT/tail_on.err | grep "silu" | grep "tiled$\|<- tiled" | cut -c1-140 > $OUT/tall_table.txt; grep -h "silu" $OUT/tail_on.err | grep "rows=1024\|rows=2048\|rows=1536" | cut -c1-140
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/tail_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        e = d["roofline_extra"]; pb = e.get("prefill_batch_ms") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), "frac", (d.get("roofline") or {}).get("frac"), "P", pb.get("batches"), pb.get("avg_tokens"), pb.get("forward_and_sync"))
    except Exception as e:
        print(f, "failed", e)
PY
