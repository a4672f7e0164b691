#!/bin/bash
# round 4, call 2: per-CU HBM probe; dynamic CU shares (masked streams) vs the nested HSA_CU_MASK points of call 1
OUT=gpurun_out/r04_s2; mkdir -p $OUT
timeout 300 tools/hbm_cu_probe > $OUT/hbm_cu_probe.txt 2>&1; echo "probe rc=$?"
run() { # name, args...
  name=$1; shift
  timeout 600 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
run dyn_p81_d100 --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 100
run dyn_p81_d100_backlog --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 100 --prefill-backlog-full-tokens 8192
run dyn_p75_d100_backlog --cu-mask-mode dynamic --prefill-cu 75 --decode-cu 100 --prefill-backlog-full-tokens 8192
run dyn_p62_d38 --cu-mask-mode dynamic --prefill-cu 62 --decode-cu 38
run env_p81_d100 --prefill-cu 81 --decode-cu 100
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), s.get("p50_tbt_ms"), "frac", (d.get("roofline") or {}).get("frac"), d["roofline_extra"].get("prefill_batch_ms"), d["roofline_extra"].get("decode_step_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $OUT/dyn_p81_d100.err | cut -c1-400
