#!/bin/bash
# round 4, call 11: the prefill instance on a LOW-priority HIP stream (hipStreamCreateWithPriority), decode on the NULL stream:
# does the queue priority let whole-chip prefill coexist with a short TBT tail?  Same box: default policy as the baseline.
OUT=gpurun_out/r04_s11; mkdir -p $OUT
run() { name=$1; shift; timeout 700 python bench.py --steps 2 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; grep -h "priority" $OUT/$name.err | head -2 | cut -c1-160; }
run none_p100_d100_plow --cu-mask-mode none --prefill-cu 100 --decode-cu 100 --prefill-priority 1
run none_p100_d100 --cu-mask-mode none --prefill-cu 100 --decode-cu 100
run env_p88_d100_plow --cu-mask-mode env --prefill-cu 88 --decode-cu 100 --prefill-priority 1
run dyn_default
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
