#!/bin/bash
# round 4, call 13: DISJOINT shares with a large prefill share (the decode instance keeps a small exclusive set and takes the chip
# while the prefill instance idles), and whole-chip prefill in small chunks (time slicing by kernel boundaries)
OUT=gpurun_out/r04_s13; mkdir -p $OUT
run() { name=$1; shift; timeout 400 python bench.py --steps 1 --warmup 1 --rate-sweep "" --no-static-split-wave --no-cpu-baseline --no-side-configs "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run dyn_p80_d100
run dyn_p81_d19 --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 19
run dyn_p75_d25 --cu-mask-mode dynamic --prefill-cu 75 --decode-cu 25
run dyn_p88_d12 --cu-mask-mode dynamic --prefill-cu 88 --decode-cu 12
run dyn_p81_d31 --cu-mask-mode dynamic --prefill-cu 81 --decode-cu 31
run env_p81_d19 --cu-mask-mode env --prefill-cu 81 --decode-cu 19
run none_chunk512 --cu-mask-mode none --prefill-cu 100 --decode-cu 100 --chunked-prefill-size 512
run none_chunk1024 --cu-mask-mode none --prefill-cu 100 --decode-cu 100 --chunked-prefill-size 1024
run dyn_p80_d100_chunk1024 --chunked-prefill-size 1024
run dyn_p88_d100_chunk1024 --prefill-cu 88 --chunked-prefill-size 1024
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d.get("saturation") or {}
        e = d["roofline_extra"]
        pb, ds = e.get("prefill_batch_ms") or {}, e.get("decode_step_ms") or {}
        print(f.split("/")[-1], d["value"], "TTFT", round(d["p50_ttft_ms"],1), round(d["p99_ttft_ms"],1), "TBT", round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2),
              "sat", s.get("output_tok_s"), s.get("p50_tbt_ms"), "frac", (d.get("roofline") or {}).get("frac"),
              "P", pb.get("batches"), pb.get("avg_tokens"), pb.get("forward_and_sync"), pb.get("batches_on_full"), "D", ds.get("steps"), ds.get("output"), ds.get("steps_on_share"), ds.get("steps_on_full"))
    except Exception as e:
        print(f, "failed", e)
PY
