#!/bin/bash
# round 3, call 11: the per-rank shapes of Llama-3-70B TP = 8 on one GPU (decode step against its 2.2 ms weight floor);
# the driver's N = 8 command on one GPU (functional dry run); rocprofv3 of the default bench + PMC of the tall GEMM
OUT=gpurun_out/r03_s11; mkdir -p $OUT
for pd in "100 100" "62 38"; do
  set -- $pd
  timeout 900 python bench.py --model llama3-70b-tp8-rank --prefill-cu $1 --decode-cu $2 --num-requests 64 --request-rate 8 --no-cpu-baseline --rate-sweep "" --no-static-split-wave --steps 1 --warmup 1 > $OUT/bench_70b_rank_p$1.json 2> $OUT/bench_70b_rank_p$1.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_70b_rank_p$1.json").read().strip().splitlines()[-1])
print("70b-tp8-rank P$1/D$2", d["value"], round(d["p50_ttft_ms"],1), round(d["p50_tbt_ms"],2), round(d["p99_tbt_ms"],2), d.get("saturation",{}).get("output_tok_s"), d.get("saturation",{}).get("p50_tbt_ms"), d["roofline_extra"].get("decode_step_ms"), (d.get("roofline") or {}).get("frac"))
PY
done 2>&1 | tee $OUT/bench_70b_rank.txt
bash tools/runs/r03_dryrun_n8.sh 2>&1 | tail -12 | cut -c1-1500
bash tools/runs/r03_s9.sh 2>&1 | tail -60 | cut -c1-400
