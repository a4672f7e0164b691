O=gpurun_out/r02_splits; mkdir -p $O
for s in 0 4 8 16; do
  a=""; [ $s != 0 ] && a="--kv-splits $s"
  timeout 600 python bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave $a > $O/bench_s$s.json 2> $O/err_s$s.txt
  python - "$O/bench_s$s.json" "$s" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
x = d["roofline_extra"]
print("splits", sys.argv[2], "tok/s", d["value"], "ttft", round(d["p50_ttft_ms"], 1), round(d["p99_ttft_ms"], 1), "tbt", round(d["p50_tbt_ms"], 2), round(d["p99_tbt_ms"], 2),
      "decode_attention", {k: x.get("decode_attention", {}).get(k) for k in ("achieved", "avg_launch_us")}, "step", x.get("decode_step_ms"))
PY
done
