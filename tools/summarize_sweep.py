"""Tabulates the qps_sweep objects of bench.py lines: python tools/summarize_sweep.py a.json b.json ..."""
import json
import sys


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{path}: no JSON line")


def main():
    for path in sys.argv[1:]:
        d = last_json_line(path)
        c = d["config"]
        print(f"## {path.split('/')[-1]}")
        print(f"   {c['workload']}")
        print("   rate req/s | output tok/s | TTFT p50 / p99 ms | TBT p50 / p99 ms")
        rows = list(d.get("qps_sweep", []))
        if d.get("steps") and d.get("p50_ttft_ms") is not None:   # (--steps 0: a sweep-only invocation has no timed step)
            rows.append({"request_rate": c["request_rate"], "output_tok_s": d["value"], "p50_ttft_ms": d["p50_ttft_ms"],
                         "p99_ttft_ms": d["p99_ttft_ms"], "p50_tbt_ms": d["p50_tbt_ms"], "p99_tbt_ms": d["p99_tbt_ms"],
                         "_timed": True})
        for r in sorted(rows, key=lambda r: r["request_rate"]):
            print(f"   {r['request_rate']:10.0f} | {r['output_tok_s']:12.1f} | {r['p50_ttft_ms']:8.1f} / {r['p99_ttft_ms']:8.1f} | "
                  f"{r['p50_tbt_ms']:6.2f} / {r['p99_tbt_ms']:6.2f}" + ("   (the timed step)" if r.get("_timed") else ""))
        print()


if __name__ == "__main__":
    main()
