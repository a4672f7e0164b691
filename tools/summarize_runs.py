"""One line per bench JSON: python tools/summarize_runs.py <dir>/*.json  (TTFT / TBT percentiles, saturation, stream-GEMM fraction,
prefill batch time, the deadline gate's holds)."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f.split("/")[-1], "unreadable:", e)
        continue
    s = d.get("saturation") or {}
    e = d.get("roofline_extra") or {}
    pb = e.get("prefill_batch_ms") or {}
    ds = e.get("decode_step_ms") or {}
    g = pb.get("step_gate") or {}
    print(f"{f.split('/')[-1]:34s} {d['value']:7.0f} tok/s TTFT {d['p50_ttft_ms']:5.1f}/{d['p99_ttft_ms']:6.1f} TBT {d['p50_tbt_ms']:5.2f}/"
          f"{d['p99_tbt_ms']:5.2f} sat {s.get('output_tok_s', 0):6.0f} frac {(d.get('roofline') or {}).get('frac', 0):.3f} "
          f"P batch {pb.get('forward_and_sync', 0):5.1f} ms x{pb.get('batches', 0)} ({pb.get('avg_tokens', 0):.0f} tok, full {pb.get('batches_on_full', 0)}) "
          f"D step {ds.get('forward_and_sync', 0) + ds.get('output', 0):.2f} ms gate holds {g.get('holds', '-')}/{g.get('gates', '-')} "
          f"held {g.get('held_ms', '-')} ms")
