"""Per-hop percentiles of the time to the first token from the logs of SEMIPD_TTFT_TRACE=<dir>
(semi_pd_amd/semi_pd/ttft_trace.py): python tools/ttft_trace.py <dir>"""
import glob
import sys
from collections import defaultdict

import numpy as np

ORDER = ["client_send", "p_recv", "p_propose", "d_got_proposal", "p_admitted", "p_launched", "p_done", "d_got_result",
         "d_streamed", "client_first_token"]


def main(d):
    first = defaultdict(dict)   # rid -> event -> time; proposals repeat: the LAST p_propose / d_got_proposal before admission counts
    for f in glob.glob(d + "/*.log"):
        for line in open(f):
            parts = line.split()
            if len(parts) < 3:
                continue
            t, ev = float(parts[0]), parts[1]
            for rid in parts[2].split(","):
                if ev in ("p_propose", "d_got_proposal"):
                    first[rid].setdefault(ev + "_first", t)
                    first[rid].setdefault(ev + "_all", []).append(t)
                else:
                    first[rid].setdefault(ev, t)
    rows = []
    for rid, e in first.items():
        if "client_first_token" not in e or "p_admitted" not in e:
            continue
        for ev in ("p_propose", "d_got_proposal"):
            ts = [t for t in e.get(ev + "_all", []) if t <= e["p_admitted"]]
            if ts:
                e[ev] = max(ts)
        if all(k in e for k in ORDER):
            rows.append([e[k] for k in ORDER])
    a = np.array(rows)
    print(f"{len(a)} requests with a complete trace")
    if not len(a):
        return
    hops = np.diff(a, axis=1) * 1e3
    for i in range(hops.shape[1]):
        h = hops[:, i]
        print(f"  {ORDER[i]:>16} -> {ORDER[i + 1]:<20} mean {h.mean():7.2f}  p50 {np.percentile(h, 50):7.2f}  "
              f"p90 {np.percentile(h, 90):7.2f} ms")
    tot = (a[:, -1] - a[:, 0]) * 1e3
    print(f"  total (client_send -> client_first_token): mean {tot.mean():.2f}  p50 {np.percentile(tot, 50):.2f} ms")
    q = (a[:, 2] - a[:, 1]) * 1e3
    print(f"  of which queued in the prefill instance before the admitting proposal: mean {q.mean():.2f} p50 "
          f"{np.percentile(q, 50):.2f} ms;  on the GPU (p_launched -> p_done): mean {hops[:, 5].mean():.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
