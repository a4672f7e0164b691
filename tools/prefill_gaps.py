"""Idle time between the kernels of a prefill batch, from a rocprofv3 --kernel-trace CSV of the prefill process:
python tools/prefill_gaps.py <dir with *_kernel_trace.csv>.  A batch = a run of kernels without a gap above 1 ms that
contains at least 200 kernels; prints per-batch span, busy time, idle time and the gap histogram."""
import csv
import glob
import sys
from collections import Counter

import numpy as np


def main(d):
    for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
        rows = []
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        if len(rows) < 1000:
            continue
        rows.sort()
        names = Counter(n.split("(")[0][:50] for _, _, n in rows)
        if not any("extend_attn" in n for n in names):
            continue                      # not the prefill process
        batches, cur = [], [rows[0]]
        for a, b in zip(rows, rows[1:]):
            if b[0] - a[1] > 1_000_000:
                batches.append(cur)
                cur = []
            cur.append(b)
        batches.append(cur)
        batches = [b for b in batches if 250 <= len(b) <= 400]      # one 1024-token request of a 32-layer model
        print(f"{f}: {len(rows)} kernels, {len(batches)} single-request batches")
        spans, busy, idle, gaps_all = [], [], [], []
        for b in batches:
            span = b[-1][1] - b[0][0]
            bz = sum(e - s for s, e, _ in b)
            gaps = [max(0, y[0] - x[1]) for x, y in zip(b, b[1:])]
            spans.append(span / 1e6); busy.append(bz / 1e6); idle.append(sum(gaps) / 1e6); gaps_all += gaps
        if not batches:
            continue
        g = np.array(gaps_all) / 1e3
        print(f"  kernels per batch {np.mean([len(b) for b in batches]):.0f}; span p50 {np.percentile(spans, 50):.2f} ms, "
              f"busy p50 {np.percentile(busy, 50):.2f} ms, idle p50 {np.percentile(idle, 50):.2f} ms")
        print(f"  gap between consecutive kernels: p50 {np.percentile(g, 50):.2f} us, p90 {np.percentile(g, 90):.2f}, "
              f"p99 {np.percentile(g, 99):.2f}, mean {g.mean():.2f}")
        # where the busy time goes, by kernel family, for the median batch
        b = batches[len(batches) // 2]
        fam = Counter()
        for s, e, n in b:
            k = "hipBLASLt" if n.startswith(("Cijk", "Custom_Cijk")) else n.replace("void semipd::", "").split("<")[0].split("(")[0][:40]
            fam[k] += (e - s) / 1e3
        print("  median batch, us by family:", ", ".join(f"{k} {v:.0f}" for k, v in fam.most_common(10)))


if __name__ == "__main__":
    main(sys.argv[1])
