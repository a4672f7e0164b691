"""Per-kernel duration percentiles of one process's rocprofv3 kernel trace, split by whether a kernel of ANOTHER process
(the other Semi-PD instance) was running at the same time.

    python tools/trace_overlap.py <decode_kernel_trace.csv> <prefill_kernel_trace.csv> [name-substring ...]

Both traces carry the same clock (rocprofv3 timestamps are system nanoseconds), so a launch of the first trace is
"overlapped" when at least half of its duration lies inside busy intervals of the second.  Prints, per kernel name
(template arguments kept up to 70 characters): launches, mean, p10 / p50 / p90 / p99 in microseconds -- all, alone,
overlapped -- and the share of the kernel's total time spent in launches slower than 3x its alone-median (the tail that
waits for workgroup slots behind the other instance's resident workgroups)."""
import csv
import sys

import numpy as np


def read_trace(path, want=None):
    names, start, end = [], [], []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            n = r.get("Kernel_Name") or r.get("Name") or ""
            if want and not any(w in n for w in want):
                continue
            names.append(n)
            start.append(int(r["Start_Timestamp"]))
            end.append(int(r["End_Timestamp"]))
    return names, np.asarray(start, dtype=np.int64), np.asarray(end, dtype=np.int64)


def busy_union(start, end):
    """Sorted, merged busy intervals."""
    order = np.argsort(start)
    s, e = start[order], end[order]
    out_s, out_e = [], []
    cs, ce = None, None
    for a, b in zip(s.tolist(), e.tolist()):
        if cs is None:
            cs, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            out_s.append(cs), out_e.append(ce)
            cs, ce = a, b
    if cs is not None:
        out_s.append(cs), out_e.append(ce)
    return np.asarray(out_s, dtype=np.int64), np.asarray(out_e, dtype=np.int64)


def overlap_len(s, e, bs, be, cum):
    """Length of [s, e) covered by the merged intervals (bs, be); cum = prefix sums of their lengths."""
    if len(bs) == 0:
        return np.zeros_like(s)
    i0 = np.searchsorted(be, s, side="right")       # first interval ending after s
    i1 = np.searchsorted(bs, e, side="left")        # first interval starting at or after e
    total = np.where(i1 > i0, cum[np.minimum(i1, len(cum) - 1)] - cum[np.minimum(i0, len(cum) - 1)], 0)
    # clip the partial first / last intervals
    first_clip = np.where(i1 > i0, np.maximum(0, s - bs[np.minimum(i0, len(bs) - 1)]), 0)
    last_clip = np.where(i1 > i0, np.maximum(0, be[np.minimum(np.maximum(i1 - 1, 0), len(be) - 1)] - e), 0)
    return total - first_clip - last_clip


def pct(a):
    if len(a) == 0:
        return "      -      -      -      -      -"
    return "%7.1f %6.1f %6.1f %6.1f %6.1f" % (a.mean(), *np.percentile(a, [10, 50, 90, 99]))


def short(name):
    n = name.replace("void ", "").replace("semipd::", "")
    cut = n.find("(")
    if cut > 0:
        n = n[:cut]
    return n[:70]


def main():
    a_path, b_path = sys.argv[1], sys.argv[2]
    want = sys.argv[3:] or None
    names, s, e = read_trace(a_path, want)
    _, os_, oe = read_trace(b_path)
    # the other process's long kernels only: a 5 us elementwise launch holds no CU for long
    keep = (oe - os_) >= 20000
    bs, be = busy_union(os_[keep], oe[keep])
    cum = np.concatenate([[0], np.cumsum(be - bs)])
    dur = (e - s).astype(np.float64)
    ov = overlap_len(s, e, bs, be, cum) / np.maximum(dur, 1.0)
    by = {}
    for i, n in enumerate(names):
        by.setdefault(short(n), []).append(i)
    span = (max(e.max(), oe.max()) - min(s.min(), os_.min())) / 1e9 if len(s) and len(os_) else 0.0
    print(f"# {a_path} vs {b_path}: other instance busy (kernels >= 20 us) {(be - bs).sum() / 1e9:.2f} s of {span:.2f} s")
    print("# kernel | launches | all: mean p10 p50 p90 p99 us | alone (< 10 % overlapped) | overlapped (>= 50 %) | share of launches "
          "overlapped | share of time in launches > 3x alone-median")
    rows = sorted(by.items(), key=lambda kv: -dur[kv[1]].sum())
    for n, idx in rows[:14]:
        idx = np.asarray(idx)
        d = dur[idx] / 1e3
        alone = d[ov[idx] < 0.1]
        over = d[ov[idx] >= 0.5]
        med = np.median(alone) if len(alone) else np.median(d)
        tail = d[d > 3 * med].sum() / d.sum()
        print(f"{n:70s} | {len(d):7d} | {pct(d)} | {pct(alone)} | {pct(over)} | {len(over) / len(d):5.2f} | {tail:5.2f}")


if __name__ == "__main__":
    main()
