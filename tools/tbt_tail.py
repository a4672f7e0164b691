"""Where does the TBT tail of a policy come from?  Two inputs, both optional:

  --trace DIR     SEMIPD_TTFT_TRACE logs of a bench run: decode-step completion times (d_step_done) against the prefill
                  instance's batch intervals (p_launched .. p_done): how long are the steps that complete inside / outside a
                  prefill batch, and which batches hold the slow ones.
  --kernels DIR   rocprofv3 --kernel-trace CSVs of the same kind of run: the decode process's kernels grouped into steps; for
                  the slowest steps, which kernels were long (against their own median) and what the prefill process ran then.
  --reduce DIR OUT.npz   (on the GPU box) shrink the CSVs to start / end / name-id arrays so that they fit gpurun_out/.
"""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np


def load_marks(d):
    ev = defaultdict(list)
    for f in glob.glob(os.path.join(d, "*.log")):
        for line in open(f):
            parts = line.split()
            if len(parts) >= 2:
                ev[parts[1]].append((float(parts[0]), parts[2] if len(parts) > 2 else ""))
    for k in ev:
        ev[k].sort()
    return ev


def trace_report(d, slow_ms):
    ev = load_marks(d)
    steps = np.array([t for t, _ in ev["d_step_done"]])
    if len(steps) < 3:
        print("no d_step_done marks")
        return
    dt = np.diff(steps) * 1e3
    launched = np.array([t for t, _ in ev["p_launched"]])
    done = np.array([t for t, _ in ev["p_done"]])
    n = min(len(launched), len(done))
    launched, done = launched[:n], done[:n]
    ends = steps[1:]
    starts = steps[:-1]
    inside = np.zeros(len(dt), bool)
    frac_overlap = np.zeros(len(dt))
    for i, (a, b) in enumerate(zip(starts, ends)):
        j0 = np.searchsorted(done, a)
        ov = 0.0
        for j in range(j0, n):
            if launched[j] >= b:
                break
            ov += max(0.0, min(b, done[j]) - max(a, launched[j]))
        frac_overlap[i] = ov / max(b - a, 1e-9)
        inside[i] = ov > 0
    pct = lambda x, q: float(np.percentile(x, q)) if len(x) else float("nan")
    print(f"{len(dt)} step intervals: p50 {pct(dt, 50):.2f} p90 {pct(dt, 90):.2f} p99 {pct(dt, 99):.2f} max {dt.max():.2f} ms; "
          f"prefill batches {n}, mean {np.mean(done - launched) * 1e3:.1f} ms, busy {np.sum(done - launched) / (steps[-1] - steps[0]):.2f}")
    for name, m in (("overlapping a prefill batch", inside), ("no prefill batch in flight", ~inside)):
        x = dt[m]
        print(f"  {name}: {len(x)} steps, p50 {pct(x, 50):.2f} p90 {pct(x, 90):.2f} p99 {pct(x, 99):.2f} max {x.max() if len(x) else 0:.2f}")
    slow = np.where(dt > slow_ms)[0]
    print(f"  {len(slow)} intervals above {slow_ms} ms ({100.0 * len(slow) / len(dt):.1f} %)")
    bs = [int(b) if b else 0 for _, b in ev["d_step_done"]][1:]
    for i in slow[:40]:
        j = np.searchsorted(done, starts[i])
        rel = (starts[i] - launched[j]) * 1e3 if j < n else float("nan")
        blen = (done[j] - launched[j]) * 1e3 if j < n else float("nan")
        print(f"    step {i}: {dt[i]:.1f} ms, batch {bs[i]}, overlap {frac_overlap[i]:.2f}; began {rel:+.1f} ms after prefill batch {j} "
              f"was launched (that batch: {blen:.1f} ms); previous / next interval {dt[i - 1] if i else 0:.1f} / {dt[i + 1] if i + 1 < len(dt) else 0:.1f}")


def find_csvs(d):
    return sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))


def reduce_csvs(d, out):
    procs = {}
    names = {}
    for f in find_csvs(d):
        st, en, ni = [], [], []
        with open(f) as fh:
            r = csv.DictReader(fh)
            for row in r:
                nm = row.get("Kernel_Name") or row.get("Name") or ""
                nm = nm.replace("void ", "").replace("semipd::", "")[:60]
                st.append(int(row["Start_Timestamp"]))
                en.append(int(row["End_Timestamp"]))
                ni.append(names.setdefault(nm, len(names)))
        if st:
            key = os.path.basename(f).split("_")[0]
            procs[key] = (np.array(st), np.array(en), np.array(ni, np.int32))
    arrs = {}
    for k, (st, en, ni) in procs.items():
        o = np.argsort(st)
        arrs[f"{k}_start"], arrs[f"{k}_end"], arrs[f"{k}_name"] = st[o], en[o], ni[o]
    arrs["names"] = np.array(sorted(names, key=names.get))
    np.savez_compressed(out, **arrs)
    print("reduced", {k: len(v[0]) for k, v in procs.items()}, "->", out)


def kernels_report(npz, top):
    z = np.load(npz, allow_pickle=False)
    names = [str(x) for x in z["names"]]
    procs = sorted({k.rsplit("_", 1)[0] for k in z.files if k != "names"})
    info = {}
    for p in procs:
        nm = z[p + "_name"]
        cnt = np.bincount(nm, minlength=len(names))
        info[p] = cnt
    dec = max(procs, key=lambda p: sum(info[p][i] for i, n in enumerate(names) if "decode_mfma" in n or "decode_stage" in n))
    pre = max(procs, key=lambda p: sum(info[p][i] for i, n in enumerate(names) if "extend_attn" in n))
    print("decode process", dec, "prefill process", pre)
    ds, de, dn = z[dec + "_start"], z[dec + "_end"], z[dec + "_name"]
    ps, pe, pn = z[pre + "_start"], z[pre + "_end"], z[pre + "_name"]
    # steps: split at the sampling kernel (argmax) of each step
    am = [i for i, n in enumerate(names) if "argmax" in n]
    is_last = np.isin(dn, am)
    bounds = np.where(is_last)[0]
    med = {}
    dur = (de - ds) / 1e3
    for i in np.unique(dn):
        med[i] = float(np.median(dur[dn == i]))
    steps = []
    a = 0
    for b in bounds:
        if b - a > 50:
            steps.append((a, b + 1))
        a = b + 1
    sd = np.array([(de[b - 1] - ds[a]) / 1e6 for a, b in steps])
    print(f"{len(steps)} decode steps: p50 {np.percentile(sd, 50):.2f} p90 {np.percentile(sd, 90):.2f} p99 {np.percentile(sd, 99):.2f} max {sd.max():.2f} ms")
    order = np.argsort(-sd)[:top]
    for k in order:
        a, b = steps[k]
        t0, t1 = ds[a], de[b - 1]
        busy = float(np.sum(de[a:b] - ds[a:b])) / 1e6
        gaps = (ds[a + 1:b] - de[a:b - 1]) / 1e3
        excess = defaultdict(float)
        for i in range(a, b):
            excess[dn[i]] += dur[i] - med[dn[i]]
        worst = sorted(excess.items(), key=lambda kv: -kv[1])[:4]
        j0, j1 = np.searchsorted(pe, t0), np.searchsorted(ps, t1)
        pk = defaultdict(float)
        for j in range(j0, j1):
            pk[pn[j]] += (min(pe[j], t1) - max(ps[j], t0)) / 1e6
        ptop = sorted(pk.items(), key=lambda kv: -kv[1])[:4]
        print(f"  step {k}: {sd[k]:.2f} ms = kernels {busy:.2f} + gaps {float(np.sum(np.maximum(gaps, 0))) / 1e3:.2f} (largest gap {gaps.max():.0f} us); "
              f"excess over medians: " + ", ".join(f"{names[i][:28]} +{v / 1e3:.2f} ms" for i, v in worst))
        print("      prefill meanwhile: " + (", ".join(f"{names[i][:36]} {v:.2f} ms" for i, v in ptop) or "nothing"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--kernels")
    ap.add_argument("--reduce", nargs=2)
    ap.add_argument("--slow-ms", type=float, default=12.0)
    ap.add_argument("--top", type=int, default=8)
    a = ap.parse_args()
    if a.reduce:
        reduce_csvs(*a.reduce)
    if a.trace:
        trace_report(a.trace, a.slow_ms)
    if a.kernels:
        kernels_report(a.kernels, a.top)


if __name__ == "__main__":
    sys.exit(main())
