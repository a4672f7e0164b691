"""Mean counter value per dispatch, per kernel, of a rocprofv3 --pmc run: python tools/pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import sys
from collections import defaultdict

root, want = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if want and not any(w in name for w in want):
                continue
            short = name.replace("void ", "").replace("semipd::", "").split("(")[0][:60]
            a = acc[(short, row.get("Counter_Name"))]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for (name, counter), (n, tot) in sorted(acc.items()):
    print(f"{name:60s} {counter:12s} dispatches {n:4d} mean {tot / n:14.1f}")
