"""Summarise a rocprofv3 --pmc FETCH_SIZE --kernel-trace CSV directory of a bench.py run: mean counter value
per dispatch of the decode attention kernel.  Usage: python tools/pmc_in_situ_summary.py <dir> <counter>"""
import csv
import glob
import sys
from collections import defaultdict

root, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "")
            key = ("decode_mfma" if "decode_mfma_kernel" in name else "decode_stage2" if "decode_stage2" in name
                   else "extend_attn" if "extend_attn_kernel" in name else None)
            if key is None:
                continue
            a = acc[(f.split("/")[-2] if "/" in f else f, key)]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for (proc, key), (n, tot) in sorted(acc.items()):
    print(f"{proc} {key}: dispatches {n}, mean {counter} {tot / n:.1f}, total {tot:.0f}")
