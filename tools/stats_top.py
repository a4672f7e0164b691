"""Print the top rows of a rocprofv3 *_kernel_stats.csv with shortened kernel names."""
import csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== {path}: {len(rows)} kernels, {tot/1e6:.1f} ms total")
    for r in rows[:int(18)]:
        n = r["Name"]
        n = n.replace("void semipd::", "").split("(")[0]
        if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
            import re
            m = re.search(r"MT(\d+x\d+x\d+)", n)
            n = "hipBLASLt " + (m.group(1) if m else "") + (" SK" if "_SK" in n else "")
        print(f"  {n[:60]:60s} calls={int(r['Calls']):7d} avg={float(r['AverageNs'])/1e3:9.1f}us total={float(r['TotalDurationNs'])/1e6:9.1f}ms {float(r['Percentage']):5.1f}%")
