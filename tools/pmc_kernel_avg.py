"""Average of one PMC counter over the real (non-latched) launches of the kernels whose name contains a pattern.
usage: pmc_kernel_avg.py <counter_collection.csv> <counter> <pattern> [min_value]"""
import csv, sys
path, counter, pat = sys.argv[1], sys.argv[2], sys.argv[3]
mn = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and pat in r["Kernel_Name"]]
real = [v for v in vals if v > mn]
print(counter, pat, "launches", len(vals), "real", len(real), "avg_real", sum(real) / max(1, len(real)), "sum", sum(vals))
