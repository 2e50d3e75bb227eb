"""Average duration of a kernel from a rocprofv3 kernel-trace CSV, with and without the latched no-op launches (< 5 us)."""
import csv, sys
pat = sys.argv[2]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if pat in r["Kernel_Name"]]
real = [x for x in d if x > 5000]
print(pat, "launches", len(d), "avg all %.1f us" % (sum(d) / len(d) / 1e3), "| launches > 5 us", len(real), "avg %.1f us" % (sum(real) / len(real) / 1e3))
