"""Per outer panel of the last factorization in a rocprofv3 kernel trace: start of the wide trailing update (stream 2), its
duration, the period to the next one and the time the serial chain of the next panel took on the main stream.
usage: dense_trace_periods.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ld = [r for r in rows if "k_ldl" in r["Kernel_Name"] or "assemble" in r["Kernel_Name"]]
idx = max(i for i, r in enumerate(ld) if "assemble" in r["Kernel_Name"])
f = [r for r in ld[idx:] if "chain" not in r["Kernel_Name"]]
t0 = int(f[0]["Start_Timestamp"])
qs = sorted(set(r["Queue_Id"] for r in f))
main_q = f[0]["Queue_Id"]
wide = [r for r in f if r["Queue_Id"] != main_q]
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
print("p  b_start  b_dur  period  chain_busy(main stream kernels inside the period)")
for i, r in enumerate(wide):
    nxt = S(wide[i + 1]) if i + 1 < len(wide) else E(f[-1])
    inside = [k for k in f if k["Queue_Id"] == main_q and S(k) >= S(r) and S(k) < nxt]
    busy = sum(E(k) - S(k) for k in inside)
    print("%2d %8.1f %6.1f %7.1f %7.1f (%d kernels)" % (i, S(r), E(r) - S(r), nxt - S(r), busy, len(inside)))
print("total span %.1f us" % (E(f[-1])))
