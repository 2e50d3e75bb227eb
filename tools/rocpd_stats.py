"""Kernel statistics (the columns of rocprofv3 --stats) from a rocprofv3 rocpd database: calls, total, average, share, min, max.
usage: rocpd_stats.py results.db [out.csv]"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name, name)))
tot = sum(r[2] for r in rows)
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")] + [(r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]) for r in rows]
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
