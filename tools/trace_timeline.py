"""Scratch: summarise one dense factorization from a rocprofv3 kernel trace CSV (start/end per kernel, per queue)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the k-th assemble kernel
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_dense_assemble")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
i0 = idx[which]; i1 = idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
seg = rows[i0:i1]
last_syrk = max(int(r["End_Timestamp"]) for r in seg if "syrk" in r["Kernel_Name"])
print("factor span us", (last_syrk - t0) / 1e3)
busy = collections.defaultdict(float)
for r in seg:
    if int(r["Start_Timestamp"]) <= last_syrk:
        busy[(r["Kernel_Name"].split("(")[0][:24], r["Queue_Id"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]): print(k, "%.1f us" % v)
n = 0
for r in seg:
    if int(r["Start_Timestamp"]) > last_syrk: break
    name = r["Kernel_Name"].split("(")[0][:20]
    if n < int(sys.argv[3]) if len(sys.argv) > 3 else 60:
        print("%9.1f %9.1f q%s %s grid=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], name, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]))
    n += 1
