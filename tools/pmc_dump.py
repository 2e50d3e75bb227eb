"""Print the counters of the largest dispatch (by grid) of each kernel whose name contains <substr>, from rocprofv3 --pmc output.
usage: pmc_dump.py <dir with *counter_collection.csv> <substr>"""
import csv, sys, glob, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
meta = {}
for r in rows:
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Dispatch_Id"])
    by[key][r["Counter_Name"]] = by[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[key] = int(r["Grid_Size"])
best = {}
for key, g in meta.items():
    if key[0] not in best or g > meta[best[key[0]]]:
        best[key[0]] = key
for name, key in best.items():
    print(name, "dispatch", key[1], "grid", meta[key])
    for c, v in sorted(by[key].items()):
        print("   %-36s %.4g" % (c, v))
