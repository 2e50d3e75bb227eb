#!/bin/bash
# kernel timeline of the last dense factorization of tools/dense_lab.py (rocprofv3 --kernel-trace), summarized
N=${1:-10000}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dl && rocprofv3 --kernel-trace --output-format csv -d /tmp/dl -o d -- python3 $GRAFT_REPO_ROOT/tools/dense_lab.py $N 3 > /dev/null 2>&1
F=$(find /tmp/dl -name "*kernel_trace.csv" | head -1)
cp "$F" $GRAFT_REPO_ROOT/gpurun_out/last_dense_trace.csv 2>/dev/null; python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$F")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ld=[r for r in rows if "k_ldl" in r["Kernel_Name"] or "assemble" in r["Kernel_Name"]]
idx=max(i for i,r in enumerate(ld) if "assemble" in r["Kernel_Name"])
f=[r for r in ld[idx:] if "chain" not in r["Kernel_Name"]]
t0=int(f[0]["Start_Timestamp"])
print("kernels in last factor", len(f), "span ms %.3f" % ((max(int(r["End_Timestamp"]) for r in f)-t0)/1e6))
def show(rs):
    for r in rs:
        print("%-14s q%s start %8.1f dur %7.1f grid %sx%s" % (r["Kernel_Name"].split("(")[0].replace("void ","")[:14], r["Queue_Id"], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"], r["Grid_Size_Y"]))
show(f[:30]); print("..."); show(f[len(f)//2:len(f)//2+14]); print("..."); show(f[-16:])
agg=collections.defaultdict(lambda:[0,0])
for r in f:
    k=r["Kernel_Name"].split("(")[0]; agg[k][0]+=1; agg[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in agg.items(): print(k, v[0], "%.2f ms"%(v[1]/1e6))
PY
