#!/bin/bash
# kernel timeline of the generic qpdo_solve path on small problems (tools/small_latency.py): launches per pass, busy and idle time
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sl && rocprofv3 --kernel-trace --output-format csv -d /tmp/sl -o s -- python3 $GRAFT_REPO_ROOT/tools/small_latency.py > /tmp/sl.out 2>&1
cat /tmp/sl.out | grep -v "^$" | tail -5
F=$(find /tmp/sl -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=sorted(csv.DictReader(open("$F")), key=lambda r:int(r["Start_Timestamp"]))
# last solve of the C3 instance: find the last 'k_resid_m' block before KAT... simply take the window of the 3rd-last..: use kernel names
names=[r["Kernel_Name"].split("(")[0].replace("void ","") for r in rows]
# split into solves at k_begin/warm start markers is fragile: print aggregate over the whole run instead
agg=collections.defaultdict(lambda:[0,0])
for r,nm in zip(rows,names): agg[nm][0]+=1; agg[nm][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
tot=sum(v[1] for v in agg.values()); span=int(rows[-1]["End_Timestamp"])-int(rows[0]["Start_Timestamp"])
print("kernels %d busy %.2f ms span %.2f ms" % (len(rows), tot/1e6, span/1e6))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]: print("%-40s %6d %8.1f us avg %6.2f" % (k[:40], v[0], v[1]/1e3, v[1]/1e3/v[0]))
# a window of 60 consecutive kernels in the middle, with gaps
mid=len(rows)//2
prev=None
for r,nm in list(zip(rows,names))[mid:mid+70]:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print("%-34s dur %6.1f gap %6.1f" % (nm[:34], (en-st)/1e3, (st-prev)/1e3 if prev else 0.0)); prev=en
PY
