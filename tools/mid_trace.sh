#!/bin/bash
# kernel timeline of ONE mid-size qpdo_solve on the generic path (default: C1): every launch of the last solve with duration and gap
W=${1:-C1}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sl && rocprofv3 --kernel-trace --output-format csv -d /tmp/sl -o s -- python3 $GRAFT_REPO_ROOT/tools/mid_trace.py $W > /tmp/sl.out 2>&1
grep -v "^$" /tmp/sl.out | tail -3
F=$(find /tmp/sl -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=sorted(csv.DictReader(open("$F")), key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].split("(")[0].replace("void ","") for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith("k_resid_m")]
per=len(idx)//3
start=idx[-per]
sub=list(zip(rows,names))[start:]
agg=collections.defaultdict(lambda:[0,0]); prev=None; gaps=0
for r,nm in sub:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"]); agg[nm][0]+=1; agg[nm][1]+=en-st
    if prev: gaps+=max(0,st-prev)
    prev=en
span=int(sub[-1][0]["End_Timestamp"])-int(sub[0][0]["Start_Timestamp"])
print("last solve: %d launches, %d passes, span %.2f ms, busy %.2f ms, gaps %.2f ms" % (len(sub), per, span/1e6, sum(v[1] for v in agg.values())/1e6, gaps/1e6))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:30]: print("%-44s %5d  total %8.1f us  avg %6.2f" % (k[:44], v[0], v[1]/1e3, v[1]/1e3/v[0]))
print("--- launches of passes 10..12")
i10=idx[-per+10]-start; i13=idx[-per+13]-start
prev=None
for r,nm in sub[i10:i13]:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print("%-40s dur %6.1f gap %6.1f  grid %s" % (nm[:40], (en-st)/1e3, (st-prev)/1e3 if prev else 0.0, r["Grid_Size_X"])); prev=en
PY
