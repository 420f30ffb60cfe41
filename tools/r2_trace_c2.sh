#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python3 $R/bench.py --workload c2 --no-extras --repeats 2 > $R/gpurun_out/prof_c2.log 2>&1
TRACE=$(find $R/gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv,re,collections
rows=[]
for r in csv.DictReader(open("$TRACE")):
    n=re.sub(r"^void\s+","",r["Kernel_Name"]); n=re.sub(r"<.*","",n).replace("lcx::","")
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n))
rows.sort()
pairs=collections.defaultdict(list)
for (s0,e0,n0),(s1,e1,n1) in zip(rows,rows[1:]):
    g=(s1-e0)/1e3
    if g<200: pairs[(n0,n1)].append(g)
for k,v in sorted(pairs.items(), key=lambda kv:-sum(kv[1])):
    if len(v)>=20 and sum(v)/len(v) > 0.3: print("%-26s -> %-26s n=%5d gap avg %6.2f  med %6.2f"%(k[0],k[1],len(v),sum(v)/len(v),sorted(v)[len(v)//2]))
PY
