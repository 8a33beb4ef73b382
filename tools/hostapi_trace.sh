#!/bin/bash
# GPU-side timeline of the synchronous host API (config 2): scan, gap, finalize, turnaround
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ht
cat > /tmp/ht.py <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
import os
D=int(os.environ.get("HT_D","384")); K=int(os.environ.get("HT_K","50"))
c=H.gauss(1234,200000,D); q=H.gauss(5678,1,D)
ix=FlatIPIndex.from_array(c)
for _ in range(400): ix.search(q,K)
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/ht -o t -- python /tmp/ht.py >/dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("/tmp/ht/**/*kernel_trace.csv",recursive=True)[0]
rows=[(r["Kernel_Name"][:20],int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:r[1])
rows=[r for r in rows if "ls_scan" in r[0] or "ls_finalize" in r[0]][-600:]
import statistics as st
scan=[];gap=[];fin=[];turn=[]
for i in range(len(rows)-2):
    a,b,c=rows[i],rows[i+1],rows[i+2]
    if "scan" in a[0] and "finalize" in b[0] and "scan" in c[0]:
        scan.append(a[2]-a[1]); gap.append(b[1]-a[2]); fin.append(b[2]-b[1]); turn.append(c[1]-b[2])
m=lambda x: round(st.median(x)/1e3,2)
print("scan",m(scan),"gap",m(gap),"finalize",m(fin),"turnaround(end fin -> next scan start)",m(turn),"period",round((m(scan)+m(gap)+m(fin)+m(turn)),2))
PY
