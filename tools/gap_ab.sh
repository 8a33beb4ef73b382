#!/bin/bash
# GPU-side gaps between the consecutive kernels of a batched search, per library variant:
#   gpurun -- 'bash tools/gap_ab.sh c3 "default r1"'
WL=$1; VARS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
for v in $VARS; do
  unset LEANSEARCH_LIB
  [ "$v" != default ] && export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$v.so
  rm -rf /tmp/pk; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o t -- python $R/bench.py --workload $WL --steps 200 --warmup 20 --secondary none --no-host-api --no-cpu-baseline --no-verify > /tmp/pk.log 2>&1)
  echo "== $WL [$v]"
  python - <<PY
import csv,glob,collections
rows=[]
for fn in glob.glob("/tmp/pk/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']))
rows.sort()
def short(n):
    for k in ('prep_f16','tau','select','convert'):
        if k in n: return k
    if 'gemm_filter' in n: return 'sample' if 'true>' in n else 'main'
    return n[:20]
gaps=collections.defaultdict(list); durs=collections.defaultdict(list)
for (s0,e0,n0),(s1,e1,n1) in zip(rows[:-1],rows[1:]):
    a,b=short(n0),short(n1)
    if a in ('prep_f16','sample','tau','main','select') and b in ('prep_f16','sample','tau','main','select'):
        gaps[a+'->'+b].append(s1-e0)
    durs[a].append(e0-s0)
tot=0
for k in ('prep_f16->sample','sample->tau','tau->main','main->select','select->prep_f16'):
    v=sorted(gaps.get(k,[0])); m=v[len(v)//2]; tot+=m
    print(f'   gap {k:20s} median {m/1e3:7.2f} us  (n={len(v)})')
dt=0
for k in ('prep_f16','sample','tau','main','select'):
    v=sorted(durs.get(k,[0])); m=v[len(v)//2]; dt+=m
    print(f'   dur {k:20s} median {m/1e3:7.2f} us')
print(f'   sum of durations {dt/1e3:.1f} us + gaps {tot/1e3:.1f} us = {(dt+tot)/1e3:.1f} us')
PY
done
