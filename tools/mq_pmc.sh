#!/bin/bash
# PMC passes over ls_mq launches of one shape (developer tool):  gpurun -- 'bash tools/mq_pmc.sh'
# prints, per (d, nq), the per-launch mean of a few TLB / L2-write counters of ls_mq_kernel.
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/mq_one.py <<PY
import sys; sys.path.insert(0, '$R')
import torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
d, nq = int(sys.argv[1]), int(sys.argv[2])
ix = FlatIPIndex.from_array(H.gauss(1234, 200_000, d))
q = torch.from_numpy(H.gauss(5678, nq, d)).cuda()
for _ in range(40): ix.search_device(q, 50, pipeline=True)
ix.check(); ix.close()
PY
for cfg in "1024 2" "1024 16" "384 16" "1024 1"; do
  set -- $cfg
  i=0
  for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
             "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
    i=$((i+1)); O=/tmp/mqpmc_$1_$2_$i; rm -rf $O
    rocprofv3 --pmc $pmc --output-format csv -d $O -o p -- python /tmp/mq_one.py $1 $2 > /dev/null 2>&1
    python - "$O" "$1" "$2" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ls_mq_kernel' in r['Kernel_Name'] or 'ls_scan_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(f"d={sys.argv[2]} nq={sys.argv[3]}: " + ", ".join(f"{k} {sum(v[5:]) / max(1, len(v[5:])):,.0f}" for k, v in sorted(acc.items())), flush=True)
PY
  done
done
