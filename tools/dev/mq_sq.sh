#!/bin/bash
# SQ counters + kernel time of ls_mq launches of one shape:  bash tools/dev/mq_sq.sh "384 32" "1024 32" ...
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/mq_one.py <<PY
import sys; sys.path.insert(0, '$R')
import torch
from lean_explore_amd.index import FlatIPIndex
from tests import helpers as H
d, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 50
ix = FlatIPIndex.from_array(H.gauss(1234, 200_000, d))
q = torch.from_numpy(H.gauss(5678, nq, d)).cuda()
for _ in range(60): ix.search_device(q, k, pipeline=True)
ix.check(); ix.close()
PY
for cfg in "$@"; do
  set -- $cfg
  O=/tmp/mqsq_$1_$2; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python /tmp/mq_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/a -o p -- python /tmp/mq_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -o p -- python /tmp/mq_one.py $1 $2 $3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/c -o p -- python /tmp/mq_one.py $1 $2 $3 > /dev/null 2>&1
  python - "$O" "$1" "$2" <<'PY'
import sys, glob, csv, collections
O=sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(O + '/[abc]/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ls_mq_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v[5:]) / max(1, len(v[5:])) for k, v in acc.items()}
dur = None
for f in glob.glob(O + '/t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ls_mq_kernel' in r['Name']: dur = float(r['AverageNs']) / 1e3
print(f"d={sys.argv[2]} nq={sys.argv[3]}: kernel {dur} us; " + ", ".join(f"{k} {v:,.0f}" for k, v in sorted(m.items())), flush=True)
if dur and 'GRBM_GUI_ACTIVE' in m:
    clk = m['GRBM_GUI_ACTIVE'] / 8 / dur / 1e3
    print(f"   shader clock ~ {clk:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time); MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0) / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} of all SIMD cycles; "
          f"LDS conflicts / LDS instructions {m.get('SQ_LDS_BANK_CONFLICT',0) / max(1, m.get('SQ_INSTS_LDS',1)):.4f}; conflicts / LDS active cycles {m.get('SQ_LDS_BANK_CONFLICT',0) / max(1, m.get('SQ_LDS_IDX_ACTIVE',1)):.4f}")
PY
done
