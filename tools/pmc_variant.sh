#!/bin/bash
# SQ / LDS / clock counters of the full MFMA pass for one library variant (see tools/ab_bench.py):
#   gpurun -- 'bash tools/pmc_variant.sh c4 default "--c4-rows 4000000 --steps 6 --warmup 2"'
# One rocprofv3 --pmc pass per counter set (kernel dispatch only, never combined with traces).
WL=$1; VAR=$2; EXTRA=${3:-}
R=$(cd "$(dirname "$0")/.." && pwd)
[ "$VAR" != default ] && export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$VAR.so
echo "== $WL [$VAR]"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL"; do
  rm -rf /tmp/pq; (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/bench.py --workload $WL $EXTRA --secondary none --no-host-api --no-cpu-baseline --no-verify >/dev/null 2>&1)
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for fn in glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(fn)):
        kn=row['Kernel_Name']
        if 'ls_gemm_filter_kernel' in kn and 'false' in kn:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items(): print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
dur=[]
for fn in glob.glob("/tmp/pq/**/*kernel_trace.csv",recursive=True):
    for row in csv.DictReader(open(fn)):
        if 'ls_gemm_filter_kernel' in row['Kernel_Name'] and 'false' in row['Kernel_Name']:
            dur.append(float(row['End_Timestamp'])-float(row['Start_Timestamp']))
if dur: print(f"   kernel_ns_avg                {sum(dur)/len(dur):16.0f}  (n={len(dur)})")
PY
done
