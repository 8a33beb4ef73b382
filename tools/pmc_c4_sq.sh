#!/bin/bash
# SQ / LDS counters of the full MFMA pass on config 4's shape, for compile-time variants:
#   tools/pmc_c4_sq.sh "" "-DLS_GEMM_ABL_NODMA"
R=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  (cd $R/lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $v" >/dev/null 2>&1)
  echo "== [$v]"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL"; do
    rm -rf /tmp/pq; (cd /tmp && TMPDIR=/tmp LS_GEMM_ABL_NOPASS=1 rocprofv3 --pmc $set --output-format csv -d /tmp/pq -o p -- python $R/bench.py --workload c4 --c4-rows 2000000 --steps 6 --warmup 2 --no-cpu-baseline --no-verify >/dev/null 2>&1)
    python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for fn in glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(fn)):
        kn=row['Kernel_Name']
        if 'ls_gemm_filter_kernel' in kn and 'false' in kn:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items(): print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  done
done
(cd $R/lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s >/dev/null 2>&1)
