#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_c3_$tag -o p -- python $R/tools/c3time.py > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_c3_$tag/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for row in csv.DictReader(open(fn)):
        if 'gemm_filter_kernel<48, false>' in row['Kernel_Name'] or 'ILi48ELb0' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
