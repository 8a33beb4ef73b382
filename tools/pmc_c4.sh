#!/bin/bash
# HBM fetch traffic of the batched MFMA pass at config 4's per-GPU shape (reduced row count):
# is the corpus slice shared by its two query tiles through the XCD's L2, or fetched twice?
ROOT=$(cd "$(dirname "$0")/.." && pwd); ROWS=${1:-4000000}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pc4 -o p -- python $ROOT/bench.py --workload c4 --c4-rows $ROWS --steps 10 --warmup 2 --no-cpu-baseline --no-verify > /tmp/pc4.log 2>&1
python - <<PY
import csv,glob
v=[]
for f in glob.glob("/tmp/pc4/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and "ls_gemm_filter_kernel" in r["Kernel_Name"] and "Lb0" in r["Kernel_Name"]+"Lb0": v.append((r["Kernel_Name"][:60], float(r["Counter_Value"])))
main=[x for n,x in v if "false" in n]
alg=$ROWS*768*2
if main:
    m=sum(main)/len(main)*1024*2
    print(f"main pass: FETCH_SIZE x2 = {m/1e9:.2f} GB per launch; corpus = {alg/1e9:.2f} GB; ratio {m/alg:.2f}  ({len(main)} launches)")
PY
