#!/bin/bash
# per-kernel rocprofv3 durations of the config-3 bench for ls_gemm.hip compile-time variants
cd "$(dirname "$0")/.."
ROOT=$(pwd)
for v in "$@"; do
  (cd lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $v" >/dev/null 2>&1)
  rm -rf /tmp/prof_ab
  (cd /tmp && TMPDIR=/tmp env ${NOPASS:+LS_GEMM_ABL_NOPASS=1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o t -- python $ROOT/bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-baseline --no-verify >/dev/null 2>&1)
  echo "[$v]"
  python - <<PY
import csv,glob
f=glob.glob("/tmp/prof_ab/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>50: print("   ", r["Name"][:48], r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
done
(cd lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s >/dev/null 2>&1)
