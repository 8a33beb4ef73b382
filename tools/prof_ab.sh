#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) for library variants, same box:
#   gpurun -- 'bash tools/prof_ab.sh c3 "default r1" "--steps 300 --warmup 20"'
WL=$1; VARS=$2; EXTRA=${3:-}
R=$(cd "$(dirname "$0")/.." && pwd)
for v in $VARS; do
  unset LEANSEARCH_LIB
  [ "$v" != default ] && export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$v.so
  rm -rf /tmp/pk; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o t -- python $R/bench.py --workload $WL $EXTRA --secondary none --no-host-api --no-cpu-baseline --no-verify > /tmp/pk.log 2>&1)
  echo "== $WL [$v]  $(grep -o '"ms_per_step": [0-9.]*' /tmp/pk.log | tail -1)"
  python - <<PY
import csv,glob
for fn in glob.glob("/tmp/pk/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        n=r["Name"]
        if "ls_" in n and "convert" not in n:
            print(f'   {n[:58]:58s} calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
done
