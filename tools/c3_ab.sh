#!/bin/bash
# Config 3, this tree against the round-3 tree (scratch/r03: `git worktree add scratch/r03 7b45926` + make), interleaved
# on ONE box:   gpurun -- 'bash tools/c3_ab.sh [extra bench args for this tree]'
R=$(cd "$(dirname "$0")/.." && pwd)
A="--workload c3 --steps 300 --warmup 20 --secondary none --no-host-api --no-cpu-baseline"
for rep in 1 2 3; do
  for t in r03 r04; do
    if [ $t = r03 ]; then D=$R/scratch/r03; X=""; else D=$R; X="$@"; fi
    (cd $D && timeout 300 python bench.py $A $X 2>/dev/null | tail -1) | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$t rep $rep: %.1f us/batch  %.0f q/s  pass kernel %.1f us (frac %.3f)  whole-batch frac %.3f  recall %s repaired %s' % (j['ms_per_step']*1e3, j['value'], r['kernel_ms']*1e3, r['frac'], r['frac_whole_batch'], j['recall_at_k'], j['repaired_queries']))"
  done
done
