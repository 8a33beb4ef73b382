#!/bin/bash
# A/B of scan-kernel variants on ONE box via bench.py (c2 and c2p, kernel time by the timed region's events)
R=$(cd "$(dirname "$0")/../.." && pwd)
for r in 1 2; do
  for l in $1; do
    if [ "$l" = default ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$l.so; fi
    for wl in c2 c2p c2m; do
      python $R/bench.py --workload $wl --steps 3000 --warmup 200 --secondary none --no-host-api --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python -c "
import sys, json; o = json.loads(sys.stdin.read()); print('== $l $wl round $r: kernel %.2f us, step %.2f us, frac %.4f' % (o['roofline']['kernel_ms'] * 1e3, o['ms_per_step'] * 1e3, o['roofline']['frac']))"
    done
  done
done
