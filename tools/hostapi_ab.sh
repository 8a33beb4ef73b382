#!/bin/bash
# same-box A/B of library variants on the synchronous host API latency: tools/hostapi_ab.sh default fin1024 ...
for r in 1 2 3; do for lib in "$@"; do
  if [ $lib = default ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$PWD/lean-explore_amd/variants/libleansearch_$lib.so; fi
  echo "== $lib (round $r): $(python tools/hostapi_time.py 2>&1 | grep 'nq=1 k=50' | sed 's/.*k=50: //')"
done; done
