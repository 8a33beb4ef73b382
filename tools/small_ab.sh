#!/bin/bash
# same-box A/B of library variants on the small-shard step time: tools/small_ab.sh default nosmall ...
for r in 1 2; do for lib in "$@"; do
  if [ $lib = default ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$PWD/lean-explore_amd/variants/libleansearch_$lib.so; fi
  echo "== $lib (round $r): $(python tools/small_n_sweep.py 2>&1 | grep N= | tr '\n' ' ')"
done; done
