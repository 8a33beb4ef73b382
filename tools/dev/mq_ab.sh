#!/bin/bash
# A/B of ls_mq variants on ONE box: tools/dev/mq_ab.sh "lib1 lib2 ..." rounds   (lib = default | variant name)
R=$(cd "$(dirname "$0")/../.." && pwd)
for r in $(seq 1 ${2:-2}); do
  for l in $1; do
    if [ "$l" = default ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$l.so; fi
    NQS=${MQ_NQS_ALL:-1,8,16}; [ "$l" != r05 ] && NQS=${MQ_NQS_NEW:-1,8,16,24,32}
    echo "== $l round $r"; MQ_SHAPES=${MQ_SHAPES:-0,1} MQ_NQS=$NQS python $R/tools/multiq_time.py 2>&1 | grep "^N="
  done
done
