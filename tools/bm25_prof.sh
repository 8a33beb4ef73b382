#!/bin/bash
# Per-kernel GPU time of one BM25 name query (tools/bm25_time.py: 3 tokens over 200 k names):
#   gpurun -- 'bash tools/bm25_prof.sh r03'
TAG=${1:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/bm25_$TAG; mkdir -p $OUT
python $R/tools/bm25_time.py > $OUT/time.txt 2>&1; cat $OUT/time.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o t -- python $R/tools/bm25_time.py > $OUT/under_rocprof.txt 2>&1
f=$(find /tmp/bp -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv | cut -c1-150
