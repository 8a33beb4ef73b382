#!/bin/bash
# Per-kernel durations of one bench.py workload under rocprofv3 (kernel trace + stats only).
#   gpurun -- 'bash tools/prof_kernels.sh c3 r02 [extra bench args]'
# Output: gpurun_out/prof_<tag>_<wl>/kernel_stats.csv (+ the bench line that ran under the profiler)
set -u
WL=${1:-c3}; TAG=${2:-r02}; shift 2 || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
  python "$ROOT/bench.py" --workload "$WL" --secondary none --no-host-api --no-cpu-baseline "$@" \
  > "$OUT/bench_trace.log" 2>&1
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -12 "$OUT/kernel_stats.csv" | cut -c1-160
grep '^{' "$OUT/bench_trace.log" | tail -1 | cut -c1-400
