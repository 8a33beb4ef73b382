#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline block. Run on the GPU box:
#   gpurun -- 'bash profiles/collect.sh c2 r01'
# Writes raw CSVs under gpurun_out/prof_<tag>_<workload>/ ; profiles/summarize.py turns them
# into the small files committed under profiles/.
# PMC passes are separate runs with --pmc only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; never combine --pmc with trace domains other than kernel dispatch).
set -u
WL=${1:-c2}; TAG=${2:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
  python "$ROOT/bench.py" --workload "$WL" --no-cpu-baseline > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- \
  python "$ROOT/bench.py" --workload "$WL" --steps 300 --warmup 20 --no-cpu-baseline --no-verify > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- \
  python "$ROOT/bench.py" --workload "$WL" --steps 300 --warmup 20 --no-cpu-baseline --no-verify > "$OUT/bench_write.log" 2>&1
cd "$ROOT" && python profiles/summarize.py "$WL" "$TAG"
