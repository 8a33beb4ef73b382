#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline blocks. Run on the GPU box:
#   gpurun -- 'bash profiles/collect.sh c2 r02'        (workloads: c2 c2m c2p c2x8 c2px8 c2x32 c3 c4 bm25)
# Raw CSVs go under /tmp/prof_<tag>_<workload>/ (tens of MB: they stay on the box);
# profiles/summarize.py condenses them into the small files that are committed under profiles/
# (written to gpurun_out/profiles/, which gpurun copies back).
# Counter passes are separate runs with --pmc only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; --pmc is never combined with trace domains other than kernel dispatch).
set -u
WL=${1:-c2}; TAG=${2:-r02}
ROOT=$(pwd)
OUT=/tmp/prof_${TAG}_${WL}
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$WL" = bm25 ]; then
  CMD="python $ROOT/tools/bm25_time.py"
  STEPS=""
else
  CMD="python $ROOT/bench.py --workload $WL --secondary none --no-host-api --no-cpu-baseline"
  case $WL in c4) STEPS="--steps 20 --warmup 3";; c3) STEPS="--steps 300 --warmup 20";; *) STEPS="--steps 2000 --warmup 100";; esac
fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD $STEPS > "$OUT/bench_trace.log" 2>&1
[ "$WL" = bm25 ] || STEPS="$STEPS --no-verify"
case $WL in c4) PSTEPS="--steps 6 --warmup 2 --no-verify";; bm25) PSTEPS="";; *) PSTEPS="--steps 200 --warmup 20 --no-verify";; esac
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $CMD $PSTEPS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- $CMD $PSTEPS > "$OUT/bench_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU \
  --output-format csv -d "$OUT/pmc_sq" -o p -- $CMD $PSTEPS > "$OUT/bench_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU \
  --output-format csv -d "$OUT/pmc_sq2" -o p -- $CMD $PSTEPS > "$OUT/bench_sq2.log" 2>&1
cd "$ROOT" && python profiles/summarize.py "$WL" "$TAG"
