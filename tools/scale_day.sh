#!/bin/bash
# tools/scale_day.sh [max GPUs] - everything to run the first time an N-GPU MI355X node is available (verdict r5
# item 7; no exchange has run between two physical GPUs yet). For G = 1, 2, 4, 8 (up to the GPUs visible) and both
# process models - ONE process with the library's sharded handle (ls_create_sharded: RCCL inside the library) and
# one process per GPU (torch.distributed over RCCL, what the driver's `bench.py --gpus N` runs) - it
#   1. asserts that sharded results are bit-identical to one index over all rows (tools/scale_check.py),
#   2. runs bench.py for c2 (N = 200 k, d = 384 fp32, batch 1: strong scaling) and c4 (12.5 M rows x 768 fp16 PER
#      GPU, batch 256: weak scaling) and keeps every output under gpurun_out/scale_day/,
#   3. prints the handle's exchange record (ls_shard_exchange_info) and the table north_star asks for
#      (queries/s at 1 / 2 / 4 / 8 GPUs, roofline fraction, speed-up over G = 1) next to the PREDICTION of
#      ~1.0 / 1.6 / 2.8 / 4.5 x, c4 (weak): ~2.0 / 3.9 / 7.5 x - so the first real run
#      tests a prediction.
# Rehearsal on one GPU (tests/test_bench_gpu.py):
#   SCALE_SHARE=1 SCALE_C4_ROWS=100000 SCALE_STEPS_C2=48 SCALE_STEPS_C4=3 SCALE_CHECK_ROWS=20000 SCALE_CHECK_C2_ROWS=40000 \
#       bash tools/scale_day.sh 2
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
NVIS=$(python -c "import torch; print(torch.cuda.device_count())")
MAXG=${1:-$NVIS}
OUT=$R/gpurun_out/scale_day
mkdir -p "$OUT"
C2S=${SCALE_STEPS_C2:-5000}; C4S=${SCALE_STEPS_C4:-30}; C4R=${SCALE_C4_ROWS:-12500000}
CHK=${SCALE_CHECK_ROWS:-1000000}; CHK2=${SCALE_CHECK_C2_ROWS:-200000}
SHARE=""
if [ -n "${SCALE_SHARE:-}" ]; then export LS_BENCH_SHARE_GPU=1; SHARE="--share"; fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--secondary none --no-host-api --no-cpu-baseline"
FAIL=0
GS=""
for G in 1 2 4 8; do
  if [ "$G" -le "$MAXG" ] && { [ "$G" -le "$NVIS" ] || [ -n "$SHARE" ]; }; then GS="$GS $G"; fi
done
echo "scale_day: GPUs visible $NVIS, running G =$GS ${SHARE:+(rehearsal: shards share GPUs)}"
# 1. bit-equality with one index, both process models
python tools/scale_check.py --gpus "$(echo $GS | tr ' ' ',')" --c2-rows "$CHK2" --c4-rows "$CHK" $SHARE 2>&1 | grep -E "^\[inlib|scale_check" | tee "$OUT/check_inlib.txt"
grep -q "scale_check inlib: OK" "$OUT/check_inlib.txt" || FAIL=1
for G in $GS; do
  [ "$G" = 1 ] && continue
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$G" --master-addr 127.0.0.1 --master-port $((29700 + G)) \
      tools/scale_check.py --dist --c2-rows "$CHK2" --c4-rows "$CHK" $SHARE 2>&1 | grep -E "^\[dist|scale_check" | tee "$OUT/check_dist_g$G.txt"
  grep -q "scale_check dist: OK" "$OUT/check_dist_g$G.txt" || FAIL=1
done
# 2. bench.py, both models, c2 and c4
for G in $GS; do
  python bench.py --gpus "$G" --workload c2 --steps "$C2S" --warmup 100 $COMMON > "$OUT/inlib_c2_g$G.txt" 2> "$OUT/inlib_c2_g$G.err" || FAIL=1
  python bench.py --gpus "$G" --workload c4 --steps "$C4S" --warmup 3 --c4-rows "$C4R" $COMMON > "$OUT/inlib_c4_g$G.txt" 2> "$OUT/inlib_c4_g$G.err" || FAIL=1
  if [ "$G" = 1 ]; then
    cp "$OUT/inlib_c2_g1.txt" "$OUT/dist_c2_g1.txt"
    cp "$OUT/inlib_c4_g1.txt" "$OUT/dist_c4_g1.txt"
  else
    for WL in c2 c4; do
      S=$C2S; [ $WL = c4 ] && S=$C4S
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$G" --master-addr 127.0.0.1 --master-port $((29800 + G)) \
          bench.py --gpus "$G" --workload $WL --steps "$S" --warmup 3 --c4-rows "$C4R" $COMMON > "$OUT/dist_${WL}_g$G.txt" 2> "$OUT/dist_${WL}_g$G.err" || FAIL=1
    done
  fi
done
# 3. the table
python - "$OUT" $GS <<'PY'
import json, sys
out, gs = sys.argv[1], [int(g) for g in sys.argv[2:]]
pred = {"c2": {1: 1.0, 2: 1.6, 4: 2.8, 8: 4.5}, "c4": {1: 1.0, 2: 2.0, 4: 3.9, 8: 7.5}}
print("\n| model | workload | GPUs | queries/s | x G=1 | predicted x | roofline frac | exchange | recall |")
print("|---|---|---|---|---|---|---|---|---|")
for model in ("inlib", "dist"):
    for wl in ("c2", "c4"):
        base = None
        for g in gs:
            try:
                text = open(f"{out}/{model}_{wl}_g{g}.txt").read().splitlines()
                o = json.loads([l for l in text if l.startswith("{")][-1])
                det = [l for l in text if l.startswith("[bench details] ")]
                full = json.loads(det[-1][len("[bench details] "):]) if det else o
            except Exception as e:
                print(f"| {model} | {wl} | {g} | (no line: {e!r}) | | | | | |")
                continue
            base = base or o["value"]
            exch = full.get("exchange", full.get("config", {}).get("exchange", "-"))
            print(f"| {model} | {wl} | {g} | {o['value']:,.0f} | {o['value'] / base:.2f} | {pred[wl][g]:.1f} | "
                  f"{o['roofline']['frac']:.3f} | {exch} | {o.get('recall_at_k')} |")
PY
grep -h "exchange:" "$OUT/check_inlib.txt" | head -4
if [ $FAIL = 0 ]; then echo "scale_day: OK"; else echo "scale_day: FAILED (see $OUT)"; fi
exit $FAIL
