#!/bin/bash
# Everything profiles/make_tables.py needs for one round, on ONE gpurun box (~30 GPU-minutes):
#   gpurun --timeout 3000 -- 'bash profiles/run_all.sh r06'
# then:  cp gpurun_out/profiles/* profiles/ && python profiles/make_tables.py r06
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/profiles; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default_driver_style.json 2> $O/${TAG}_bench_default.err
for wl in c1 c2 c2m c2p c2x8 c2px8 c2x32 c3 c4; do
  case $wl in c4) S="--steps 30 --warmup 3";; c3) S="--steps 1000 --warmup 50";; *) S="--steps 5000 --warmup 200";; esac
  python bench.py --workload $wl $S --secondary none --no-host-api > $O/${TAG}_bench_$wl.json 2>/dev/null
done
for wl in c2 c2m c2p c2x8 c2px8 c2x32 c3 c4 bm25; do
  bash profiles/collect.sh $wl $TAG > /dev/null 2>&1
done
ls -la $O | tail -40
