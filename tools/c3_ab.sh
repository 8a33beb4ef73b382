#!/bin/bash
# A/B timing of ls_gemm.hip compile-time variants on the GPU box (main-kernel time only matters;
# ablations produce garbage results, so verification is off):
#   tools/c3_ab.sh "" "-DLS_GEMM_ABL_NOCHECK" "-DLS_GEMM_ABL_NOSTAGE" ...
cd "$(dirname "$0")/.."
NOPASS=${NOPASS:-}
for v in "$@"; do
  (cd lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $v" >/dev/null 2>&1)
  env ${NOPASS:+LS_GEMM_ABL_NOPASS=1} python bench.py --workload c3 --steps 300 --warmup 30 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'us/batch', round(d['ms_per_step']*1e3,1), 'main_us', round(r['kernel_ms']*1e3,1), 'frac', round(r['frac'],3))"
done
(cd lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s >/dev/null 2>&1)
