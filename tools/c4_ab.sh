#!/bin/bash
# A/B of ls_gemm.hip compile-time variants on config 4's per-GPU shape (reduced rows)
cd "$(dirname "$0")/.."
for v in "$@"; do
  (cd lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $v" >/dev/null 2>&1)
  env ${NOPASS:+LS_GEMM_ABL_NOPASS=1} python bench.py --workload c4 --c4-rows ${C4_ROWS:-4000000} --steps 20 --warmup 3 --no-cpu-baseline ${NOPASS:+--no-verify} 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v] ms/batch', round(d['ms_per_step'],3), 'main_ms', round(r['kernel_ms'],3), 'hbm', round(r['frac'],3), 'mfma', r['mfma_frac'], 'recall', d['recall_at_k'])"
done
(cd lean-explore_amd/csrc && rm -f _build/ls_gemm.o && make -s >/dev/null 2>&1)
