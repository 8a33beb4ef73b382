#!/bin/bash
# Round 5, verdict item 1: what could a tile shape with fewer LDS reads per MFMA gain on config 3's pass?
# Timing ablations (wrong results by construction): the two-accumulator pass reads only 1/2 (lds2) or 1/4
# (lds4) of its A fragments from LDS and reuses the registers for the other row blocks - exactly the LDS
# traffic a "64 queries per wave" (QG = 4) geometry would have, every other instruction unchanged.
# Built on top of LS_ABL_NOPASS (nothing passes the filter: the reused fragments give wrong scores), so the
# baseline of the comparison is the `np` variant, not the shipped library:
#   make -C lean-explore_amd/csrc variant NAME=np     VFLAGS="-DLS_ABL_NOPASS=1 -DLS_ABL_NOREPAIR=1"
#   make -C lean-explore_amd/csrc variant NAME=lds2np VFLAGS="-DLS_ABL_LDSREADS=2 -DLS_ABL_NOPASS=1 -DLS_ABL_NOREPAIR=1"   (lds4np alike)
#   gpurun -- 'bash tools/tile_shape_ab.sh > gpurun_out/r05_tile_shape.txt 2>&1'
# Per variant: interleaved bench.py timing (pass kernel by dispatch-attached events) with board power and
# clocks sampled by rocm-smi while it runs, then one rocprofv3 --pmc pass for the instruction mix of the
# pass kernel (`, 0>(` instantiation).
R=$(cd "$(dirname "$0")/.." && pwd)
VARIANTS=${VARIANTS:-"np lds2np lds4np np lds2np lds4np default"}
A="--workload c3 --steps 300 --warmup 20 --secondary none --no-host-api --no-cpu-baseline --no-verify"
for v in $VARIANTS; do
  if [ $v = default ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$v.so; fi
  rm -f /tmp/smi_$v.txt
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null >> /tmp/smi_$v.txt; echo >> /tmp/smi_$v.txt; sleep 0.1; done ) &
  SMI=$!
  python $R/bench.py $A --steps 3000 2>/dev/null | tail -1 > /tmp/ts_$v.json
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$v" <<'PY'
import json, sys, re
v = sys.argv[1]
j = json.loads(open(f"/tmp/ts_{v}.json").read()); r = j["roofline"]
pw, sclk = [], []
for ln in open(f"/tmp/smi_{v}.txt"):
    ln = ln.strip()
    if not ln.startswith("{"): continue
    try: d = json.loads(ln)
    except Exception: continue
    for k, val in d.get("card0", {}).items():
        m = re.search(r"([\d.]+)", str(val))
        if not m: continue
        x = float(m.group(1))
        if "Power" in k and ("Socket" in k or "Average" in k): pw.append(x)
        if k.startswith("sclk"): sclk.append(x)
top = lambda a: (sum(sorted(a)[len(a)//2:]) / max(1, len(a) - len(a)//2)) if a else float("nan")
print(f"{v:8s}: pass kernel {r['kernel_ms']*1e3:7.2f} us (frac {r['frac']:.4f})  whole batch {j['ms_per_step']*1e3:7.2f} us "
      f"(frac {r['frac_whole_batch']:.4f}) | power {top(pw):.0f} W, sclk {top(sclk):.0f} MHz (loaded half of {len(pw)} samples)", flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp
for v in $(echo $VARIANTS | tr " " "\n" | awk '!s[$0]++'); do
  if [ $v = default ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$v.so; fi
  [ $v = default ] || [ -f "$LEANSEARCH_LIB" ] || continue
  rm -rf /tmp/ts_pmc_$v
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d /tmp/ts_pmc_$v -o p -- python $R/bench.py $A --steps 100 > /tmp/ts_pmc_$v.log 2>&1
  python - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
acc = {}
for f in glob.glob(f"/tmp/ts_pmc_{v}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row.get("Kernel_Name", "")
        if "ls_gemm_filter_kernel" in n and ", 0>(" in n:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in acc.items()}
if m:
    print(f"{v:8s}: per pass launch " + ", ".join(f"{k} {int(x):,}" for k, x in sorted(m.items())) +
          f" | LDS/MFMA {m['SQ_INSTS_LDS']/m['SQ_INSTS_MFMA']:.3f}, (VALU-MFMA)/MFMA {(m['SQ_INSTS_VALU']-m['SQ_INSTS_MFMA'])/m['SQ_INSTS_MFMA']:.2f} (if SQ_INSTS_VALU counts MFMAs) "
          f"| matrix pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(m['GRBM_GUI_ACTIVE']/8*1024):.3f}", flush=True)
PY
done
