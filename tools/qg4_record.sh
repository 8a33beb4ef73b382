#!/bin/bash
# Round 5, verdict item 1, the REAL shape: config 3 through the row-split, 64-queries-per-wave pass
# (ls_gemm_filter_rs2_kernel, ls_debug_option 18) against the shipped shape on one box: bench timing with board
# power / clocks sampled by rocm-smi, then the instruction mix of the pass kernel by rocprofv3 --pmc.
#   gpurun -- 'bash tools/qg4_record.sh > gpurun_out/r05_qg4_record.txt 2>&1'
R=$(cd "$(dirname "$0")/.." && pwd)
A="--workload c3 --steps 1000 --warmup 20 --secondary none --no-host-api --no-cpu-baseline"
for v in default qg4 default qg4; do
  X=""; [ $v = qg4 ] && X="--lib-option 18=1"
  rm -f /tmp/smi_$v.txt
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null >> /tmp/smi_$v.txt; echo >> /tmp/smi_$v.txt; sleep 0.1; done ) &
  SMI=$!
  python $R/bench.py $A $X 2>/dev/null | tail -1 > /tmp/q4_$v.json
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$v" <<'PY'
import json, sys, re
v = sys.argv[1]
j = json.loads(open(f"/tmp/q4_{v}.json").read()); r = j["roofline"]
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
      f"(frac {r['frac_whole_batch']:.4f}) recall {j['recall_at_k']} repaired {j['repaired_queries']} | power {top(pw):.0f} W, sclk {top(sclk):.0f} MHz", flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp
for v in default qg4; do
  X=""; [ $v = qg4 ] && X="--lib-option 18=1"
  rm -rf /tmp/q4_pmc_$v
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d /tmp/q4_pmc_$v -o p -- python $R/bench.py $A $X --steps 100 --no-verify > /tmp/q4_pmc_$v.log 2>&1
  python - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
acc = {}
for f in glob.glob(f"/tmp/q4_pmc_{v}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row.get("Kernel_Name", "")
        if ("ls_gemm_filter_rs2_kernel" in n if v == "qg4" else "ls_gemm_filter_kernel" in n) and ", 0>(" in n:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in acc.items()}
if m:
    print(f"{v:8s}: per pass launch " + ", ".join(f"{k} {int(x):,}" for k, x in sorted(m.items())) +
          f" | LDS/MFMA {m['SQ_INSTS_LDS']/m['SQ_INSTS_MFMA']:.3f} | matrix pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(m['GRBM_GUI_ACTIVE']/8*1024):.3f}", flush=True)
PY
done
