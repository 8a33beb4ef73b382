#!/bin/bash
# Config 4's MFMA pass against its two halves on ONE box, with the board's power and clocks sampled while it runs:
#   full kernel | MFMA stream alone (no tile DMA: -DLS_ABL_NODMA=1)
# build first:  make -C lean-explore_amd/csrc variant NAME=nodma VFLAGS="-DLS_ABL_NODMA=1 -DLS_ABL_NOREPAIR=1"
# gpurun -- 'bash tools/c4_bound.sh'
R=$(cd "$(dirname "$0")/.." && pwd)
for v in base nodma base nodma; do
  if [ $v = base ]; then unset LEANSEARCH_LIB; else export LEANSEARCH_LIB=$R/lean-explore_amd/variants/libleansearch_$v.so; fi
  rm -f /tmp/smi_$v.txt
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null >> /tmp/smi_$v.txt; echo >> /tmp/smi_$v.txt; sleep 0.15; done ) &
  SMI=$!
  python $R/bench.py --workload c4 --steps 120 --warmup 3 --secondary none --no-host-api --no-cpu-baseline --no-verify 2>/dev/null | tail -1 > /tmp/c4_$v.json
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$v" <<'PY'
import json, sys, re
v = sys.argv[1]
j = json.loads(open(f"/tmp/c4_{v}.json").read()); r = j["roofline"]
pw, sclk, mclk = [], [], []
for ln in open(f"/tmp/smi_{v}.txt"):
    ln = ln.strip()
    if not ln.startswith("{"): continue
    try: d = json.loads(ln)
    except Exception: continue
    c = d.get("card0", {})
    for k, val in c.items():
        m = re.search(r"([\d.]+)", str(val))
        if not m: continue
        x = float(m.group(1))
        if "Power" in k and "Socket" in k or ("Power" in k and "Average" in k): pw.append(x)
        if k.startswith("sclk"): sclk.append(x)
        if k.startswith("mclk"): mclk.append(x)
top = lambda a: (sum(sorted(a)[len(a)//2:]) / max(1, len(a) - len(a)//2)) if a else float("nan")  # upper half: the loaded samples
print(f"{v:7s}: pass kernel {r['kernel_ms']*1e3:8.1f} us  (HBM frac {r['frac']:.3f}, MFMA frac {r['mfma_frac']:.3f}; whole batch {j['ms_per_step']*1e3:.1f} us) | "
      f"power {top(pw):.0f} W, sclk {top(sclk):.0f} MHz, mclk {top(mclk):.0f} MHz over the loaded half of {len(pw)} samples")
PY
done
