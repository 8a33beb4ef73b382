#!/bin/bash
# Config 3 batches queued ASYNC on ONE stream (no chain, no events): the kernel boundaries of one batch as a trace.
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tla
rocprofv3 --kernel-trace --output-format csv -d /tmp/tla -o t -- python $R/tools/c3time.py > /tmp/tla.log 2>&1
python - <<'PY'
import csv, glob
rows = []
for fn in glob.glob("/tmp/tla/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    for k in ("prep_f16", "tau", "select"):
        if k in n: return k
    if "gemm_filter" in n: return "sample" if ", 1>(" in n else "pass"
    return n[:20]
sl = rows[-22:-2]
prev = None
for s, e, n in sl:
    print("   %-10s dur %7.1f us   gap before %6.1f us" % (short(n), (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
