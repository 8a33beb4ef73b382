#!/bin/bash
# Timeline of config 3's batches on the two internal lanes: what runs between the end of one MFMA pass and the
# start of the next.   gpurun -- 'bash tools/c3_timeline.sh'
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/bench.py --workload c3 --steps 200 --warmup 20 --secondary none --no-host-api --no-cpu-baseline --no-verify $LS_BENCH_EXTRA > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob, statistics as st
rows = []
for fn in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
def short(n):
    for k in ("prep_f16", "tau", "select"):
        if k in n: return k
    if "gemm_filter" in n: return "sample" if ", 1>(" in n else "pass"  # ", 0>" pass, ", 2>" pass + a later batch's sample phase
    return None
ev = [(s, e, short(n), q) for s, e, n, q in rows if short(n)]
passes = [x for x in ev if x[2] == "pass"][-150:]
gaps, rep = [], []
for a, b in zip(passes[:-1], passes[1:]):
    gap = b[0] - a[1]
    gaps.append(gap)
    inside = [(x[2], (x[0] - a[1]) / 1e3, (x[1] - a[1]) / 1e3) for x in ev if x[1] > a[1] - 30000 and x[0] < b[0] and x[2] != "pass"]
    rep.append(inside)
print("pass duration median %.1f us; end-of-pass -> start-of-next-pass gap median %.1f us (p10 %.1f, p90 %.1f); period %.1f us" % (
    st.median([(p[1] - p[0]) / 1e3 for p in passes]), st.median(gaps) / 1e3, sorted(gaps)[len(gaps)//10] / 1e3,
    sorted(gaps)[9*len(gaps)//10] / 1e3, st.median([(b[0] - a[0]) / 1e3 for a, b in zip(passes[:-1], passes[1:])])))
print("three consecutive gaps (kernel, start, end in us relative to the end of the earlier pass):")
for inside in rep[60:63]:
    print("   " + "; ".join("%s %.1f..%.1f" % x for x in sorted(inside, key=lambda t: t[1])))
PY
python - <<'PY'
# raw slice: every kernel of ~3 batches with its queue (one HIP stream = one queue id here)
import csv, glob
rows = []
for fn in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
def short(n):
    for k in ("prep_f16", "tau", "select"):
        if k in n: return k
    if "gemm_filter" in n: return "sample" if ", 1>(" in n else ("pass+sample" if ", 2>(" in n else "pass")
    return n[:24]
sl = rows[-60:-30]
t0 = sl[0][0]
print("raw slice (us from the first kernel shown): start..end kernel [queue/stream]")
for s, e, n, q, st in sl:
    print("   %8.1f .. %8.1f  %-12s [q %s / s %s]" % ((s - t0) / 1e3, (e - t0) / 1e3, short(n), q, st))
PY
