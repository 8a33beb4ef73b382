cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o t -- python /root/repo/bench.py --workload c1 --no-cpu-baseline > /tmp/b1.log 2>&1
grep "^{" /tmp/b1.log | cut -c1-260
python - <<PY
import csv,glob
f=glob.glob("/tmp/p1/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>50: print("   ", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
