cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kn
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kn -o t -- python /root/repo/bench.py --workload c4 --c4-rows 2000000 --steps 10 --warmup 2 --no-cpu-baseline --no-verify >/dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/kn/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>5: print("   ", r["Name"][:56], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
