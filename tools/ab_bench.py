#!/usr/bin/env python
"""Interleaved A/B timing of library variants on ONE box (box-to-box spread is ~5 %, larger than
most kernel changes):  python tools/ab_bench.py --workload c3 --libs default,ql0,r1 --rounds 3
Variants are built by `make -C lean-explore_amd/csrc variant NAME=... VFLAGS=...` into
lean-explore_amd/variants/; `default` is the shipped libleansearch.so. Each measurement is a
fresh bench.py process (LEANSEARCH_LIB selects the library)."""
import argparse
import json
import os
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--libs", default="default")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--c4-rows", type=int, default=0)
ap.add_argument("--extra", default="", help="extra bench.py arguments, e.g. --no-verify")
args = ap.parse_args()
libs = args.libs.split(",")
res = {l: [] for l in libs}
for r in range(args.rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["LEANSEARCH_LIB"] = str(ROOT / "lean-explore_amd" / "variants" / f"libleansearch_{l}.so")
        cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", args.workload, "--steps",
               str(args.steps), "--warmup", str(args.warmup), "--secondary", "none", "--no-host-api",
               "--no-cpu-baseline"]
        if args.c4_rows:
            cmd += ["--c4-rows", str(args.c4_rows)]
        cmd += args.extra.split()
        p = subprocess.run(cmd, env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", p.stderr[-500:])
            continue
        o = json.loads(line[-1])
        res[l].append((o["roofline"]["kernel_ms"], o["ms_per_step"], o["recall_at_k"], o.get("repaired_queries")))
        print(f"round {r} {l:12s} kernel_ms {o['roofline']['kernel_ms']:.4f}  ms_per_step {o['ms_per_step']:.4f} "
              f"recall {o['recall_at_k']} repaired {o['repaired_queries']}", flush=True)
print(json.dumps({l: {"kernel_ms_median": statistics.median(x[0] for x in v),
                      "ms_per_step_median": statistics.median(x[1] for x in v), "n": len(v)}
                  for l, v in res.items() if v}))
