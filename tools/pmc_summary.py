#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc CSV (tools/pmc_pass.sh): usage pmc_summary.py <dir> [kernel-substring ...]
Prints, per kernel name matching a substring (all if none), dispatch count and the mean of every counter per dispatch."""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:]
f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if subs and not any(s in k for s in subs):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
out = {}
for k in acc:
    out[k] = {"dispatches": max(cnt[k].values()), "avg": {c: acc[k][c] / cnt[k][c] for c in acc[k]}}
print(json.dumps(out, indent=1))
