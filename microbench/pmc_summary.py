#!/usr/bin/env python
"""Summarise rocprofv3 --pmc csv output: per kernel name, sum of each counter over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, cs in acc.items():
        if not any(t in k for t in ("lstm", "pack", "bwd", "head_", "reduce", "bbb", "adam", "ce_loss")):
            continue
        print(d, k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} {v:.6g}  (dispatches {n[(k, c)]})")
