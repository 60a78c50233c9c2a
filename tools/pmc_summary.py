#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for each kernel.
usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
filt = sys.argv[2:]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "")
            if filt and not any(s in k for s in filt):
                continue
            short = k.split("(")[0][-60:]
            a = acc[(short, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-62s %-36s mean %.6g  (n=%d)" % (k, c, s / n, n))
