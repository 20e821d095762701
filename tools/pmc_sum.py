"""Sums the counters of a rocprofv3 --pmc run per kernel (prefix match) -- run on the GPU box:
    python tools/pmc_sum.py <dir with pmc_counter_collection.csv files> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict
base, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
sums = defaultdict(float); calls = defaultdict(int)
for path in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            sums[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
for k in sorted(sums):
    print(f"{k:40s} {sums[k]:.6g}  ({calls[k]} rows)")
