#!/usr/bin/env python3
"""average rocprofv3 --pmc counter values per kernel name:  tools/pmc_table.py file_counter_collection.csv ..."""
import csv, sys, collections
for f in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "igemm" not in k and "wgrad" not in k:
            continue
        print(k)
        for c, v in d.items():
            print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
