#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv as a table:  tools/kstats.py file.csv [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:9.3f} avg_us {float(r['AverageNs']) / 1e3:9.1f} pct {r['Percentage']}")
