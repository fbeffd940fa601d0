#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats CSV (…_kernel_stats.csv)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:n]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} "
          f"total_ms={float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}%")
