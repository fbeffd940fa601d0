"""Print the top rows of a rocprofv3 kernel_stats.csv: name, calls, average us, share."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[:n]:
    print(f'  {r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:10.1f} us {float(r["Percentage"]):6.2f} %')
