"""Time line of ONE training step out of a rocprofv3 kernel trace (kernel_trace.csv): every kernel between two consecutive launches of the
optimizer kernel (k_adamw), start / end relative to the step's first kernel, duration and stream.  usage: step_timeline.py trace.csv [step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
a, b = ad[which - 1] + 1, ad[which]
t0 = int(rows[a]["Start_Timestamp"])
print(f"step: {b - a + 1} kernels, {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us from the first start to the end of k_adamw")
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3s}  {name}")
