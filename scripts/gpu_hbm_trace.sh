#!/bin/bash
# kernel-level (rocprofv3 trace) durations of the HBM-bound kernels per shape: the event-bracketed Python loop of
# hbm_kernels_bench.py is host-bound below ~30 us per call
OUT=$GRAFT_REPO_ROOT/gpurun_out/hbm_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o h -- python $GRAFT_REPO_ROOT/scripts/hbm_kernels_bench.py > /dev/null 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/h_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith(("k_fft", "k_sde_step")):
        d[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print(f"{k[0]:40s} grid {k[1]:>9s} x {k[2]:>3s} wg {k[3]:>4s}  n={len(v):3d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
PY
