#!/usr/bin/env python
"""Which kernels of the library use scratch memory (spills or run-time indexed private arrays)?  Measured on MI355X / ROCm 7
(rocprofv3 kernel times): k_sde_step 18.5 -> 5.8 us for 2.4 MB and 73 -> 59 us for 352 MB when two float4 tails stopped being
indexed with a run-time subscript (32 bytes of scratch per lane); k_attention_bf16 106.1 -> 104.9 us when its 14 spilled dwords
went away.  The kernels that are launched per layer or per step should report nothing here.
usage: python scripts/scratch_report.py [file.hip ...]   (default: every .hip of the library)"""
import glob, os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fourierdiffusion_amd", "csrc")
files = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(root, "*.hip")))
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-honor-nans", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                        "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"], capture_output=True, text=True, cwd=root)
    name, vg = None, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs: (\d+)", ln)
        if m:
            vg = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
        if m and int(m.group(1)) > 0:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(anonymous namespace\)::", "", dem)
            dem = re.sub(r"\(.*", "", dem)
            print(f"{os.path.basename(f):22s} {dem[:80]:80s} VGPRs {vg:4d} scratch {m.group(1):>5s} B/lane")
