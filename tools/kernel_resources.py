#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print per-kernel VGPR/SGPR/LDS/spill."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
       "-c", "-x", "hip", src, "-I", os.path.join(ROOT, "slimt_amd/csrc"), "-I", os.path.join(ROOT, "include"),
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("SLIMT_HIPCC_EXTRA", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if "error" in line: print(line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
print(f"{'kernel':92s} VGPR AGPR SGPR scratch vspill sspill LDS occ")
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"slimt_hip::|\(.*\)|void ", "", name)
    print(f"{name[:92]:92s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('SGPRs','?'):>4} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('VGPRs Spill','?'):>6} {r.get('SGPRs Spill','?'):>6} {r.get('LDS Size [bytes/block]','?'):>5} {r.get('Occupancy [waves/SIMD]','?'):>3}")
