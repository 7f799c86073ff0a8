#!/usr/bin/env python3
"""CU-occupancy estimate from a rocprofv3 kernel trace: time-average number of
workgroups resident, by kernel (grid size x overlap), over the busy span."""
import csv, glob, sys, collections
d = sys.argv[1]
tr = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(tr)))
ev = []
names = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    key = "decode" if "decode_fused" in n else ("encode" if "encode_fused" in n else "other")
    wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) if "Grid_Size_X" in r else 1
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, key, wg))
    names[key] += 1
# restrict to the steady window: middle 60% of decode launches
dec = sorted((s, e) for s, e, k, _ in ev if k == "decode")
lo, hi = dec[len(dec) // 5][0], dec[-len(dec) // 5][1]
acc = collections.defaultdict(float)
for s, e, k, wg in ev:
    a, b = max(s, lo), min(e, hi)
    if b > a:
        acc[k] += (b - a) * wg
        acc[k + "_kernels"] += (b - a)
span = hi - lo
print(f"window {span/1e6:.2f} ms; launches {dict(names)}")
for k in ("decode", "encode", "other"):
    print(f"  {k:7s}: avg concurrent kernels {acc[k + '_kernels']/span:6.2f}, avg workgroups in flight (upper bound) {acc[k]/span:8.1f}")
