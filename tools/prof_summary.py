#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats output dir: per-kernel stats and
the GPU busy timeline (union of kernel intervals)."""
import csv, glob, sys
d = sys.argv[1]
st = glob.glob(d + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(st)))[:10]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6} {float(r['AverageNs'])/1e3:10.2f}us {r['Percentage']:>6}%")
tr = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(tr)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0, t1 = iv[0][0], max(e for _, e in iv)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"span {1e-6*(t1-t0):.2f} ms, union-busy {1e-6*busy:.2f} ms, sum of kernel time {1e-6*sum(e-s for s,e in iv):.2f} ms, {len(iv)} dispatches")
