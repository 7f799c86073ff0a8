#!/usr/bin/env python3
"""Instruction kinds along a kernel's body, in bins of N lines of its assembly: tools/isa_phase_hist.py FILE.s MANGLED-SUBSTRING [N]
(where the matrix instructions, barriers, 16-byte stores and scratch accesses sit: which phase a spill belongs to)."""
import re, sys
src = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 500
s = next(i for i, l in enumerate(src) if re.match(r'^\S*%s\S*:' % re.escape(key), l))
e = next(i for i in range(s, len(src)) if src[i].startswith('.Lfunc_end'))
kinds = [("mfma_f32", r"v_mfma_f32"), ("mfma_i8", r"v_mfma_i32"), ("store_x4", r"buffer_store_dwordx4"), ("load_x4", r"buffer_load_dwordx4"),
         ("sc_load", r"scratch_load"), ("sc_store", r"scratch_store"), ("barrier", r"s_barrier"), ("branch_back", None)]
labels = {m.group(1): i for i, l in enumerate(src[s:e]) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
rows = {}
for i, l in enumerate(src[s:e]):
    b = i // N
    r = rows.setdefault(b, dict.fromkeys([k for k, _ in kinds], 0))
    for k, pat in kinds:
        if pat and re.search(pat, l): r[k] += 1
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(1), 1 << 30) < i: r["branch_back"] += 1
print("line   " + " ".join("%9s" % k for k, _ in kinds))
for b in sorted(rows):
    print("%6d " % (b * N) + " ".join("%9d" % rows[b][k] for k, _ in kinds))
