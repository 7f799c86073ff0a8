#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) accesses sit? tools/isa_scratch_map.py FILE.s MANGLED-SUBSTRING
Prints the scratch instructions of the first kernel whose mangled name contains the substring, and for each the innermost
loop (by backward branch) it lies in -- a spill outside the step loop costs nothing per step."""
import re, sys
src = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*%s\S*): ' % re.escape(key), src, re.M)
name = m.group(1)
i = m.start()
j = src.index('.Lfunc_end', i)
body = src[i:j].splitlines()
labels = {}
for k, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = k
loops = []
for k, l in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        loops.append((labels[mm.group(1)], k))
sc = [(k, l.strip()) for k, l in enumerate(body) if re.search(r'\bscratch_(load|store)', l)]
print(name)
print(len(body), "lines,", len(sc), "scratch instructions,", len(loops), "loops; the largest:", sorted(loops, key=lambda x: x[0] - x[1])[:3])
for k, l in sc:
    inside = [lp for lp in loops if lp[0] <= k <= lp[1]]
    inner = min(inside, key=lambda x: x[1] - x[0]) if inside else None
    print("%6d  %-60s  %s" % (k, l[:60], "in loop %s (%d lines)" % (inner, inner[1] - inner[0]) if inner else "outside every loop"))
