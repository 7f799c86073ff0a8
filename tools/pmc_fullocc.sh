#!/bin/bash
# PMC of ONE decode launch at full occupancy (B=4096 -> 256 workgroups): fabric-side
# fetch bytes and L2 hit/miss requests. Separate passes per counter set.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmc_full_$tag
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_full_$tag -- python3 bench.py --batch ${BATCH:-4096} --workers 1 --steps 2 --warmup 1 --no-cpu-baseline --profile-kernel none > gpurun_out/pmc_full_$tag.log 2>&1 || { echo "pmc $c failed"; tail -5 gpurun_out/pmc_full_$tag.log; exit 1; }
  f=$(ls gpurun_out/pmc_full_$tag/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "decode_fused" if "decode_fused" in n else "encode_fused" if "encode_fused" in n else None
    if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:14s} {c:20s} launches {len(v):3d} avg {sum(v)/len(v):.4g}")
PY
done
