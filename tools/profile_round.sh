#!/bin/bash
# GPU box: the round's evidence for profiles/. Usage: tools/profile_round.sh <tag>
#   1. bench.py default run (with cpu_baseline)          -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same run  -> gpurun_out/<tag>_kernel_stats.csv
#   3. PMC passes (separate runs): FETCH_SIZE, WRITE_SIZE -> gpurun_out/<tag>_pmc_{fetch,write}.csv
# Each step is bounded; a failed or timed-out step stops the chain.
TAG=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1 || { echo "rocprof stats failed"; tail -5 gpurun_out/prof_$TAG.log; exit 1; }
f=$(ls gpurun_out/prof_$TAG/*/*_kernel_stats.csv | head -1); cp "$f" gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200 > gpurun_out/${TAG}_bench_under_rocprof.txt
head -4 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${TAG}_$c -- python3 bench.py --steps 4 --warmup 1 --sustained-steps 0 --no-cpu-baseline --profile-kernel none > gpurun_out/pmc_${TAG}_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 gpurun_out/pmc_${TAG}_$c.log; exit 1; }
  f=$(ls gpurun_out/pmc_${TAG}_$c/*/*counter_collection.csv | head -1)
  python3 - "$f" "$c" gpurun_out/${TAG}_pmc_$c.json <<'PY'
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != sys.argv[2]: continue
    name = r["Kernel_Name"]
    key = "decode_fused" if "decode_fused" in name else "encode_fused" if "encode_fused" in name else None
    if key: agg[key].append(float(r["Counter_Value"]))
out = {k: {"counter": sys.argv[2], "launches": len(v), "avg_KB_per_launch": sum(v) / len(v)} for k, v in agg.items()}
json.dump(out, open(sys.argv[3], "w"), indent=1); print(out)
PY
done
