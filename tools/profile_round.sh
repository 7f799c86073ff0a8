#!/bin/bash
# GPU box: the round's evidence for profiles/. Usage: tools/profile_round.sh <tag>
#   1. bench.py default run (with cpu_baseline)          -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same run  -> gpurun_out/<tag>_kernel_stats.csv
#   3. PMC passes (separate runs): FETCH_SIZE, WRITE_SIZE -> gpurun_out/<tag>_pmc_{fetch,write}.csv
# Each step is bounded; a failed or timed-out step stops the chain.
TAG=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py ${NO_CPU:+--no-cpu-baseline} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1 || { echo "rocprof stats failed"; tail -5 gpurun_out/prof_$TAG.log; exit 1; }
f=$(ls gpurun_out/prof_$TAG/*/*_kernel_stats.csv | head -1); cp "$f" gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200 > gpurun_out/${TAG}_bench_under_rocprof.txt
head -4 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
# PMC passes (each counter in its own run, no trace domains). usage: pmc_passes <name> <bench args...>
pmc_passes() {
  local NAME=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${NAME}_$c -- python3 bench.py "$@" --steps 4 --warmup 1 --sustained-steps 0 --no-cpu-baseline --profile-kernel none > gpurun_out/pmc_${NAME}_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 gpurun_out/pmc_${NAME}_$c.log; return 1; }
    f=$(ls -t gpurun_out/pmc_${NAME}_$c/*/*counter_collection.csv | head -1)
    python3 - "$f" "$c" gpurun_out/${NAME}_pmc_$c.json <<'PY'
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != sys.argv[2]: continue
    name = r["Kernel_Name"]
    key = next((k for k in ("decode_fused", "encode_fused", "encode_tall", "encode_wide") if k in name), None)
    if key: agg[key].append(float(r["Counter_Value"]))
out = {k: {"counter": sys.argv[2], "launches": len(v), "avg_KB_per_launch": sum(v) / len(v)} for k, v in agg.items()}
json.dump(out, open(sys.argv[3], "w"), indent=1); print(out)
PY
  done
}
pmc_passes $TAG || exit 1
if [ -n "$WITH_BASE" ]; then
  timeout -k 10 300 python bench.py --preset base --no-cpu-baseline > gpurun_out/${TAG}_base_bench.json 2> gpurun_out/${TAG}_base_bench.err || { echo "base bench failed"; exit 1; }
  cut -c1-300 gpurun_out/${TAG}_base_bench.json
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_base -- python3 bench.py --preset base --no-cpu-baseline > gpurun_out/prof_${TAG}_base.log 2>&1 || { echo "rocprof base failed"; exit 1; }
  f=$(ls -t gpurun_out/prof_${TAG}_base/*/*_kernel_stats.csv | head -1); cp "$f" gpurun_out/${TAG}_base_kernel_stats.csv
  head -4 gpurun_out/${TAG}_base_kernel_stats.csv | cut -c1-160
  pmc_passes ${TAG}_base --preset base || exit 1
  timeout -k 10 120 python tools/encode_wide_phases.py > gpurun_out/${TAG}_base_encoder_phases.txt 2>&1 || exit 1
fi
