#!/bin/bash
# GPU-box check: engine parity tests, then bench, then rocprofv3 kernel stats.
# Each step is bounded; a timed-out step stops the chain (no retry).
mkdir -p gpurun_out
TAG=${1:-run}
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=6 > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -4 gpurun_out/test_$TAG.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in tests"; exit 99; fi
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_$TAG.log | head -20; exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --profile-kernel none --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
rc=$?; echo "[bench] rc=$rc"; tail -1 gpurun_out/bench_$TAG.log | cut -c1-260
if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_$TAG.log; exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 1 --profile-kernel none --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
rc=$?; echo "[rocprof] rc=$rc"
f=$(ls gpurun_out/prof_$TAG/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6} {float(r['AverageNs'])/1e3:9.2f}us {r['Percentage']:>6}%")
PY
exit 0
