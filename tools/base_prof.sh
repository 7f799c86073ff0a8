cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_base -- python3 bench.py --preset base --workers ${WORKERS:-16} --steps 16 --warmup 4 --profile-kernel none --no-cpu-baseline > gpurun_out/prof_base.log 2>&1
f=$(ls gpurun_out/prof_base/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6} {float(r['AverageNs'])/1e3:9.2f}us {r['Percentage']:>6}%")
PY
