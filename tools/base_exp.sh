#!/bin/bash
# usage: tools/base_exp.sh "<rpb> <rpb_ln>" ...   solo (1 worker) kernel times of the base encoder
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do set -- $cfg
  export SLIMT_EXP_RPB=$1 SLIMT_EXP_RPB_LN=$2
  rm -rf gpurun_out/prof_exp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_exp -- python3 bench.py --preset base --workers 1 --steps 8 --warmup 2 --profile-kernel none --no-cpu-baseline > gpurun_out/prof_exp.log 2>&1 || exit 1
  f=$(ls gpurun_out/prof_exp/*/*_kernel_stats.csv | head -1)
  echo "== rpb $1 rpb_ln $2"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_rows" in r["Name"] or "attention" in r["Name"]:
        print(f"  {r['Name'][17:60]:44s} {r['Calls']:>5} {float(r['AverageNs'])/1e3:8.2f}us")
PY
done
