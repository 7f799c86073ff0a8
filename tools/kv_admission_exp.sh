#!/bin/bash
# GPU box: decoders in flight against the Infinity Cache. For a decoder budget (workgroups admitted
# at a time) and a tiling (decode mode 2 = 16, 4 = 8 sentences per workgroup): throughput of the headline workload, the
# loaded step's attention phases, and the decoder's FETCH_SIZE per launch (a separate rocprofv3 --pmc pass).
# usage: tools/kv_admission_exp.sh <tag> "<budget> <mode>" ...
mkdir -p gpurun_out
TAG=${1:-kvadm}; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun}"
OUT=gpurun_out/${TAG}_kv_admission.txt; : > $OUT
for cfg in "$@"; do
  set -- $cfg; BUD=$1; MODE=$2
  ARGS="--decoder-budget $BUD --decode-mode $MODE --forward-steps 0 --sustained-steps 0 --no-cpu-baseline"
  line=$(timeout -k 10 200 python bench.py --steps 16 --warmup 4 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.2f M tok/s, decoder launch %.0f us, %.1f in flight' % (d['value']/1e6, r['avg_launch_us'], r['launches_in_flight']))") || { echo "bench failed: $cfg"; exit 1; }
  ph=$(SLIMT_DECODER_BUDGET=$BUD SLIMT_DECODE_MODE=$MODE timeout -k 10 120 python tools/decode_phases_loaded.py 2>/dev/null | grep -E "total|attention" | tr -s ' ' | tr '\n' ';') || { echo "phases failed: $cfg"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_${BUD}_$MODE -- python3 bench.py $ARGS --steps 4 --warmup 1 --profile-kernel none > gpurun_out/pmc_${TAG}_${BUD}_$MODE.log 2>&1 || { echo "pmc failed: $cfg"; tail -3 gpurun_out/pmc_${TAG}_${BUD}_$MODE.log; exit 1; }
  f=$(ls -t gpurun_out/pmc_${TAG}_${BUD}_$MODE/*/*counter_collection.csv | head -1)
  fetch=$(python3 - "$f" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "decode_fused" in r["Kernel_Name"]]
print("FETCH_SIZE %.0f MB per decoder launch (x2 = %.0f MB of traffic, gfx950 correction; %d launches)" % (sum(v) / len(v) / 1024, 2 * sum(v) / len(v) / 1024, len(v)))
PY
)
  echo "budget $BUD, decode mode $MODE: $line; $fetch; loaded step: $ph" | tee -a $OUT
  rm -rf gpurun_out/pmc_${TAG}_${BUD}_$MODE
done
