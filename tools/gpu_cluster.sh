#!/bin/bash
# GPU box: cluster logits (decode mode 6): parity, then config 4 against the 32- and 16-sentence tilings on this box
mkdir -p gpurun_out
TAG=${1:-cluster}
timeout -k 10 900 python -m pytest tests/test_gpu_cluster_logits.py -m gpu -q -x > gpurun_out/test_${TAG}.log 2>&1
rc=$?; echo "[cluster tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}.log | head -20; tail -30 gpurun_out/test_${TAG}.log; exit $rc; fi
for rep in 1 2; do
for args in "--decode-mode 3" "--decode-mode 2" "--decode-mode 6" "--decode-mode 0"; do
  echo -n "[config 4 $args] "; timeout -k 10 300 python bench.py --steps 10 --warmup 2 --profile-kernel none --no-cpu-baseline --forward-steps 0 --batch 512 --shortlist 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))" || exit 1
done; done | tee gpurun_out/${TAG}_c4.txt
SLIMT_DECODE_MODE=6 timeout -k 10 200 python tools/decode_phases.py 512 32 0 2>&1 | grep -A26 "step 20" | tee gpurun_out/${TAG}_phases_mode6.txt
SLIMT_DECODE_MODE=6 timeout -k 10 200 python tools/decode_phases_loaded.py 20 512 32 0 2>&1 | tail -25 | tee gpurun_out/${TAG}_loaded_mode6.txt
