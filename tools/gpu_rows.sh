#!/bin/bash
# GPU box: the occupancy-adaptive decoder (4 / 8 / 16 sentences per workgroup): parity subset, phase times per
# tiling, then the shapes it is for with the adaptive choice off and on. usage: tools/gpu_rows.sh <tag>
mkdir -p gpurun_out
TAG=${1:-rows}
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "packed_kv or admission or baseline_config or translate_tokens or output_layer_tile or edge_shapes" > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_$TAG.log | head -20; exit $rc; fi
for mode in 2 4 5; do
  SLIMT_DECODE_MODE=$mode timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_phases_mode$mode.txt 2>&1 || exit 1
  echo "mode $mode: $(grep 'step 20' gpurun_out/${TAG}_phases_mode$mode.txt) attention $(grep -A23 'step 20' gpurun_out/${TAG}_phases_mode$mode.txt | grep attention | tr -s ' ' | cut -d' ' -f3 | tr '\n' ' ')"
done
for ad in 0 1; do
  echo "== adaptive rows $ad"
  tools/sweep.sh "--adaptive-rows $ad --forward-steps 8 --sustained-steps 0 --steps 10" "--adaptive-rows $ad --forward-steps 0 --batch 64" \
     "--adaptive-rows $ad --forward-steps 0 --batch 64 --src-len 128" "--adaptive-rows $ad --forward-steps 0 --preset base --workers 8" \
     "--adaptive-rows $ad --forward-steps 0 --batch 128 --workers 8" || exit 1
done > gpurun_out/${TAG}_configs.txt 2>&1
cat gpurun_out/${TAG}_configs.txt
