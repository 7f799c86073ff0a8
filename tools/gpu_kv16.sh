#!/bin/bash
# GPU box: the tight (16-bit) K/V cache form: parity, then A/B against the 20-bit form with the same library on this box
mkdir -p gpurun_out
TAG=${1:-kv16}
timeout -k 10 900 python -m pytest tests/test_gpu_kv_narrow.py -m gpu -q -x > gpurun_out/test_${TAG}_narrow.log 2>&1
rc=$?; echo "[narrow tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}_narrow.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}_narrow.log | head -20; exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "packed_kv or tall_encoder or depths or baseline_config or smoke or engine" > gpurun_out/test_${TAG}.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}.log | head -20; exit $rc; fi
bash tools/ab_args.sh $TAG "--kv-tight-limit 0" "" || exit 1
for f in 0 1; do
  SLIMT_KV_TIGHT=$f SLIMT_DECODE_MODE=2 timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_tight${f}_phases.txt 2>&1 || exit 1
  SLIMT_KV_TIGHT=$f timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_tight${f}_loaded.txt 2>&1 || exit 1
  echo "tight $f alone: $(grep 'step 20' gpurun_out/${TAG}_tight${f}_phases.txt | sed 's/.*total//') attn $(grep -A22 'step 20' gpurun_out/${TAG}_tight${f}_phases.txt | grep attention | awk '{printf "%s ", $2}')  loaded: $(grep total gpurun_out/${TAG}_tight${f}_loaded.txt | sed 's/.*total//') attn $(grep attention gpurun_out/${TAG}_tight${f}_loaded.txt | awk '{printf "%s ", $2}')" | tee -a gpurun_out/${TAG}_ab.txt
done
