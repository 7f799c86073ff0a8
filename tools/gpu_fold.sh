#!/bin/bash
# GPU box: the shortlist generator inside the encoder launch: parity subset, then model_forward_per_batch_shortlist with
# the generator as its own launch (SLIMT_SHORTLIST_FOLD=0) and inside the encoder's. usage: tools/gpu_fold.sh <tag>
mkdir -p gpurun_out
TAG=${1:-fold}
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "shortlist or generated or batcher or host_pipeline or text_to_text or host_cpp" > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_$TAG.log | head -20; exit $rc; fi
for fold in 0 1 0 1; do
  echo -n "fold=$fold "
  SLIMT_SHORTLIST_FOLD=$fold tools/sweep.sh "--forward-steps 20 --sustained-steps 0 --steps 10" || exit 1
done > gpurun_out/${TAG}_forward.txt 2>&1
for fold in 0 1; do
  echo -n "base fold=$fold "
  SLIMT_SHORTLIST_FOLD=$fold tools/sweep.sh "--preset base --forward-steps 12 --sustained-steps 0 --steps 8" || exit 1
done >> gpurun_out/${TAG}_forward.txt 2>&1
cat gpurun_out/${TAG}_forward.txt
timeout -k 10 300 python tools/stress_generated.py 6 4 > gpurun_out/${TAG}_stress.txt 2>&1 || { echo "stress failed"; tail -5 gpurun_out/${TAG}_stress.txt; exit 1; }
timeout -k 10 300 python tools/stress_generated.py 4 3 base >> gpurun_out/${TAG}_stress.txt 2>&1 || { echo "stress (base) failed"; tail -5 gpurun_out/${TAG}_stress.txt; exit 1; }
grep mismatches gpurun_out/${TAG}_stress.txt
