#!/bin/bash
# GPU box: the narrow cache form for 33..128-token sentences: parity, then A/B against the 24-bit form (one library, one box)
mkdir -p gpurun_out
TAG=${1:-long20}
timeout -k 10 900 python -m pytest tests/test_gpu_kv_narrow.py -m gpu -q -x > gpurun_out/test_${TAG}.log 2>&1
rc=$?; echo "[narrow tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}.log | head -20; exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "packed_kv or medium_sentences or depths" > gpurun_out/test_${TAG}_b.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}_b.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}_b.log | head -20; exit $rc; fi
bash tools/ab_args.sh ${TAG}_S128 "--kv-format 2" "--kv-format 0" "--batch 64 --src-len 128" || exit 1
bash tools/ab_args.sh ${TAG}_S64 "--kv-format 2" "--kv-format 0" "--batch 128 --src-len 64" || exit 1
