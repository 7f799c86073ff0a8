#!/bin/bash
# GPU box: the narrow cache form at D = 512 (base): parity, then A/B against the 24-bit form with the same library
mkdir -p gpurun_out
TAG=${1:-base20}
timeout -k 10 900 python -m pytest tests/test_gpu_kv_narrow.py -m gpu -q -x > gpurun_out/test_${TAG}_narrow.log 2>&1
rc=$?; echo "[narrow tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}_narrow.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}_narrow.log | head -20; exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "packed_kv or baseline_config or depths" > gpurun_out/test_${TAG}.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_${TAG}.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_${TAG}.log | head -20; exit $rc; fi
bash tools/ab_args.sh $TAG "--kv-format 2" "--kv-format 0" "--preset base" || exit 1
