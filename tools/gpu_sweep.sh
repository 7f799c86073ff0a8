#!/bin/bash
# GPU box: the parity suite, then tools/sweep.sh over the argument sets given.
# usage: tools/gpu_sweep.sh <tag> "<bench args>" ...
mkdir -p gpurun_out
TAG=$1; shift
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=6 > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_$TAG.log | head -20; exit $rc; fi
tools/sweep.sh "$@" > gpurun_out/sweep_$TAG.txt 2>&1
cat gpurun_out/sweep_$TAG.txt
