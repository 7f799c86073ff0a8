#!/bin/bash
# GPU box: the whole parity suite, a short bench, alone / loaded phase times.
mkdir -p gpurun_out
TAG=${1:-f}
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=6 > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -4 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_$TAG.log | head -20; exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --profile-kernel none --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
rc=$?; echo "[bench] rc=$rc"; tail -1 gpurun_out/bench_$TAG.log | cut -c1-200
if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_$TAG.log; exit $rc; fi
timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_phases.txt 2>&1 || exit 1
timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_phases_loaded.txt 2>&1 || exit 1
grep -A23 "under load" gpurun_out/${TAG}_phases_loaded.txt
grep -A23 "step 20" gpurun_out/${TAG}_phases.txt | grep "total"
