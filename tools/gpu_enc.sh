#!/bin/bash
# GPU box: the encoders after a change: parity subset (every layer bit-exact, translate), one layer by phase, short bench.
# usage: tools/gpu_enc.sh <tag> [bench args]
mkdir -p gpurun_out
TAG=${1:-enc}
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "encoder or baseline_config or packed_kv or translate_tokens or edge_shapes or out_of_range" > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/test_$TAG.log | head -20; exit $rc; fi
timeout -k 10 120 python tools/encode_wide_phases.py 256 tiny11 > gpurun_out/${TAG}_encoder_phases.txt 2>&1 || exit 1
grep -A11 "layer 2" gpurun_out/${TAG}_encoder_phases.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --profile-kernel none --no-cpu-baseline --forward-steps 0 $2 > gpurun_out/bench_$TAG.log 2>&1
rc=$?; echo "[bench] rc=$rc"; tail -1 gpurun_out/bench_$TAG.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value %.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))"
