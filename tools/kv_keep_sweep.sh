#!/bin/bash
# GPU box: which launches keep their K/V caches temporal (SLIMT_KV_BY_LAUNCH = k of every 8; engine.cpp) with the narrow cache
# form, on ONE box, two rounds. usage: tools/kv_keep_sweep.sh <tag> [bench args]
mkdir -p gpurun_out
TAG=${1:-keep}; shift
OUT=gpurun_out/${TAG}.txt; : > $OUT
for rep in 1 2; do
  for k in default 6 7 8; do
    e="SLIMT_KV_BY_LAUNCH=$k"; [ $k = default ] && e="X=1"
    v=$(env $e timeout -k 10 200 python bench.py --steps 30 --warmup 5 --profile-kernel none --no-cpu-baseline --forward-steps 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))") || exit 1
    echo "k=$k: $v" | tee -a $OUT
  done
done
