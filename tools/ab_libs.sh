#!/bin/bash
# GPU box: several builds of the library on ONE box, round robin, a short bench line each.
# usage: tools/ab_libs.sh <tag> <rounds> <lib.so> ...   (paths relative to the repo; bench args in AB_ARGS)
mkdir -p gpurun_out
TAG=$1; ROUNDS=$2; shift 2
for rep in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    SLIMT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 30 --warmup 5 --profile-kernel none --no-cpu-baseline --forward-steps 0 $AB_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rep $(basename $lib): value %.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))" | tee -a gpurun_out/${TAG}_libs.txt || exit 1
  done
done
