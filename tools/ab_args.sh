#!/bin/bash
# GPU box: A/B of two bench argument sets with ONE library on ONE box, A B A B (+ the decoder step by phase under load).
# usage: tools/ab_args.sh <tag> "<args A>" "<args B>" [common bench args]      e.g. "--kv-format 2" "--kv-format 0"
mkdir -p gpurun_out
TAG=$1; A=$2; B=$3; COMMON=$4
for rep in 1 2; do
  for v in A B; do
    args=$A; [ $v = B ] && args=$B
    timeout -k 10 300 python bench.py --steps 30 --warmup 5 --profile-kernel none --no-cpu-baseline --forward-steps 0 $COMMON $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v$rep [$args] value %.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))" | tee -a gpurun_out/${TAG}_ab.txt || exit 1
  done
done
