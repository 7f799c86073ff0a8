#!/bin/bash
# GPU box: A/B of two builds of the library on ONE box over several bench argument sets, A B A B per set.
# usage: tools/ab_lib_args.sh <tag> <libA.so> <libB.so> "<args 1>" "<args 2>" ...
mkdir -p gpurun_out
TAG=$1; A=$2; B=$3; shift 3
for args in "$@"; do
  for rep in 1 2; do
    for v in A B; do
      lib=$A; [ $v = B ] && lib=$B
      SLIMT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --forward-steps 0 --sustained-steps 10 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v$rep [$args] value %.2f M  sustained %.2f M  launch %.0f us' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6, d['roofline']['avg_launch_us']))" | tee -a gpurun_out/${TAG}_ab.txt || exit 1
    done
  done
done
