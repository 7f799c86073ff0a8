#!/bin/bash
# GPU box: A/B of two builds of the library on the SAME box (box-to-box differences are as large as most single changes):
# encoder layer by phase and a short bench, A B A B. usage: tools/ab_lib.sh <tag> <libA.so> <libB.so> [bench args]
mkdir -p gpurun_out
TAG=$1; A=$2; B=$3
for rep in 1 2; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    SLIMT_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/encode_wide_phases.py 256 tiny11 > gpurun_out/${TAG}_${v}${rep}_enc.txt 2>&1 || exit 1
    echo "$v$rep encoder: $(grep 'layer 2' gpurun_out/${TAG}_${v}${rep}_enc.txt)  $(grep -A10 'layer 2' gpurun_out/${TAG}_${v}${rep}_enc.txt | grep -v layer | awk '{printf "%s ", $(NF-1)}')"
    if [ -n "$AB_DECODE" ]; then  # the decoder step by phase, alone (16 sentences per workgroup) and under the 20-worker load
      SLIMT_DECODE_MODE=2 SLIMT_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_${v}${rep}_dec.txt 2>&1 || exit 1
      SLIMT_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_${v}${rep}_decl.txt 2>&1 || exit 1
      echo "$v$rep decoder alone: $(grep 'step 20' gpurun_out/${TAG}_${v}${rep}_dec.txt | sed 's/.*total//')  loaded: $(grep 'total' gpurun_out/${TAG}_${v}${rep}_decl.txt | sed 's/.*total//')"
    fi
    SLIMT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 30 --warmup 5 --profile-kernel none --no-cpu-baseline --forward-steps 0 $4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v$rep bench: value %.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))" || exit 1
  done
done
