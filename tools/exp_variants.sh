#!/bin/bash
# usage: tools/exp_variants.sh <workers> <lib-suffix>...   (kernel experiments; libs built with SLIMT_HIP_LIB)
W=$1; shift
for v in "$@"; do
  lib=/root/repo/slimt_amd/lib/exp_$v.so
  [ "$v" = base ] && lib=/root/repo/slimt_amd/lib/libslimt_hip.so
  for w in $W; do
    echo "$v workers $w: $(SLIMT_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 96 --warmup 16 --workers $w --profile-kernel none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])[\"value\"])")" || exit 1
  done
done
