#!/bin/bash
# usage: tools/exp_variants.sh <lib-suffix>...   (kernel experiments; libs built with SLIMT_HIP_LIB)
for v in "$@"; do
  lib=/root/repo/slimt_amd/lib/exp_$v.so
  [ "$v" = base ] && lib=/root/repo/slimt_amd/lib/libslimt_hip.so
  echo "== $v"
  SLIMT_HIP_LIB=$lib timeout -k 10 200 python tools/decode_phases.py 256 2>/dev/null | sed -n '1p;9,11p;22,23p;/encoder layer/,$p' || exit 1
  for w in 16; do
    echo "$v workers $w: $(SLIMT_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 96 --warmup 16 --workers $w --profile-kernel none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])[\"value\"])")" || exit 1
  done
done
