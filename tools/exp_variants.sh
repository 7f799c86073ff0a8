#!/bin/bash
# usage: tools/exp_variants.sh "<mode> <workers>" <lib-suffix>...   (kernel experiments; libs built with SLIMT_HIP_LIB)
set -- "$@"; CFG=$1; shift
for v in "$@"; do
  lib=/root/repo/slimt_amd/lib/exp_$v.so
  [ "$v" = base ] && lib=/root/repo/slimt_amd/lib/libslimt_hip.so
  set -- $CFG
  echo "$v mode $1 workers $2: $(GPU_MAX_HW_QUEUES=64 SLIMT_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 128 --warmup 32 --workers $2 --decode-mode $1 --profile-kernel none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])[\"value\"])")" || exit 1
done
