#!/bin/bash
# GPU box: timing-only bound (wrong results) of a 16-bit K/V cache form against the 24- and 20-bit ones, one box
mkdir -p gpurun_out; OUT=gpurun_out/${1:-kv16}.txt; : > $OUT
run() { local name=$1 lib=$2; shift 2
  local v=$(env "$@" SLIMT_HIP_LIB=$PWD/slimt_amd/lib/$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --profile-kernel none --no-cpu-baseline --forward-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))")
  echo "$name: $v" | tee -a $OUT; }
for rep in 1 2; do
  run "24-bit (inlined)" exp_NOKV20.so X=1 || exit 1
  run "20-bit" libslimt_hip.so X=1
  run "16-bit emulation (4 of 6 quads, SDWA unpack + packed add), all launches kept" exp_KV16.so SLIMT_KV_BY_LAUNCH=8
done
for lib in exp_NOKV20.so libslimt_hip.so exp_KV16.so; do
  e=X=1; [ $lib = exp_KV16.so ] && e=SLIMT_KV_BY_LAUNCH=8
  env $e SLIMT_HIP_LIB=$PWD/slimt_amd/lib/$lib SLIMT_DECODE_MODE=2 timeout -k 10 120 python tools/decode_phases.py 256 2>&1 | grep -A22 "step 20" | grep "total\|attention" | tr '\n' ' ' | sed "s/^/$lib alone: /" | tee -a $OUT; echo | tee -a $OUT
  env $e SLIMT_HIP_LIB=$PWD/slimt_amd/lib/$lib timeout -k 10 120 python tools/decode_phases_loaded.py 2>&1 | grep "total\|attention" | tr '\n' ' ' | sed "s/^/$lib loaded: /" | tee -a $OUT; echo | tee -a $OUT
done
