#!/bin/bash
# GPU box: timing-only bounds (wrong results) for the decoder's K/V and weight streams, on ONE box:
#   base | KV56 (5 of 6 K/V quads: the bytes of a 20-bit cache, no unpack cost) | KVSMALL (K/V footprint ~0) | WSMALL (weights ~0)
# each with 16 and 32 sentences per workgroup (decode mode 0 / 3), all caches kept where the variant shrinks them.
# usage: tools/exp_bounds.sh <tag>
mkdir -p gpurun_out
TAG=${1:-bounds}
OUT=gpurun_out/${TAG}.txt; : > $OUT
run() {  # name lib mode env...
  local name=$1 lib=$2 mode=$3; shift 3
  local v=$(env "$@" SLIMT_HIP_LIB=$PWD/slimt_amd/lib/$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --decode-mode $mode --profile-kernel none --no-cpu-baseline --forward-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))")
  echo "$name mode $mode $*: $v" | tee -a $OUT
}
for rep in 1 2; do
  run base libslimt_hip.so 0 X=1 || exit 1
  run base libslimt_hip.so 3 X=1
  run KV56 exp_KV56.so 0 SLIMT_KV_LAUNCH_BUDGET_MB=330
  run KV56 exp_KV56.so 0 X=1
  run KVSMALL exp_KVSMALL.so 0 SLIMT_KV_BY_LAUNCH=8
  run KVSMALL exp_KVSMALL.so 3 SLIMT_KV_BY_LAUNCH=8
  run WSMALL exp_WSMALL.so 0 X=1
done
for v in libslimt_hip.so exp_KV56.so exp_KVSMALL.so; do
  e=X=1; [ $v = exp_KV56.so ] && e=SLIMT_KV_LAUNCH_BUDGET_MB=330; [ $v = exp_KVSMALL.so ] && e=SLIMT_KV_BY_LAUNCH=8
  env $e SLIMT_HIP_LIB=$PWD/slimt_amd/lib/$v timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_${v}_loaded.txt 2>&1 || exit 1
  echo "$v loaded: $(grep total gpurun_out/${TAG}_${v}_loaded.txt | sed 's/.*total//') attn $(grep attention gpurun_out/${TAG}_${v}_loaded.txt | awk '{printf "%s ", $2}') logits $(grep logits gpurun_out/${TAG}_${v}_loaded.txt | awk '{print $2}')" | tee -a $OUT
done
