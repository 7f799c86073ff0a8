#!/bin/bash
# usage: tools/kernel_times.sh "<mode> <workers>" ...  -> throughput + avg decode/encode launch time under load
for cfg in "$@"; do set -- $cfg; for k in decode_fused encode_fused; do echo "mode $1 workers $2 $k: $(GPU_MAX_HW_QUEUES=64 timeout -k 10 120 python bench.py --steps 128 --warmup 32 --workers $2 --decode-mode $1 --profile-kernel $k --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'Mtok/s; avg launch us', round(d['roofline']['avg_launch_us'],1))")" || exit 1; done; done
