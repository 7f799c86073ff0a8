#!/bin/bash
# usage: tools/sweep.sh "<bench args>" ...   one bench line (value, sustained, decoder launch us, in flight) per argument set
for args in "$@"; do
  echo -n "[$args] "
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
extra=''.join('  %s %.2f M' % (k, d[k]['value']/1e6) for k in ('model_forward','model_forward_per_batch_shortlist','single_stream') if k in d)
print('value %.2f M  sustained %.2f M  launch %.0f us  in flight %.1f%s' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6, r['avg_launch_us'], r['launches_in_flight'], extra))" || echo failed
done
