#!/bin/bash
# GPU box: XCD-affine decoder placement, off / 1 / 2 home XCDs: throughput, loaded phase stamps, a parity subset.
mkdir -p gpurun_out
TAG=${1:-xcd}
for x in 0 1 2; do
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --xcd-affinity $x > gpurun_out/${TAG}_bench_x$x.json 2> gpurun_out/${TAG}_bench_x$x.err || { echo "bench x=$x failed"; tail -5 gpurun_out/${TAG}_bench_x$x.err; exit 1; }
    python - gpurun_out/${TAG}_bench_x$x.json $x <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"xcd_affinity {sys.argv[2]}: {d['value']/1e6:.2f} M tok/s, sustained {d['sustained']['value']/1e6:.2f} M, decoder launch {d['roofline']['avg_launch_us']:.0f} us, in flight {d['roofline']['launches_in_flight']:.1f}")
PY
  done
  SLIMT_XCD_AFFINITY=$x timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_phases_loaded_x$x.txt 2>&1 || { echo "phases x=$x failed"; tail -5 gpurun_out/${TAG}_phases_loaded_x$x.txt; exit 1; }
  grep -E "total|attention|logits" gpurun_out/${TAG}_phases_loaded_x$x.txt
done
