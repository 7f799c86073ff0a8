#!/bin/bash
# GPU box: Model::forward rates from host buffers: bench.py's model_forward* lines and the C++ Service
# with a frozen shortlist / a lexical shortlist per batch, with and without alignment rows.
TAG=${1:-hp}
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"value {d['value']/1e6:.2f} M, sustained {d['sustained']['value']/1e6:.2f} M")
for k in ("model_forward", "model_forward_no_alignments", "model_forward_per_batch_shortlist", "single_stream"):
    if k in d: print(f"{k}: {d[k]['value']/1e6:.2f} M tok/s")
PY
rm -f gpurun_out/${TAG}_service_bench.jsonl
for cfg in "10 32768 4096 0" "10 32768 4096 1" "10 32768 lex 0" "10 32768 lex 1" "6 32768 4096 1" "10 32768 0 1"; do
  timeout -k 10 200 python tools/async_bench.py $cfg >> gpurun_out/${TAG}_service_bench.jsonl 2>> gpurun_out/${TAG}_service.err || { echo "service bench $cfg failed"; tail -3 gpurun_out/${TAG}_service.err; exit 1; }
done
python - gpurun_out/${TAG}_service_bench.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(f"{d['target_tokens_per_s']/1e6:6.2f} M tok/s  {d['workload'][:150]}")
PY
