#!/bin/bash
# GPU box: everything DESIGN.md / profiles/ quote for a round, in one call.
# usage: tools/round_evidence.sh <tag>     (writes gpurun_out/<tag>_*; copy what is quoted into profiles/)
# Order matters: the PMC passes come first and their digests are copied into profiles/ (of this
# box's copy of the repo) and named in the manifest profiles/CURRENT BEFORE the final bench.py run, so that the
# bench line's `traffic` and `mfma_busy_frac` cite files of the same tag (copy gpurun_out/CURRENT to profiles/ too).
# Stage (2nd argument): a = counters, bench line of record, phases, occupancy; b = the other shapes, the 2-rank
# rehearsal, service and text benches; all (default) = both (more than one gpurun call's 20 minutes).
TAG=${1:-r03}
STAGE=${2:-all}
mkdir -p gpurun_out
if [ "$STAGE" != "b" ]; then
NO_CPU=1 WITH_BASE=1 tools/profile_round.sh $TAG || exit 1
tools/pmc_sq.sh $TAG > gpurun_out/${TAG}_sq_run.log 2>&1 || { echo "pmc_sq failed"; tail -5 gpurun_out/${TAG}_sq_run.log; exit 1; }
PRESET=base BATCH=2048 tools/pmc_sq.sh ${TAG}_base > gpurun_out/${TAG}_base_sq_run.log 2>&1 || { echo "pmc_sq base failed"; exit 1; }
tail -3 gpurun_out/${TAG}_sq_run.log; tail -3 gpurun_out/${TAG}_base_sq_run.log
cp gpurun_out/${TAG}_pmc_FETCH_SIZE.json gpurun_out/${TAG}_pmc_WRITE_SIZE.json gpurun_out/${TAG}_base_pmc_FETCH_SIZE.json \
   gpurun_out/${TAG}_base_pmc_WRITE_SIZE.json gpurun_out/${TAG}_sq_pmc.json gpurun_out/${TAG}_base_sq_pmc.json profiles/ || exit 1
# the manifest bench.py reads: ONE evidence set, named (no "newest by file name")
python - $TAG <<'PY' || exit 1
import json, sys
t = sys.argv[1]
m = {"tag": t,
     "tiny11": {"FETCH_SIZE": f"{t}_pmc_FETCH_SIZE.json", "WRITE_SIZE": f"{t}_pmc_WRITE_SIZE.json", "sq_pmc": f"{t}_sq_pmc.json"},
     "base": {"FETCH_SIZE": f"{t}_base_pmc_FETCH_SIZE.json", "WRITE_SIZE": f"{t}_base_pmc_WRITE_SIZE.json", "sq_pmc": f"{t}_base_sq_pmc.json"}}
for name in ("profiles/CURRENT", "gpurun_out/CURRENT"):
    json.dump(m, open(name, "w"), indent=1)
PY
# the bench line of record for this tag (with cpu_baseline), citing the counters above
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python - gpurun_out/${TAG}_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(f"value {d['value']/1e6:.2f} M, sustained {d['sustained']['value']/1e6:.2f} M; {r['kernel']} {r['avg_launch_us']:.0f} us, frac {r['frac']:.5f}, "
      f"mfma_busy {r.get('mfma_busy_frac')}, traffic {r['traffic']}, sources {r['traffic_source']} {r['counters_source']}")
for k in ("model_forward", "model_forward_no_alignments", "model_forward_per_batch_shortlist", "single_stream"):
    print(f"  {k}: {d[k]['value']/1e6:.2f} M tok/s")
print(f"  cpu_baseline: {d['cpu_baseline']['value']:.0f} tok/s on {d['cpu_baseline']['cores']} cores")
PY
timeout -k 10 300 python bench.py --preset base --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_base_bench.json 2> gpurun_out/${TAG}_base_bench.err || exit 1
SLIMT_DECODE_MODE=2 timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_phases.txt 2>&1 || exit 1  # 16 sentences per workgroup (the loaded tiling)
for mode in 4 5; do SLIMT_DECODE_MODE=$mode timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_phases_mode$mode.txt 2>&1 || exit 1; done
timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_phases_loaded.txt 2>&1 || exit 1
timeout -k 10 120 python tools/decode_phases_loaded.py 20 64 128 > gpurun_out/${TAG}_phases_loaded_S128.txt 2>&1 || exit 1
timeout -k 10 120 python tools/encode_wide_phases.py 256 tiny11 > gpurun_out/${TAG}_encoder_phases.txt 2>&1 || exit 1
timeout -k 10 120 python tools/encode_wide_phases.py 256 base > gpurun_out/${TAG}_base_encoder_phases.txt 2>&1 || exit 1
timeout -k 10 200 python tools/occupancy_trace.py 20 480 256 > gpurun_out/${TAG}_occupancy.txt 2>&1 || exit 1
grep -E "total" gpurun_out/${TAG}_phases_loaded.txt gpurun_out/${TAG}_encoder_phases.txt
fi
[ "$STAGE" = "a" ] && exit 0
{
  tools/sweep.sh "--forward-steps 0" "--forward-steps 0 --batch 64" "--batch 64 --merge 4" "--forward-steps 0 --batch 64 --adaptive-rows 0" "--forward-steps 0 --batch 512 --shortlist 0" \
    "--forward-steps 0 --batch 340 --src-len 24" "--forward-steps 0 --batch 128 --src-len 48" "--forward-steps 0 --ragged --eos-bias 8" \
    "--forward-steps 0 --batch 512 --workers 12" "--forward-steps 0 --batch 128 --src-len 64" "--forward-steps 0 --batch 64 --src-len 96" \
    "--forward-steps 0 --batch 64 --src-len 128" "--forward-steps 0 --ragged" "--forward-steps 0 --preset base" \
    "--total-sentences 4096 --batch 512 --steps 10" "--total-sentences 4096 --batch 256 --steps 10" \
    "--forward-steps 0 --batch 4096 --workers 1 --sustained-steps 0" \
    "--forward-steps 0 --kv-format 2" "--forward-steps 0 --preset base --kv-format 2" "--forward-steps 0 --batch 512 --shortlist 0 --decode-mode 6" \
    "--forward-steps 0 --kv-tight-limit 0" "--forward-steps 0 --preset base --kv-tight-limit 0"
} > gpurun_out/${TAG}_configs.txt 2>&1
cat gpurun_out/${TAG}_configs.txt
SLIMT_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --workers 8 --no-cpu-baseline --sustained-steps 0 --forward-steps 0 > gpurun_out/${TAG}_rehearsal_2ranks_on_1gpu.json 2> gpurun_out/${TAG}_rehearsal.err
cut -c1-200 gpurun_out/${TAG}_rehearsal_2ranks_on_1gpu.json
rm -f gpurun_out/${TAG}_service_bench.jsonl
# a timed pass of ~1 s (30 rounds over the requests; 3 rounds = 0.1 s, of which fill and drain are 6 %), 16 requests outstanding per
# client (profiles/r06_service_steady_state.txt)
export SLIMT_SERVICE_ROUNDS=30 SLIMT_SERVICE_WINDOW=16
for cfg in "10 32768 4096 0" "10 32768 4096 1" "10 32768 4096 flat" "10 32768 lex 0" "10 32768 lex 1" "6 32768 4096 1" "10 32768 0 1"; do
  timeout -k 10 200 python tools/async_bench.py $cfg >> gpurun_out/${TAG}_service_bench.jsonl 2>> gpurun_out/${TAG}_service.err || { echo "service bench $cfg failed"; exit 1; }
done
SLIMT_SERVICE_REPLICAS=2 timeout -k 10 200 python tools/async_bench.py 5 32768 4096 1 >> gpurun_out/${TAG}_service_bench.jsonl 2>> gpurun_out/${TAG}_service.err
# the reference's default word budget (max_words 1024, Frontend.hh:21-39): merged launches (default, 8 batches) against one batch per launch
for cfg in "8 4096 0" "1 4096 0" "8 lex 0" "1 lex 0" "8 4096 1"; do
  set -- $cfg
  SLIMT_SERVICE_MAX_WORDS=1024 SLIMT_SERVICE_MERGE=$1 timeout -k 10 200 python tools/async_bench.py 10 32768 $2 $3 >> gpurun_out/${TAG}_service_bench.jsonl 2>> gpurun_out/${TAG}_service.err || { echo "service bench max_words 1024 $cfg failed"; exit 1; }
done
# SURVEY 8(d)'s secondary workload: lengths ~ U{8..64}, cut into batches by the reference's batcher rule (the Service's queue)
for cfg in "10 32768 4096 0" "10 32768 4096 1"; do
  SLIMT_SERVICE_MAX_LEN=64 timeout -k 10 250 python tools/async_bench.py $cfg >> gpurun_out/${TAG}_service_bench.jsonl 2>> gpurun_out/${TAG}_service.err || { echo "secondary workload $cfg failed"; exit 1; }
done
python - gpurun_out/${TAG}_service_bench.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(f"{d['target_tokens_per_s']/1e6:6.2f} M tok/s  {d['workload'][:120]}")
PY
timeout -k 10 250 python tools/text_bench.py 3000 8 > gpurun_out/${TAG}_text_bench.json 2> gpurun_out/${TAG}_text_bench.err
cut -c1-600 gpurun_out/${TAG}_text_bench.json
timeout -k 10 100 python tools/sync_workers_bench.py > gpurun_out/${TAG}_sync_workers.jsonl 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_sync_workers.jsonl
