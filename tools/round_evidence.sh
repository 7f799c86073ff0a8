#!/bin/bash
# GPU box: everything DESIGN.md / profiles/ quote for a round, in one call.
# usage: tools/round_evidence.sh <tag>     (writes gpurun_out/<tag>_*)
TAG=${1:-r02}
mkdir -p gpurun_out
WITH_BASE=1 tools/profile_round.sh $TAG || exit 1
timeout -k 10 120 python tools/decode_phases.py 256 > gpurun_out/${TAG}_phases.txt 2>&1 || exit 1
timeout -k 10 120 python tools/decode_phases_loaded.py > gpurun_out/${TAG}_phases_loaded.txt 2>&1 || exit 1
timeout -k 10 200 python tools/occupancy_trace.py 20 480 256 > gpurun_out/${TAG}_occupancy.txt 2>&1 || exit 1
{
  tools/sweep.sh "--steps 20 --warmup 5" "--batch 64" "--batch 512 --shortlist 0" "--batch 512 --workers 12" \
    "--batch 128 --src-len 64" "--batch 64 --src-len 128" "--ragged" "--preset base" \
    "--total-sentences 4096 --batch 512 --steps 10" "--total-sentences 4096 --batch 256 --steps 10" \
    "--batch 4096 --workers 1 --sustained-steps 0"
} > gpurun_out/${TAG}_configs.txt 2>&1
cat gpurun_out/${TAG}_configs.txt
SLIMT_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --workers 8 --no-cpu-baseline --sustained-steps 0 > gpurun_out/${TAG}_rehearsal_2ranks_on_1gpu.json 2> gpurun_out/${TAG}_rehearsal.err
cut -c1-300 gpurun_out/${TAG}_rehearsal_2ranks_on_1gpu.json
for cfg in "10 32768 4096 0" "10 32768 4096 1" "6 32768 4096 0" "10 32768 0 1"; do timeout -k 10 200 python tools/async_bench.py $cfg >> gpurun_out/${TAG}_service_bench.jsonl 2>> gpurun_out/${TAG}_service.err; done
cut -c1-330 gpurun_out/${TAG}_service_bench.jsonl
timeout -k 10 250 python tools/text_bench.py 3000 8 > gpurun_out/${TAG}_text_bench.json 2> gpurun_out/${TAG}_text_bench.err
cut -c1-400 gpurun_out/${TAG}_text_bench.json
