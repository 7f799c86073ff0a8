#!/bin/bash
# One-GPU measurements behind DESIGN section 7's predictions for the driver's 8-GPU runs:
# what ONE rank does at N = 1 / 2 / 4 / 8 of BASELINE config 5 (the same 4,096 sentences at every N: 4096 / N per rank),
# by how its share is cut into batches and contexts; and the weak-scaling headline per rank. Usage: tools/scale_predict.sh OUT
OUT=${1:-gpurun_out/r06_scale_prediction.txt}
: > $OUT
row() {  # label, then bench.py arguments
  label=$1; shift
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --forward-steps 0 --sustained-steps 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-72s %8.2f M tok/s   %7.3f ms per pass over the rank\'s share' % ('$label', d['value'] / 1e6, d['ms_per_step']))" >> $OUT
}
echo "# config 5, ONE rank's share on one GPU (tokens/s of that rank; x N = the job's, if the ranks do not disturb each other)" >> $OUT
row "N=8: 512 sentences, 1 batch of 512, 1 context"            --total-sentences 512 --batch 512 --workers 1
row "N=8: 512 sentences, 4 batches of 128, 4 contexts"         --total-sentences 512 --batch 128 --workers 4
row "N=8: 512 sentences, 8 batches of 64, 8 contexts"          --total-sentences 512 --batch 64 --workers 8
row "N=4: 1024 sentences, 4 batches of 256, 4 contexts"        --total-sentences 1024 --batch 256 --workers 4
row "N=4: 1024 sentences, 8 batches of 128, 8 contexts"        --total-sentences 1024 --batch 128 --workers 8
row "N=2: 2048 sentences, 8 batches of 256, 8 contexts"        --total-sentences 2048 --batch 256 --workers 8
row "N=2: 2048 sentences, 16 batches of 128, 16 contexts"      --total-sentences 2048 --batch 128 --workers 16
row "N=1: 4096 sentences, 16 batches of 256, 16 contexts"      --total-sentences 4096 --batch 256 --workers 16
echo "# weak scaling (the headline, per rank): 20 workers x batch 256" >> $OUT
row "headline, one rank"                                        --batch 256
cat $OUT
