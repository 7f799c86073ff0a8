#!/bin/bash
# engine-level rows on a REALISTIC decode -- sentences that emit EOS at different steps (eos_bias 6),
# fixed and ragged (U{8..32}) source lengths; tokens counted as Model.cc:127-137 records them; mean live-slot fraction of
# a 16-sentence decoder tile. Usage: tools/eos_rows.sh OUT
OUT=${1:-gpurun_out/r06_staggered_eos.txt}
: > $OUT
row() {
  label=$1; shift
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --forward-steps 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
lf = d.get('decoder_tile_live_fraction')
tps = d['config']['tokens_per_step_all_gpus'] / d['config']['sentences_per_step_all_gpus']
print('%-58s %8.2f M tok/s  sustained %8.2f   %5.1f tokens per sentence   live slots %s' % ('$label', d['value'] / 1e6, d['sustained']['value'] / 1e6, tps, ('%.3f' % lf) if lf is not None else '1.000 (nobody ends)'))" >> $OUT
}
row "headline: S = 32, nobody emits EOS (T = 48)"
row "S = 32, staggered EOS (eos_bias 6)"                   --eos-bias 6
row "lengths U{8..32}, nobody emits EOS"                   --ragged
row "lengths U{8..32}, staggered EOS (eos_bias 6)"         --ragged --eos-bias 6
row "lengths U{8..32}, staggered EOS (eos_bias 4)"         --ragged --eos-bias 4
cat $OUT
