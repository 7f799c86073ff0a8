for eb in 8 10 12 14; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --forward-steps 0 --sustained-steps 0 --ragged --eos-bias $eb 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('eos_bias $eb', round(d['value']/1e6,2), d['decoder_tile_live_fraction'], d['config']['tokens_per_step_all_gpus'] / d['config']['sentences_per_step_all_gpus'])"; done
