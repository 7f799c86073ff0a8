for mode in 3 2; do
  SLIMT_DECODE_MODE=$mode timeout -k 10 200 python tools/decode_phases.py 512 32 0 2>&1 | grep -A26 "step 20" > gpurun_out/r05_c4_phases_mode$mode.txt
  echo "== mode $mode alone"; cat gpurun_out/r05_c4_phases_mode$mode.txt
  SLIMT_DECODE_MODE=$mode timeout -k 10 200 python tools/decode_phases_loaded.py 20 512 32 0 > gpurun_out/r05_c4_loaded_mode$mode.txt 2>&1
  echo "== mode $mode loaded"; cat gpurun_out/r05_c4_loaded_mode$mode.txt
done
for args in "--batch 512 --shortlist 0" "--batch 512 --shortlist 0 --decode-mode 2" "--batch 512 --shortlist 0 --workers 10"; do
  echo -n "[$args] "; timeout -k 10 300 python bench.py --steps 10 --warmup 2 --profile-kernel none --no-cpu-baseline --forward-steps 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  sustained %.2f M' % (d['value']/1e6, d.get('sustained',{}).get('value',0)/1e6))"
done
