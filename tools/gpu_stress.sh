#!/bin/bash
# GPU box: the stress runs on the final kernels: concurrent translates against the CPU checker (tiny11 S = 20 / 100, base), run-to-run
# determinism, device memory after create / destroy cycles, generated shortlists under uneven load, merged launches under uneven load. usage: tools/gpu_stress.sh <tag>
mkdir -p gpurun_out
TAG=${1:-stress}; OUT=gpurun_out/${TAG}_stress.txt; : > $OUT
run() { echo "## $*" >> $OUT; timeout -k 10 300 "$@" >> $OUT 2>&1 || { echo "FAILED: $*"; tail -5 $OUT; exit 1; }; }
run python tools/stress_vs_oracle.py tiny11 24 4 32 20
run python tools/stress_vs_oracle.py tiny11 12 4 9 100
run python tools/stress_vs_oracle.py tiny11 12 4 17 50
run python tools/stress_vs_oracle.py base 12 4 32 20
run env SLIMT_STRESS_CENTRES=7 python tools/stress_vs_oracle.py tiny11 8 4 64 32
run env SLIMT_STRESS_CENTRES=8 python tools/stress_vs_oracle.py tiny11 8 4 40 60
run python tools/stress_vs_oracle.py base 8 4 64 32  # (calibrates its centres on the first batch)
run python tools/stress_determinism.py
run python tools/leak_check.py
run python tools/stress_generated.py 6 4
run python tools/stress_generated.py 4 3 base
run python tools/stress_many.py 6 9
run python tools/stress_many.py 4 6 base
grep -v amdgpu.ids $OUT | grep -i "mismatch\|delta\|##"
