#!/bin/bash
# SQ-side PMC of the two persistent kernels at full occupancy (B=4096 -> 256 workgroups each):
# where do the wave cycles go (issue / wait / which pipe)? Separate passes per counter set.
# usage: tools/pmc_sq.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
TAG=${1:-sq}
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > gpurun_out/${TAG}_sq_counter_names.txt
wc -l gpurun_out/${TAG}_sq_counter_names.txt
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  rm -rf gpurun_out/pmc_${TAG}_$i
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python3 bench.py ${PRESET:+--preset $PRESET} --batch ${BATCH:-4096} --workers 1 --steps 2 --warmup 1 --sustained-steps 0 --no-cpu-baseline --profile-kernel none > gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pmc set $i ($c) failed"; tail -3 gpurun_out/pmc_${TAG}_$i.log; continue; }
  f=$(ls gpurun_out/pmc_${TAG}_$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY' | tee -a gpurun_out/${TAG}_sq_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = next((x for x in ("decode_fused", "encode_fused", "encode_tall", "encode_wide") if x in n), None)
    if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:14s} {c:28s} launches {len(v):3d} avg {sum(v)/len(v):.5g}")
PY
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM_WR
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_CVT
SETS
