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
# JSON digest: counters per kernel + busy fractions (profiles/<tag>_sq_pmc.json; bench.py's roofline.mfma_busy_frac)
python3 - gpurun_out/${TAG}_sq_pmc.txt gpurun_out/${TAG}_sq_pmc.json <<'PY'
import json, re, sys, collections
k = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+(\S+)\s+launches\s+(\d+)\s+avg\s+(\S+)", line)
    if m:
        k[m.group(1)][m.group(2)] = float(m.group(4))
out = {}
for name, c in k.items():
    d = dict(c)
    if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
        simd_cycles = cyc * 256 * 4                 # 256 CUs x 4 SIMDs
        d["kernel_cycles"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles       # counts cycles
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles      # counts quad-cycles
        if "SQ_WAVE_CYCLES" in c:
            for key, label in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_issue_stall_frac"),
                               ("SQ_ACTIVE_INST_ANY", "wave_issuing_frac")):
                if key in c:
                    d[label] = c[key] / c["SQ_WAVE_CYCLES"]
    if c.get("TCC_REQ_sum"):
        d["l2_hit_frac"] = c.get("TCC_HIT_sum", 0.0) / (c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    out[name] = d
json.dump({"workload": "bench.py --batch %s --workers 1: every CU holds a workgroup of the kernel (full occupancy); one launch profiled alone" % sys.argv[1].split("/")[-1], "kernels": out}, open(sys.argv[2], "w"), indent=1)
for name, d in out.items():
    print(name, {x: round(d[x], 4) for x in ("mfma_busy_frac", "valu_busy_frac", "wave_parked_frac", "wave_issue_stall_frac", "wave_issuing_frac", "l2_hit_frac", "lds_bank_conflict_frac") if x in d})
PY
