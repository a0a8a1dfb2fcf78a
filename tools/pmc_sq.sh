#!/bin/bash
# SQ-level counters for one op_bench invocation:  tools/pmc_sq.sh <tag> <op_bench args...>
# (separate rocprofv3 --pmc passes, no trace domains; run from the repo root on the GPU box)
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $set -d $out/$n -o p -- python3 tools/op_bench.py "$@" 3 > $out/$n.log 2>&1 || { echo "pass failed: $set"; tail -5 $out/$n.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "igemm" in k or "wgrad" in k or "convt" in k.lower():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} {sum(v)/len(v):16.0f}  (x{len(v)})")
PY
