#!/bin/bash
# SQ-level counters for one op_bench invocation:  tools/pmc_sq.sh <tag> <op_bench args...>
# (separate rocprofv3 --pmc passes, no trace domains; run from the repo root on the GPU box).
# Writes gpurun_out/pmc_<tag>/summary.json: per kernel, the average counter value per dispatch plus the derived
# MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs ... see tools/summarize_sq.py).
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $set -d $out/$n -o p -- python3 tools/op_bench.py "$@" 3 > $out/$n.log 2>&1 || { echo "pass failed: $set"; tail -5 $out/$n.log; }
done
python3 tools/summarize_sq.py $out "$tag" "$*"
# keep the summary only: the raw counter CSVs of five passes are tens of MB per kernel (gpurun merges at most 64 MiB back)
find $out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
