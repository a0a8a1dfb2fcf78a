#!/bin/bash
# The round-end evidence in TWO GPU calls of <= 20 minutes each (GPU box, repo root):
#   tools/round_end.sh <round tag, e.g. r04> 1   final_measure.sh (bench lines of every configuration, kernel trace, per-layer table,
#                                                 HBM traffic passes), the traffic JSON into profiles/ ON THE BOX and the default bench
#                                                 line again so that roofline.traffic is filled in, the one-rank RCCL trace, the kernel
#                                                 trace of the quantify flow
#   tools/round_end.sh <round tag> 2              SQ counter passes of the main MFMA kernels (tools/sq_set.sh)
r=${1:-r05}
part=${2:-1}
export TMPDIR=/tmp
set -e
if [ "$part" = "1" ]; then
  bash tools/final_measure.sh ${r}_final > gpurun_out/${r}_final_measure.log 2>&1
  cp gpurun_out/${r}_final_pmc_traffic.json profiles/${r}_pmc_traffic.json
  python3 bench.py > gpurun_out/${r}_final_bench.json 2> gpurun_out/${r}_final_bench.err
  python3 tools/show_bench.py gpurun_out/${r}_final_bench.json
  bash tools/rccl1_trace.sh ${r}_rccl1 > gpurun_out/${r}_rccl1.log 2>&1
  out=$PWD/gpurun_out/prof_${r}_quantify
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --mode quantify --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${r}_quantify_prof_bench.json 2> $out/err.log
  cp "$(find $out -name '*kernel_stats.csv' | head -1)" gpurun_out/${r}_quantify_kernel_stats_after.csv
else
  bash tools/sq_set.sh $r > gpurun_out/${r}_sq_set.log 2>&1
fi
echo done
