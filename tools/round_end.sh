#!/bin/bash
# Everything the round-end evidence needs, in one GPU call (GPU box, repo root):  tools/round_end.sh <round tag, e.g. r03>
# 1. tools/final_measure.sh <tag>_final (bench lines of every configuration, kernel trace, per-layer table, HBM traffic passes)
# 2. the traffic JSON into profiles/ ON THE BOX, then the default bench line again so that roofline.traffic is filled in
# 3. SQ counter passes of the main MFMA kernels, 4. kernel trace of the quantify flow
r=${1:-r03}
export TMPDIR=/tmp
set -e
bash tools/final_measure.sh ${r}_final > gpurun_out/${r}_final_measure.log 2>&1
cp gpurun_out/${r}_final_pmc_traffic.json profiles/${r}_pmc_traffic.json
python3 bench.py > gpurun_out/${r}_final_bench.json 2> gpurun_out/${r}_final_bench.err
python3 tools/show_bench.py gpurun_out/${r}_final_bench.json
bash tools/sq_set.sh $r > gpurun_out/${r}_sq_set.log 2>&1
out=$PWD/gpurun_out/prof_${r}_quantify
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --mode quantify --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${r}_quantify_prof_bench.json 2> $out/err.log
cp "$(find $out -name '*kernel_stats.csv' | head -1)" gpurun_out/${r}_quantify_kernel_stats_after.csv
echo done
