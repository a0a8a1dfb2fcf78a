#!/bin/bash
# One-rank RCCL rehearsal under the kernel trace (GPU box, repo root):  tools/rccl1_trace.sh <tag>
# UNETDC_DP_FORCE=1 makes bench.py build a ONE-rank process group and really issue every bucket's ncclAllReduce behind the HIP
# backward (dp.py, single_rank_collectives).  The trace shows where RCCL's kernels land between the persistent compute kernels.
tag=${1:-r04_rccl1}
export TMPDIR=/tmp
export UNETDC_DP_FORCE=1
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_bench.json 2> $out/err.log || { echo "rocprofv3 failed"; tail -5 $out/err.log; exit 1; }
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 tools/rccl_trace_summary.py "$f" > gpurun_out/${tag}_trace.txt
# the same command WITHOUT the process group, same tool: what the step looks like under the profiler without collectives
unset UNETDC_DP_FORCE
out2=$PWD/gpurun_out/prof_${tag}_nodp
rm -rf $out2; mkdir -p $out2
rocprofv3 --kernel-trace --output-format csv -d $out2 -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_nodp_bench.json 2> $out2/err.log || { echo "rocprofv3 failed"; tail -5 $out2/err.log; exit 1; }
f2=$(find $out2 -name '*kernel_trace.csv' | head -1)
echo "==== same command without UNETDC_DP_FORCE ====" >> gpurun_out/${tag}_trace.txt
python3 tools/rccl_trace_summary.py "$f2" | head -4 >> gpurun_out/${tag}_trace.txt
head -70 gpurun_out/${tag}_trace.txt
