#!/bin/bash
# Per-kernel time of the headline bench under rocprofv3:  tools/rocprof_stats.sh <tag> [bench args...]   (GPU box, repo root)
# Writes gpurun_out/<tag>_kernel_stats.csv (copy to profiles/ to keep it) and gpurun_out/<tag>_bench.json.
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" > gpurun_out/${tag}_bench.json 2> $out/err.log || { echo "rocprofv3 failed"; tail -5 $out/err.log; exit 1; }
f=$(find $out -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
NS = 14   # steps in the trace: 1 initialisation + 2 warm-up + 5 timed + 3 + 3 instrumented (implicit-GEMM leg, HBM leg) steps of bench.py
print(f"total kernel time {tot/1e6/NS:.3f} ms/step over {NS} steps, {sum(int(r['Calls']) for r in rows)/NS:.0f} launches/step")
for r in rows[:45]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls'])/NS:6.1f}/step {float(r['TotalDurationNs'])/NS/1e3:9.1f} us/step  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
