#!/bin/bash
# HBM-side traffic of every kernel of one training step:  tools/pmc_traffic.sh <tag>      (GPU box, repo root)
# Two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace domains, the program
# itself after `--`.  Result: gpurun_out/<tag>_pmc_traffic.json (copy to profiles/ to have bench.py report it).
tag=${1:-r04}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_traffic_$tag
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $c -d $out/$c -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $out/$c.log 2>&1 \
    || { echo "pass failed: $c"; tail -5 $out/$c.log; exit 1; }
done
python3 tools/summarize_pmc.py $(find $out/FETCH_SIZE -name '*counter_collection.csv' | head -1) \
                               $(find $out/WRITE_SIZE -name '*counter_collection.csv' | head -1) gpurun_out/${tag}_pmc_traffic.json
