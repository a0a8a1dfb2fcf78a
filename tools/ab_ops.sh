#!/bin/bash
# Operator-level A/B on ONE box (GPU box, repo root):  tools/ab_ops.sh <tag> "<ENV=.. arm A>" "<arm B>" -- "<op_bench args>" ...
# Every op_bench line is run under each arm, arms interleaved; "default" = no extra environment.
tag=$1; shift
arms=()
while [ "$1" != "--" ]; do arms+=("$1"); shift; done
shift
out=gpurun_out/${tag}_ops.txt
: > $out
for spec in "$@"; do
  for arm in "${arms[@]}"; do
    if [ "$arm" = "default" ]; then r=$(python3 tools/op_bench.py $spec 30 2>/dev/null | tail -1)
    else r=$(env $arm python3 tools/op_bench.py $spec 30 2>/dev/null | tail -1); fi
    printf "%-28s %s\n" "$arm" "$r" >> $out
  done
done
cat $out
