#!/bin/bash
# A/B of one environment switch on the same box, interleaved rounds:  tools/ab_env.sh VAR "<op_bench args>" [rounds]
# A: VAR=0, B: VAR unset (default)
var=$1; args=$2; rounds=${3:-2}
for r in $(seq $rounds); do
  env $var=0 python3 tools/op_bench.py $args 30 2>/dev/null | sed "s/^/A($var=0)  /"
  python3 tools/op_bench.py $args 30 2>/dev/null | sed 's/^/B(default)  /'
done
