#!/bin/bash
# A/B of whole-step switches on ONE box, interleaved:  tools/ab_step.sh <tag> "ENV1=.. ENV2=.." "ENV=.." ...   (GPU box, repo root)
# Each argument is one arm (a space-separated list of VAR=value, or "default"); every arm runs `rounds` times, interleaved.
tag=$1; shift
rounds=${ROUNDS:-2}
out=gpurun_out/${tag}_ab.txt
: > $out
for r in $(seq $rounds); do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    f=gpurun_out/${tag}_arm${i}_r${r}.json
    if [ "$arm" = "default" ]; then python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $f 2>/dev/null
    else env $arm python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $f 2>/dev/null; fi
    python3 - "$f" "$arm" >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d.get("hbm_leg") or {}
print(f"{sys.argv[2]:45s} {d['ms_per_step']:8.3f} ms/step {d['value']:8.1f} img/s | hbm leg {h.get('ms_per_step', 0):.3f} ms")
PY
  done
done
cat $out
