#!/bin/bash
# A/B of switches on the HBM-bound entry points (GPU box, repo root):  tools/ab_hbm.sh <tag> "ENV=.." ...   -> per arm: step time + every hbm_leg entry
tag=$1; shift
rounds=${ROUNDS:-2}
out=gpurun_out/${tag}_ab.txt
: > $out
for r in $(seq $rounds); do
  for arm in "$@"; do
    f=gpurun_out/${tag}_tmp.json
    if [ "$arm" = "default" ]; then python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $f 2>/dev/null
    else env $arm python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $f 2>/dev/null; fi
    python3 - "$f" "$arm" >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d.get("hbm_leg") or {}
ent = "  ".join(f"{e['entry'][7:]} {e['ms_per_step']*1e3:.0f}" for e in h.get("entries", []))
print(f"{sys.argv[2]:28s} {d['ms_per_step']:7.3f} ms | hbm {h.get('ms_per_step', 0):.3f} | {ent}")
PY
  done
done
cat $out
