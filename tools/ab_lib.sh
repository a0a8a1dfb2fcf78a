#!/bin/bash
# A/B of two BUILDS on one box, arms interleaved (GPU box, repo root):  tools/ab_lib.sh <tag> [-- "<op_bench args>" ...]
# A = unet_dc_segmentation_amd/libunetdc_hip_base.so (tools/build_base.sh <rev>), B = the current build.
# Runs the op_bench lines given after "--" under both libraries, then the whole training step twice per arm.
tag=$1; shift
[ "$1" = "--" ] && shift
base=$PWD/unet_dc_segmentation_amd/libunetdc_hip_base.so
out=gpurun_out/${tag}_lib_ab.txt
: > $out
for spec in "$@"; do
  a=$(UNETDC_LIB=$base python3 tools/op_bench.py $spec 30 2>/dev/null | tail -1)
  b=$(python3 tools/op_bench.py $spec 30 2>/dev/null | tail -1)
  echo "A(base) $a" >> $out
  echo "B(new)  $b" >> $out
done
for r in 1 2; do
  for arm in A B; do
    f=gpurun_out/${tag}_${arm}_r${r}.json
    if [ $arm = A ]; then UNETDC_LIB=$base python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $f 2>/dev/null
    else python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $f 2>/dev/null; fi
    python3 - "$f" "$arm" >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d.get("step_ms") or {}
print(f"{sys.argv[2]} step {d['ms_per_step']:8.3f} ms ({d['value']:7.1f} img/s) median {s.get('median', 0):.3f} p10 {s.get('p10', 0):.3f} p90 {s.get('p90', 0):.3f} | dominant {d['roofline']['kernel']} {d['roofline']['achieved']:.0f} TF")
PY
  done
done
cat $out
