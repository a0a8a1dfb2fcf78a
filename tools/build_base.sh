#!/bin/bash
# Build libunetdc_hip_base.so from the kernel sources of a git revision (default HEAD): the "A" side of tools/ab_bench.sh
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive $rev unet_dc_segmentation_amd/csrc include | tar -x -C $tmp
srcs=$(python3 - <<PY
import re,subprocess
s=subprocess.run(["git","show","$rev:unet_dc_segmentation_amd/build.py"],capture_output=True,text=True).stdout
print(" ".join(re.search(r'SOURCES = \[(.*?)\]', s, re.S).group(1).replace('"','').replace(',',' ').split()))
PY
)
cd $tmp/unet_dc_segmentation_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -fno-slp-vectorize $srcs -o $OLDPWD/unet_dc_segmentation_amd/libunetdc_hip_base.so && echo "base built from $rev"
rm -rf $tmp
