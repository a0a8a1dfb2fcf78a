#!/bin/bash
# Probe library for tools/probes/cu_mask_probe.py: the product sources with ONE change -- the CU count behind the persistent
# grid of the lattice kernels is read from UNETDC_CUS at every launch.  Built from a patched COPY; the tree is not touched.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/pkg $tmp/include
cp -r $root/unet_dc_segmentation_amd/csrc $tmp/pkg/csrc
cp $root/include/*.h $tmp/include/
rm -rf $tmp/pkg/csrc/_obj
cd $tmp/pkg/csrc
python3 - <<'PY'
s = open("igemm_lattice.hip").read()
a = "  const long g = 256L * wgs_per_cu;\n"
assert a in s
s = s.replace(a, '  const char* e_ = getenv("UNETDC_CUS");\n  const long g = (e_ ? atol(e_) : 256L) * wgs_per_cu;\n')
open("igemm_lattice.hip", "w").write(s)
PY
srcs=$(python3 -c "import re;s=open('$root/unet_dc_segmentation_amd/build.py').read();print(' '.join(re.search(r'SOURCES = \[(.*?)\]', s, re.S).group(1).replace('\"','').replace(',',' ').split()))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -fno-slp-vectorize $srcs -o $root/unet_dc_segmentation_amd/libunetdc_hip_probe.so
rm -rf $tmp
echo built
