#!/bin/bash
# DIAGNOSTIC build of the library with cycle stamps in the tap-split ring weight gradient (-DUNETDC_WGRAD_STAMPS), kept OUT of
# the product library:  bash tools/probes/wgrad_stamps.sh   (here: cross-compile)  ->  unet_dc_segmentation_amd/libunetdc_hip_stamps.so
# On the GPU box:  UNETDC_LIB=$PWD/unet_dc_segmentation_amd/libunetdc_hip_stamps.so python3 tools/probes/wgrad_stamps.py
cd "$(dirname "$0")/../../unet_dc_segmentation_amd/csrc"
srcs=$(python3 -c "import re;s=open('../build.py').read();print(' '.join(re.search(r'SOURCES = \[(.*?)\]', s, re.S).group(1).replace('\"','').replace(',',' ').split()))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -fno-slp-vectorize -DUNETDC_WGRAD_STAMPS $srcs -o ../libunetdc_hip_stamps.so && echo built
