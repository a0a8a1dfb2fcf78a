#!/bin/bash
# A/B of two builds of the library on the same box, interleaved rounds:  tools/ab_bench.sh "<op_bench args>" [rounds]
# A = unet_dc_segmentation_amd/libunetdc_hip_base.so (kept copy of the previous build), B = the current build.
args=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  UNETDC_LIB=$PWD/unet_dc_segmentation_amd/libunetdc_hip_base.so python3 tools/op_bench.py $args 30 | sed 's/^/A  /'
  python3 tools/op_bench.py $args 30 | sed 's/^/B  /'
done
