#!/bin/bash
# development aid: build a variant of libcvo_hip.so into tmp_libs/  usage: build_variant.sh <name> [extra hipcc -D flags...]
set -e
name=$1; shift
cd "$(dirname "$0")/../cvo_slam_amd/csrc"
out=/tmp/variant_$name; mkdir -p $out ../../tmp_libs
F="-DCVO_WAVES_PER_SIMD=2 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function $*"
for f in cvo_kernels cvo_score_kernels cvo_pcd_kernels cvo_selftest cvo_capi; do hipcc $F -c $f.hip -o $out/$f.o & done
hipcc $F -DCVO_KNS=cvohip_w3 -UCVO_BLOCK_MAX -DCVO_BLOCK_MAX=768 -c cvo_kernels.hip -o $out/cvo_kernels_w3.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_libs/libcvo_hip_$name.so $out/*.o
echo built tmp_libs/libcvo_hip_$name.so
