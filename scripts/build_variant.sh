#!/bin/bash
# usage: build_variant.sh <name> "<EXTRA flags>"   -> /root/repo/build_variants/libvr_hip_<name>.so
set -e
HERE=/root/repo/volume-rendering_amd/csrc
SRCS="$HERE/vr_kernels.hip $HERE/vr_hip_api.cpp $HERE/host/HipRenderer.cpp $HERE/host/RaycasterBase.cpp $HERE/host/camera.cpp $HERE/host/ModelBase.cpp $HERE/host/frame_stats.cpp $HERE/host/vr_host_api.cpp"
[ -f $HERE/vr_multi.cpp ] && SRCS="$SRCS $HERE/vr_multi.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $2 -Rpass-analysis=kernel-resource-usage -x hip -shared -o /root/repo/build_variants/libvr_hip_$1.so $SRCS 2> /root/repo/build_variants/$1.log || { tail -20 /root/repo/build_variants/$1.log; exit 1; }
grep -A12 "raymarch_kernelILi1ELi1ELi0ELi1" /root/repo/build_variants/$1.log | grep -E "SGPRs:|VGPRs:|Occupancy|LDS Size" | head -4 | tr '\n' ' '; echo " <- $1"
