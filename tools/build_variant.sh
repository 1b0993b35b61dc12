#!/bin/bash
# Build feedback_gnn_amd/lib/ab/libfgnn_hip_<tag>.so with ONE kernel file recompiled under extra flags (macros), for A/B timing:
#   tools/build_variant.sh w6 fgnn_bp4.hip -DFGNN_BP4_WAVES=6 ;  FGNN_LIB_PATH=feedback_gnn_amd/lib/ab/libfgnn_hip_w6.so python tools/ab_bp4.py
set -e
tag=$1; f=$2; shift 2
cd "$(dirname "$0")/../feedback_gnn_amd/csrc"
make -s
mkdir -p ../lib/ab /tmp/fgnn_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -c $f -o /tmp/fgnn_ab/$tag.o
objs=$(ls ../lib/obj/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fgnn_ab/$tag.o -o ../lib/ab/libfgnn_hip_$tag.so
echo built ../lib/ab/libfgnn_hip_$tag.so
