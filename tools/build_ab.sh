#!/bin/bash
# Build feedback_gnn_amd/lib/ab/libfgnn_hip_old.so with the committed (HEAD) version of ONE kernel file, for A/B timing of a
# working-tree change against it:  tools/build_ab.sh fgnn_gnn.hip ;  FGNN_LIB_PATH=feedback_gnn_amd/lib/ab/libfgnn_hip_old.so python tools/ab_*.py
set -e
f=$1
cd "$(dirname "$0")/../feedback_gnn_amd/csrc"
make -s
mkdir -p ../lib/ab /tmp/fgnn_ab
git show HEAD:feedback_gnn_amd/csrc/$f > _ab_old_$f
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -c _ab_old_$f -o /tmp/fgnn_ab/old.o
rm -f _ab_old_$f
objs=$(ls ../lib/obj/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fgnn_ab/old.o -o ../lib/ab/libfgnn_hip_old.so
echo built ../lib/ab/libfgnn_hip_old.so
