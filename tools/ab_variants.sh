#!/bin/bash
# Time every A/B build in feedback_gnn_amd/lib/ab/ with one probe script (default tools/ab_bp4.py) on the GPU box
probe=${1:-tools/ab_bp4.py}; shift || true
for so in feedback_gnn_amd/lib/ab/libfgnn_hip_*.so; do
  FGNN_LIB_PATH=$so python $probe "$@" 2>&1 | grep "^\["
done
python $probe "$@" 2>&1 | grep "^\["
