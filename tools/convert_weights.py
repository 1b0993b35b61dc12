#!/usr/bin/env python3
"""Convert the reference's trained feedback-GNN weight pickles to neutral .npz files.

Runs in the build container only (reads /root/reference/sionna/fec/ldpc/weights/*.npy with the
restricted unpickler of feedback_gnn_amd/weights_io.py — no TensorFlow needed) and writes
feedback_gnn_amd/weights/<same stem>.npz: 12 float32 arrays, 3 923 parameters each.
"""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from feedback_gnn_amd.weights_io import read_weight_list, write_weight_list  # noqa: E402

SRC = "/root/reference/sionna/fec/ldpc/weights"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "feedback_gnn_amd", "weights")

for p in sorted(glob.glob(os.path.join(SRC, "*.npy"))):
    w = read_weight_list(p)
    assert len(w) == 12 and sum(a.size for a in w) == 3923, p
    out = os.path.join(DST, os.path.splitext(os.path.basename(p))[0] + ".npz")
    write_weight_list(w, out)
    print(os.path.basename(out), [a.shape for a in w])
