"""MI355X-native BP4 + feedback-GNN decoder for CSS quantum LDPC codes.

Public names mirror `sionna.fec.ldpc` of the reference (sionna/fec/ldpc/__init__.py:10-13):
    QLDPCBPDecoder, Feedback_GNN, Sandwich_BP_GNN_Evaluation_Model, load_weights, save_weights,
    css_code and the code constructions of codes_q.
Importing this package needs neither a GPU nor the compiled library; constructing a decoder does
(there is no CPU fallback — see feedback_gnn_amd/_lib.py).
"""
from .codes_q import *  # noqa: F401,F403
from .codes_q import css_code  # noqa: F401
from .gf2 import row_echelon, rank, kernel, row_basis, compute_code_distance, inverse, int2bin, int_mod_2  # noqa: F401
from .weights_io import save_weights, read_weight_list, write_weight_list  # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily so that host-only use stays light
    if name in ("QLDPCBPDecoder",):
        from .decoding_q import QLDPCBPDecoder
        return QLDPCBPDecoder
    if name in ("Feedback_GNN", "Sandwich_BP_GNN_Evaluation_Model", "Pauli", "load_weights", "First_Stage_BP_Model",
                "Second_Stage_GNN_BP_Model"):
        from . import feedback_gnn as _f
        return getattr(_f, name)
    if name in ("sim_ber", "count_block_errors", "PlotBER", "allreduce_counts", "shard_range", "pack_decisions",
                "unpack_decisions", "gather_packed", "gather_decisions", "broadcast_weights"):
        from . import utils as _u
        return getattr(_u, name)
    if name in ("LDPCBPDecoder", "BP_BSC_Model"):
        from . import decoding as _d
        return getattr(_d, name)
    if name in ("OSD0_Decoder", "BP4_OSD_Model", "BP2_OSD_Model"):
        from . import bp_osd as _o
        return getattr(_o, name)
    if name in ("GNN_BP4", "MLP"):
        from . import gnn as _gn
        return getattr(_gn, name)
    if name in ("TannerGraph", "GnnWeights"):
        from . import graph as _g
        return getattr(_g, name)
    raise AttributeError(name)
