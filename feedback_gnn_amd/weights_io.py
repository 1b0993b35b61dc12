"""Weight I/O for the feedback GNN.

Mirrors `save_weights` / `load_weights` of /root/reference sionna/fec/ldpc/gnn.py:755-791, which
pickle `system.get_weights()`.  The files the reference ships (`sionna/fec/ldpc/weights/*.npy`) are,
despite the suffix, pickles of a list of 12 `tf.Tensor`s whose reduce hook is
`tensorflow.python.framework.ops.convert_to_tensor(ndarray, ...)`.  They are read here WITHOUT
TensorFlow by a restricted unpickler that maps that one callable to `numpy.asarray` and refuses
every global outside a short NumPy allow-list, so loading a weight file cannot run arbitrary code.

Array order (Keras creation order, feedback_gnn.py:115-128):
  [W_out(40,3), b_out(3), Wx1(4,40), bx1(40), Wx2(40,20), bx2(20),
   Wz1(4,40), bz1(40), Wz2(40,20), bz2(20), We(43,40), be(40)]
"""
import io
import os
import pickle

import numpy as np

WEIGHT_NAMES = ("w_out", "b_out", "wx1", "bx1", "wx2", "bx2", "wz1", "bz1", "wz2", "bz2", "we", "be")

_BUNDLED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")


def _as_array(value, *args, **kwargs):
    return np.asarray(value)


class _TensorListUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"),
        ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"),
        ("numpy", "dtype"),
    }

    def find_class(self, module, name):
        if module.startswith("tensorflow.") and name in ("convert_to_tensor", "convert_to_tensor_v2"):
            return _as_array
        if module.startswith("tensorflow.") and name in ("as_dtype", "DType"):
            return lambda *a, **k: None
        if (module, name) in self._ALLOWED:
            mod = module.replace("numpy.core", "numpy._core") if not hasattr(np, "core") else module
            return getattr(__import__(mod, fromlist=[name]), name)
        raise pickle.UnpicklingError(f"refusing to unpickle global {module}.{name}")


def read_weight_list(path):
    """Return the list of float32 arrays stored at ``path`` (reference pickle, or .npz written by
    `write_weight_list`).  Names without a directory are also looked up in the bundled weights."""
    if not os.path.exists(path):
        for cand in (os.path.join(_BUNDLED, os.path.basename(path)),
                     os.path.join(_BUNDLED, os.path.splitext(os.path.basename(path))[0] + ".npz")):
            if os.path.exists(cand):
                path = cand
                break
        else:
            raise FileNotFoundError(path)
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:2] == b"PK":  # zip container -> npz
        z = np.load(io.BytesIO(blob))
        keys = sorted(z.files, key=lambda k: int(k.split("_")[0]))
        return [np.asarray(z[k], dtype=np.float32) for k in keys]
    arrays = _TensorListUnpickler(io.BytesIO(blob)).load()
    return [np.asarray(a, dtype=np.float32) for a in arrays]


def write_weight_list(arrays, path):
    """Store a weight list in a neutral format (.npz, keys ``<index>_<name>``)."""
    named = {}
    for i, a in enumerate(arrays):
        tag = WEIGHT_NAMES[i] if len(arrays) == len(WEIGHT_NAMES) else "w"
        named[f"{i:02d}_{tag}"] = np.asarray(a, dtype=np.float32)
    with open(path, "wb") as f:
        np.savez(f, **named)


def save_weights(system, model_path):
    """Save ``system.get_weights()`` (gnn.py:755-772).  Written as .npz, not as a pickle."""
    write_weight_list(system.get_weights(), model_path)


def load_weights(system, model_path):
    """Load weights into ``system`` via ``system.set_weights`` (gnn.py:774-791)."""
    system.set_weights(read_weight_list(model_path))
