"""World-size-2 gloo test of the multi-GPU path's logic: the Monte-Carlo sample stream is sharded by
global sample index with no data-path collective, and only the three counters are all-reduced.

The decoder stand-in on CPU is the oracle (there is no CPU product path); what is under test is
feedback_gnn_amd.utils.{shard_range, allreduce_counts} and the property that any sharding of the Philox
stream reproduces the single-process samples and counts exactly.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feedback_gnn_amd.utils import allreduce_counts, broadcast_weights, gather_packed, shard_range
from helpers import llr_const, oracle_library_forms

TOTAL, P, SEED = 601, 0.09, 0x5EED  # odd: the two shards differ by one row


def _counts_for(lo, hi):
    g = oracle_library_forms("gb48")
    ex, ez = g.pauli_noise(SEED, P, lo, hi - lo)
    sx, sz = g.syndrome(ex, ez)
    o = g.bp4_decode(sx, sz, 12, "boxplus-phi", 0.8, llr_const=llr_const(0.1))
    flags = g.residual(ex, ez, o["x_hat"], o["z_hat"])[2]
    packed = np.packbits(np.concatenate([o["x_hat"], o["z_hat"]], axis=1), axis=1)
    return np.array([(flags & 1).sum(), ((flags >> 1) & 1).sum(), hi - lo], dtype=np.int64), packed


class _Weights:
    """Stand-in with the get_weights / set_weights contract of Feedback_GNN (12 arrays in Keras order, 3 923 parameters)."""
    SHAPES = [(40, 3), (3,), (4, 40), (40,), (40, 20), (20,), (4, 40), (40,), (40, 20), (20,), (43, 40), (40,)]

    def __init__(self, seed, extra=False):
        rng = np.random.RandomState(seed)
        self.w = [rng.uniform(-1, 1, size=s).astype(np.float32) for s in self.SHAPES + ([(5,)] if extra else [])]

    def get_weights(self):
        return self.w

    def set_weights(self, w):
        assert [a.shape for a in w] == [a.shape for a in self.w]
        self.w = [np.asarray(a, dtype=np.float32) for a in w]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(TOTAL, rank, world)
        c, packed = _counts_for(lo, hi)
        counts = allreduce_counts(torch.from_numpy(c.copy()))
        # product all-gather of the bit-packed decisions (2n bits per codeword) onto every rank, uneven shards included;
        # the rows are packed by numpy here (the HIP bit-pack kernel is covered on the GPU box, tests/test_gpu_rccl.py)
        gathered = gather_packed(torch.from_numpy(packed))
        # rank 1 alone holds the weights (as after training there): one broadcast puts them on every rank; a rank with another
        # architecture is refused before anything is overwritten
        holder = _Weights(seed=100 + rank)
        broadcast_weights(holder, src=1)
        odd = _Weights(seed=7, extra=rank == 0)
        try:
            broadcast_weights(odd, src=1)
            refused = False
        except ValueError:
            refused = True
        q.put((rank, counts.numpy().copy(), gathered.numpy().copy(), [w.copy() for w in holder.get_weights()], refused,
               [w.copy() for w in odd.get_weights()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_counts, ref_packed = _counts_for(0, TOTAL)
    assert ref_counts[1] > 0, "the test point must produce block errors"
    src_weights = _Weights(seed=101).get_weights()  # what rank 1 held
    assert sum(w.size for w in src_weights) == 3923
    for rank, counts, packed, weights, refused, odd in results:
        assert np.array_equal(counts, ref_counts), (rank, counts, ref_counts)
        assert np.array_equal(packed, ref_packed)
        assert len(weights) == 12 and all(np.array_equal(a, b) for a, b in zip(weights, src_weights)), rank
        assert refused and all(np.array_equal(a, b) for a, b in zip(odd, _Weights(seed=7, extra=rank == 0).get_weights())), rank
