"""Two ranks (gloo, sharing the one GPU of the test box) run the sharded Monte-Carlo loop: sim_ber(dist=True) must produce,
on every rank, the counts of the single-process run over the same global sample stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
P, BATCH, ITERS = 0.13, 600, 3


def _build_model(rank, world):
    import feedback_gnn_amd as F
    from helpers import WEIGHTS_882, code
    c = code("ghp882")
    d0 = F.QLDPCBPDecoder(code=c, num_iter=32, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True)
    d1 = F.QLDPCBPDecoder(code=c, num_iter=16, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True, graph=d0.graph)
    G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh",
                       use_bias=True, graph=d0.graph)
    F.load_weights(G, WEIGHTS_882)
    return F.Sandwich_BP_GNN_Evaluation_Model(c, [d0, d1], [G], num_layers=2, rank=rank, world_size=world)


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import feedback_gnn_amd as F
        model = _build_model(rank, world)
        F.sim_ber(model, [P], batch_size=BATCH // world, max_mc_iter=ITERS, verbose=False, dist=True, early_stop=False)
        st = F.sim_ber.last
        # decisions of ALL ranks on every rank: HIP bit-pack kernel -> all-gather (through host memory on gloo) -> unpack kernel
        model2 = _build_model(rank, world)
        d = model2.decode(BATCH // world, P)
        xa, za = F.gather_decisions(d["x_hat"], d["z_hat"])
        import hashlib
        digest = hashlib.sha256(xa.cpu().numpy().tobytes() + za.cpu().numpy().tobytes()).hexdigest()
        q.put((rank, int(st["flag_errors"][0]), int(st["block_errors"][0]), int(st["num_blocks"][0]), tuple(xa.shape), digest))
    finally:
        dist.destroy_process_group()


def test_two_rank_sim_ber_matches_single_process():
    import feedback_gnn_amd as F
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _build_model(0, 1)
    F.sim_ber(model, [P], batch_size=BATCH, max_mc_iter=ITERS, verbose=False, early_stop=False)
    st = F.sim_ber.last
    ref = (int(st["flag_errors"][0]), int(st["block_errors"][0]), int(st["num_blocks"][0]))
    assert ref[1] > 0 and ref[2] == BATCH * ITERS
    import hashlib
    d = _build_model(0, 1).decode(BATCH, P)  # the same first BATCH global samples in one piece
    want = hashlib.sha256(d["x_hat"].cpu().numpy().tobytes() + d["z_hat"].cpu().numpy().tobytes()).hexdigest()
    for rank, fl, bl, nb, shape, digest in results:
        assert (fl, bl, nb) == ref, (rank, fl, bl, nb, ref)
        assert shape == (BATCH, 882) and digest == want, (rank, shape)
