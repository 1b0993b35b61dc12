"""RCCL on the GPU box: the nccl backend of torch.distributed (= RCCL on ROCm) is initialised on the box's GPU and carries the
two collectives of the multi-GPU path on device tensors — the counter all-reduce and the all-gather of
bit-packed decisions.  One rank: the box has one GPU; what this pins is that the RCCL code path (communicator creation,
all_reduce / all_gather_into_tensor kernels on a HIP stream, the bit-pack / unpack kernels) runs and returns the right bytes.
The multi-rank logic itself is covered over gloo (tests/test_distributed_cpu.py, tests/test_gpu_dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_pack_unpack_kernels_match_numpy():
    from feedback_gnn_amd.utils import pack_decisions, unpack_decisions
    rng = np.random.default_rng(7)
    for B, n in ((1, 7), (5, 48), (33, 882), (64, 1270)):  # 2n not a multiple of 8 included (n = 7: 14 bits)
        x = rng.integers(0, 2, size=(B, n), dtype=np.uint8)
        z = rng.integers(0, 2, size=(B, n), dtype=np.uint8)
        packed = pack_decisions(torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda())
        assert packed.shape == (B, (2 * n + 7) // 8)
        assert np.array_equal(packed.cpu().numpy(), np.packbits(np.concatenate([x, z], axis=1), axis=1))
        x2, z2 = unpack_decisions(packed, n)
        assert np.array_equal(x2.cpu().numpy(), x) and np.array_equal(z2.cpu().numpy(), z)
    with pytest.raises(ValueError):
        pack_decisions(torch.zeros((2, 8), dtype=torch.uint8), torch.zeros((2, 8), dtype=torch.uint8))  # host tensors: no CPU path


def test_nccl_allreduce_counts_on_device(nccl_group):
    """`allreduce_counts` on a CUDA tensor takes the RCCL branch (utils.py); with world size 1 it must call the collective
    (not short-circuit) and leave the counters unchanged."""
    from feedback_gnn_amd import utils
    assert nccl_group.get_backend() == "nccl"
    c = torch.tensor([3, 1, 4096], dtype=torch.int64, device="cuda")
    out = utils.allreduce_counts(c, force=True)
    torch.cuda.synchronize()
    assert out.is_cuda and out.tolist() == [3, 1, 4096]
    # the Monte-Carlo harness on top of it: dist=True runs one all-reduce per read-back on the nccl group
    import feedback_gnn_amd as F
    from helpers import code
    c8 = code("gb48")
    dec = F.QLDPCBPDecoder(code=c8, num_iter=12, normalization_factor=0.8, cn_type="boxplus-phi", stage_one=True)
    model = F.Sandwich_BP_GNN_Evaluation_Model(c8, [dec], [], num_layers=1, p0=0.1)
    F.sim_ber(model, [0.09], batch_size=500, max_mc_iter=2, verbose=False, dist=True, early_stop=False)
    a = {k: v.copy() for k, v in F.sim_ber.last.items() if k in ("flag_errors", "block_errors", "num_blocks")}
    model2 = F.Sandwich_BP_GNN_Evaluation_Model(c8, [dec], [], num_layers=1, p0=0.1)
    F.sim_ber(model2, [0.09], batch_size=500, max_mc_iter=2, verbose=False, dist=False, early_stop=False)
    for k in a:
        assert np.array_equal(a[k], F.sim_ber.last[k]), k
    assert a["num_blocks"][0] == 1000 and a["block_errors"][0] > 0


def test_nccl_gather_decisions_on_device(nccl_group):
    from feedback_gnn_amd.utils import gather_decisions, gather_packed, pack_decisions
    from helpers import gpu_graph
    g = gpu_graph("ghp882")
    ex, ez = g.pauli_noise(0x5EED, 0.05, 0, 257)  # any [B, n] bit arrays do; 257 rows: not a multiple of anything
    packed = pack_decisions(ex, ez)
    all_packed = gather_packed(packed)
    torch.cuda.synchronize()
    assert all_packed.is_cuda and torch.equal(all_packed, packed)
    xa, za = gather_decisions(ex, ez)
    assert torch.equal(xa, ex) and torch.equal(za, ez)


def test_nccl_broadcast_weights_on_device(nccl_group):
    """`broadcast_weights` (north star: "RCCL broadcast"): the feedback GNN's 3 923 parameters as ONE broadcast of a device tensor on
    the nccl group; at world size 1 the collective still runs (not short-circuited) and the weights — host copy and device tables —
    are what they were, so the decoder's output is unchanged bit for bit."""
    import feedback_gnn_amd as F
    from feedback_gnn_amd.utils import broadcast_weights
    from helpers import WEIGHTS_882, code, llr_const
    c = code("ghp882")
    dec = F.QLDPCBPDecoder(code=c, num_iter=8, normalization_factor=1.0, cn_type="boxplus-phi", stage_one=True)
    G = F.Feedback_GNN(code=c, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean", activation="tanh", use_bias=True,
                       graph=dec.graph)
    F.load_weights(G, WEIGHTS_882)
    g = dec.graph
    ex, ez = g.pauli_noise(0x5EED, 0.1, 0, 64)
    sx, sz = g.syndrome(ex, ez)
    o = g.bp4_decode(sx, sz, 8, "boxplus-phi", 1.0, llr_const=llr_const(0.05))
    before = g.feedback_gnn(G.device_weights, o["llr"], o["z_logit"], o["x_logit"], sx, sz).clone()
    w0 = [w.copy() for w in G.get_weights()]
    assert broadcast_weights(G, src=0) is G
    torch.cuda.synchronize()
    assert all(np.array_equal(a, b) for a, b in zip(w0, G.get_weights())) and G.count_params() == 3923
    after = g.feedback_gnn(G.device_weights, o["llr"], o["z_logit"], o["x_logit"], sx, sz)
    assert torch.equal(before, after)
