"""feedback_gnn_amd.launch.spawn_ranks — what `python bench.py --gpus N` / `examples/evaluate.py --gpus N` start their ranks with: the
environment of a torch.distributed job on 127.0.0.1 (checked by a real gloo all-reduce between the ranks), rank 0's stdout captured,
and a failing rank taking the others down instead of leaving them in a collective."""
import os
import sys
import textwrap
import time

from feedback_gnn_amd.launch import spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_ranks_rendezvous_and_rank0_stdout_is_captured(tmp_path):
    script = _script(tmp_path, """
        import os, sys, torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1, int(sys.argv[1])], dtype=torch.int64)
        dist.all_reduce(t)
        print(f"rank {rank} of {world}: {t.tolist()}")
        dist.destroy_process_group()
    """)
    codes, out = spawn_ranks(script, ["7"], 3, capture_rank0=True)
    assert codes == [0, 0, 0]
    lines = [l for l in out.splitlines() if not l.startswith("[Gloo]")]  # gloo prints a rendezvous banner on stdout
    assert lines == ["rank 0 of 3: [6, 21]"]  # 1 + 2 + 3 and 3 * 7; only rank 0's stdout comes back


def test_a_failing_rank_ends_the_job(tmp_path):
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)  # a rank that would wait in a collective for its peer
    """)
    t0 = time.time()
    codes, out = spawn_ranks(script, [], 2, capture_rank0=True)
    assert time.time() - t0 < 60, "the surviving rank must be ended, not waited for"
    assert codes[1] == 3 and codes[0] != 0


def test_eight_ranks_shard_the_sample_stream_and_reduce_like_bench(tmp_path):
    """The 8-rank job the driver starts (`bench.py --gpus 8`: one process per GPU) without a GPU: eight fresh interpreters rendezvous
    on 127.0.0.1 over gloo, each takes the batches `Sandwich_BP_GNN_Evaluation_Model(rank=r, world_size=8)` would take of the global
    Philox stream (consecutive blocks of world_size * batch, rank r the r-th batch of each block), decodes them with the CPU oracle
    standing in for the kernels, and the job reduces exactly as bench.py does: SUM of the three counters, MAX of the elapsed
    times, all-gather of the per-rank times.  The summed counters equal one process over the same global samples."""
    import json
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import llr_const, oracle_library_forms
    B, K, W, P, SEED = 24, 2, 1, 0.09, 0x5EED
    script = _script(tmp_path, f"""
        import os, sys, time, json, numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        from helpers import llr_const, oracle_library_forms
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        g = oracle_library_forms("gb48")
        nxt, counts = 0, np.zeros(3, np.int64)
        t0 = time.perf_counter()
        for step in range({W} + {K}):
            first = nxt + rank * {B}          # feedback_gnn_amd/feedback_gnn.py: _take_samples
            nxt += world * {B}
            if step < {W}:
                continue
            ex, ez = g.pauli_noise({SEED}, {P}, first, {B})
            sx, sz = g.syndrome(ex, ez)
            o = g.bp4_decode(sx, sz, 12, "boxplus-phi", 0.8, llr_const=llr_const(0.1))
            fl = g.residual(ex, ez, o["x_hat"], o["z_hat"])[2]
            counts += np.array([int((fl & 1).sum()), int(((fl >> 1) & 1).sum()), {B}], np.int64)
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        mx = el.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        parts = [torch.empty_like(el) for _ in range(world)]; dist.all_gather(parts, el)
        c = torch.from_numpy(counts); dist.all_reduce(c, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps(dict(world=world, counts=c.tolist(), max=float(mx), per_rank=[float(p) for p in parts])))
        dist.destroy_process_group()
    """)
    codes, out = spawn_ranks(script, [], 8, capture_rank0=True)
    assert codes == [0] * 8
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert d["world"] == 8 and len(d["per_rank"]) == 8 and abs(max(d["per_rank"]) - d["max"]) < 1e-12
    g = oracle_library_forms("gb48")
    lo, n = W * 8 * B, K * 8 * B  # the timed steps cover global samples [W*8B, (W+K)*8B)
    ex, ez = g.pauli_noise(SEED, P, lo, n)
    sx, sz = g.syndrome(ex, ez)
    o = g.bp4_decode(sx, sz, 12, "boxplus-phi", 0.8, llr_const=llr_const(0.1))
    fl = g.residual(ex, ez, o["x_hat"], o["z_hat"])[2]
    assert d["counts"] == [int((fl & 1).sum()), int(((fl >> 1) & 1).sum()), n] and d["counts"][0] > 0
