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
