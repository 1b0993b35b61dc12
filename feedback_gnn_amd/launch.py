"""Start one process per GPU without a launcher (`python script.py --gpus N`), the way the reference pins one process per GPU id
(/root/reference n882.py:9,15-21, n1270.py:10) — but as ranks of ONE torch.distributed job over one global sample stream.

The parent never touches a GPU: it only counts devices (`torch.cuda.device_count()` does not initialise one), starts fresh
interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them, and ends the remaining ranks (exact
PIDs) if one fails, so nobody sits in a collective until its timeout.  Nothing here imports the HIP library."""
import os
import socket
import subprocess
import sys
import tempfile
import time


def visible_gpus():
    import torch
    return torch.cuda.device_count()


def spawn_ranks(script, argv, world_size, capture_rank0=False):
    """Run ``python script argv...`` as ranks 0..world_size-1 on 127.0.0.1.  Returns ``(exit_codes, rank0_stdout or None)``.
    With ``capture_rank0`` rank 0's stdout is collected (the other ranks' stdout is dropped), else every rank inherits stdout."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out0 = tempfile.TemporaryFile(mode="w+") if capture_rank0 else None
    procs = []
    for r in range(world_size):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        stdout = None
        if capture_rank0:
            stdout = out0 if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env, stdout=stdout))
    codes = [None] * world_size
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.1)
    text = None
    if capture_rank0:
        out0.seek(0)
        text = out0.read()
        out0.close()
    return codes, text
