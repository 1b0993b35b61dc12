"""bench.py's side of the driver contract: the accounting it prices the dominant kernel with, and the one JSON line it prints."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_bytes_are_the_survey_figures():
    """16E + 12n + 4m per codeword-iteration = 98 784 B ([[882,24]]) / 142 240 B ([[1270,28]]); epilogue
    47 628 B / 68 580 B; 64 iterations of [[882,24]] = 6.37 MB per codeword, i.e. 1.256 M codewords/s at 8 TB/s."""
    import bench
    for n, m, E, per_iter, epi in ((882, 882, 5292, 98784, 47628), (1270, 1270, 7620, 142240, 68580)):
        assert bench.algorithmic_bytes_per_codeword(n, m, E, 1) - bench.algorithmic_bytes_per_codeword(n, m, E, 0) == per_iter
        assert bench.algorithmic_bytes_per_codeword(n, m, E, 0) == epi
    b64 = bench.algorithmic_bytes_per_codeword(882, 882, 5292, 64)
    assert b64 == 64 * 98784 + 47628 and abs(b64 / 1e6 - 6.37) < 0.005
    assert abs(bench.HBM_PEAK_GBS * 1e9 / b64 / 1e6 - 1.256) < 0.001


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    """A small run of the real bench (same code path as the driver's, batch 2 048): one JSON line with every contract field,
    the roofline and cpu_baseline objects, and a GPU result that equals the CPU oracle's on the sampled codewords."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2048",
                          "--cpu-sample", "256", "--no-extras"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "codewords/s" and "[[882,24]]" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2 * 2048 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"].startswith("valu-issue") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "traffic" in r
    gn = r["gnn"]  # the second kernel's roofline: f32 MFMA, measured in the same run with the same HIP-event recorder
    assert gn["bound"] == "mfma" and gn["unit"] == "TFLOP/s" and gn["peak"] == 157.3 and gn["launches_timed"] == 2
    assert gn["algorithmic_flops_per_launch"] == 13406400 * 2048
    assert abs(gn["achieved"] - gn["algorithmic_flops_per_launch"] / (gn["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * gn["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches_timed"] == 2
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "codewords/s" and c["cores"] >= 1 and c["value"] > 0 and "256 codewords" in c["sample"]
    assert c["gpu_matches_oracle_bit_exact"] is True
    assert d["counts"]["samples"] == 2 * 2048


def test_gnn_flops_are_the_survey_figures():
    """13.4 MFLOP ([[882,24]]) / 19.3 MFLOP ([[1270,28]]) per feedback-GNN pass and codeword."""
    import bench
    assert bench.gnn_flops_per_codeword(882, 5292) == 13406400
    assert bench.gnn_flops_per_codeword(1270, 7620) == 19304000


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts N ranks itself; with fewer visible GPUs than ranks it must fail loudly instead of
    degenerating to one rank (this container has no GPU at all, so --gpus 2 must be refused before anything is spawned)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FGNN_BENCH_BACKEND")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-build"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode != 0 and "--gpus 2" in res.stderr and res.stdout.strip() == ""
    # a launcher's WORLD_SIZE that disagrees with --gpus is an error too, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-build"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=env2)
    assert res.returncode != 0 and "WORLD_SIZE=4" in res.stderr


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent spawns two ranks before touching the GPU and relays rank 0's
    line.  On the one-GPU test box the ranks share the device and reduce over gloo (FGNN_BENCH_BACKEND); the flow — rendezvous,
    sharded sample stream, MAX of the times, SUM of the counters — is the one the nccl run uses."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FGNN_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "2048", "--p", "0.1", "--cpu-sample", "0", "--no-extras"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4096 and d["counts"]["samples"] == 2 * 2 * 2048
    assert abs(d["value"] - 2 * 4096 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    # the same 8 192 global samples in one process give the same counters (sharding by global sample index)
    res1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4096",
                           "--p", "0.1", "--cpu-sample", "0", "--no-extras", "--no-build"], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res1.returncode == 0, res1.stderr[-2000:]
    d1 = json.loads([l for l in res1.stdout.splitlines() if l.strip()][0])
    assert d1["counts"] == d["counts"] and d["counts"]["block_errors"] > 0
