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
    # the bound the kernel is subject to: VALU issue.  At this small batch profiles/traffic.json has no entry (the PMC passes are
    # taken at the benchmark shape), so the instruction-count figures are null WITH a reason, never a stale number
    assert r["bound"] == "valu" and r["unit"] == "G wave-instructions/s" and r["peak"] == 1228.8 and "traffic" in r
    assert r["traffic"] is None and r["frac"] is None and "no entry" in r["traffic_source"]
    assert r["launches_timed"] == 2 and r["hbm_peak_GBs"] == 8000.0
    eff = r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(r["effective_bandwidth_GBs"] - eff) < 1e-6 * eff and abs(r["effective_bandwidth_frac"] - eff / 8000.0) < 1e-9
    ch = r["contract_hbm"]  # the same figure in the contract's own shape, labelled effective
    assert ch["bound"] == "hbm" and ch["unit"] == "GB/s" and ch["peak"] == 8000.0 and ch["effective"] is True
    assert ch["achieved"] == r["effective_bandwidth_GBs"] and abs(ch["frac"] - r["effective_bandwidth_frac"]) < 1e-12 and ch["traffic"] == r["traffic"]
    # the second kernel's roofline, timed in the same run by the same HIP-event recorder: at 2 048 codewords the library runs the feedback
    # GNN on its MFMA tiles (the streaming VALU kernel takes over from 4 096 on), priced in the reference's FLOPs against the f32 peak
    gn = r["gnn"]
    assert gn["bound"] == "mfma" and gn["unit"] == "TFLOP/s" and gn["peak"] == 157.3 and gn["launches_timed"] == 2
    assert "MFMA-tile" in gn["kernel"] and gn["traffic"] is None and "no entry" in gn["traffic_source"]  # no PMC counts at this small batch
    assert abs(gn["frac"] - gn["reference_tflops_frac_of_f32_peak"]) < 1e-12 and gn["achieved"] == gn["reference_tflops"]
    assert gn["algorithmic_flops_per_launch"] == gn["executed_flops_per_launch"] == 13406400 * 2048  # literal association: one Dense per edge
    ref_tf = gn["algorithmic_flops_per_launch"] / (gn["avg_launch_ms"] * 1e-3) / 1e12
    assert abs(gn["reference_tflops"] - ref_tf) < 1e-6 * ref_tf
    assert abs(gn["reference_tflops_frac_of_f32_peak"] - ref_tf / 157.3) < 1e-9 and gn["executed_frac"] <= gn["reference_tflops_frac_of_f32_peak"] <= 1.0
    # round 6: the headline is the library default = the reference's formulas term by term
    assert d["config"]["gnn_association"] == "literal" and d["config"]["bp4_qubit_update_lse"] == "per edge (literal)" and "term by term" in d["value_is"]
    assert d["config"]["gnn_kernel"] == "MFMA tiles"  # 2 048 < 4 096 codewords: the library's own choice
    assert r["mfma_insts_per_launch"] is None and r["mfma_busy_frac"] is None and "mfma_busy_frac" in gn  # no PMC entry at this batch: nulls, never a stale number
    assert d["per_rank_ms"] == [d["ms_per_step"]]
    sh = d["dist"]["sharding"]  # one rank: its two timed batches are samples [2048, 6144)
    assert d["dist"]["ranks"][0]["timed_samples"] == [2048, 3 * 2048] == sh["timed_region_samples"] and sh["batches"] == 2
    assert sh["ranges_tile_the_region_without_overlap"] is True and sh["sum_of_rank_counts_equals_all_reduced"] is True
    assert d["dist"]["ranks"][0]["own_counts"] == d["counts"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "codewords/s" and c["cores"] >= 1 and c["value"] > 0 and "256 codewords" in c["sample"]
    assert c["gpu_matches_oracle_bit_exact"] is True
    t = d["cpu_baseline_tf_like"]
    assert t["kind"] == "port" and t["unit"] == "codewords/s" and t["cores"] >= 1 and t["value"] > 0 and "256 codewords" in t["sample"]
    assert t["decisions_identical_to_oracle_on_samples_both_decode"] >= 0.95 and t["samples_both_decode"] >= 250
    assert d["counts"]["samples"] == 2 * 2048
    # literal_forms = the headline itself (the key of earlier rounds); reassociated_forms (with --no-extras at top level, else under extras) =
    # the same step under the two opt-in re-associations, timed in the same run, with forms_agreement = the first timed batch decoded
    # under both forms, compared per sample (on THIS window of 2 048 samples at p = 0.01 they agree; that is a measurement, not a guarantee)
    lf, rf = d["literal_forms"], d["reassociated_forms"]
    assert lf["value"] == d["value"] and lf["ms_per_step"] == d["ms_per_step"] and lf["unit"] == "codewords/s" and lf["steps"] == 2
    assert rf["unit"] == "codewords/s" and rf["steps"] == 2 and rf["value"] > 0 and "opt-in" in rf["forms"].lower()
    assert abs(rf["value"] - 2 * 2048 / (rf["ms_per_step"] * 2e-3)) < 1e-6 * rf["value"]
    fa = rf["forms_agreement"]
    assert fa["samples"] == 2048 and fa["p"] == 0.01 and fa["first_sample"] == 2048 and fa["decisions_differ"] == 0 and fa["max_abs_dllr_solved"] <= 1e-4
    assert "forms_agreement" not in d
    assert set(fa["first_decoder"]) == {"decisions_differ", "max_abs_dllr", "samples_gt_1e_4"}
    # algorithmic efficiency next to issue utilisation: every exp / log of the fixed dataflow against the hardware transcendental rate
    assert r["transcendental_evals_per_codeword"] == {"exp": 64 * 20 * 882 + 4 * 882 + 5292 + 882, "log": 64 * 32 * 882 + 4 * 882 + 2 * (5292 + 882)}
    assert r["transcendental_evals_per_launch"] == 2048 * sum(r["transcendental_evals_per_codeword"].values())
    hw = r["transcendental_evals_per_launch"] / (r["avg_launch_ms"] * 1e-3) / r["hw_transcendental_peak_per_s"]
    assert abs(r["frac_of_hw_transcendental_rate"] - hw) < 1e-9 and r["hw_transcendental_peak_per_s"] == 1024 * 64 * 2.4e9 / 8


def test_config_shapes_and_the_algorithmic_counts_of_the_other_two_configs():
    """--config c3 | c4 | c5 = BASELINE.json configs[2] / [3] / [4] at their per-GPU shard; the per-codeword figures the rooflines are
    priced with: exp / log evaluations of a fixed-dataflow BP4 decode (decoding_q.py:254-273, 365-431, 455-471) and the GNN_BP4 FLOPs of
    SURVEY.md §8(d) (87.5 MFLOP per [[1270,28]] codeword-iteration, 0.83 GFLOP for 10 iterations with the last check update skipped)."""
    import bench
    a = bench.parse_args([])
    assert (a.config, a.code, a.iters, a.batch) == ("c3", "ghp882", "64,16", 65536)
    assert a.p == 0.01
    a = bench.parse_args(["--config", "c1"])  # configs[0]: the reference's CPU-runnable case (32 iterations, 256 codewords, p = 0.05)
    assert (a.code, a.iters, a.batch, a.p) == ("ghp882", "32", 256, 0.05) and bench.parse_args(["--config", "c1", "--p", "0.02"]).p == 0.02
    assert (a.cn_type, a.factor) == ("boxplus", 0.625)  # the reference's constructor defaults (decoding_q.py:18-22)
    a = bench.parse_args(["--config", "c1", "--cn-type", "boxplus-phi"])  # the QLDPC.ipynb cell 11 helper's variant
    assert (a.cn_type, a.factor) == ("boxplus-phi", 0.625)
    a = bench.parse_args([])
    assert (a.cn_type, a.factor, a.streams) == ("boxplus-phi", 1.0, 1)  # n882.py:56-62
    # the workloads of the reference's only published timings: n882.py:39,56-66 / n1270.py:57-70 with nG = 3, 5
    a = bench.parse_args(["--config", "n882_3r"])
    assert (a.code, a.iters, a.batch, a.p) == ("ghp882", "64,16,16,16", 5000, 0.05)
    a = bench.parse_args(["--config", "n882_5r"])
    assert (a.code, a.iters, a.batch, a.p) == ("ghp882", "64,16,16,16,16,16", 5000, 0.05)
    a = bench.parse_args(["--config", "n1270_3r"])
    assert (a.code, a.iters, a.batch, a.p) == ("ghp1270", "64,16,16,16", 5000, 0.07)
    a = bench.parse_args(["--config", "qldpc_882"])  # plain BP4 as QLDPC.ipynb cell 12 runs it (29.7 k cw/s published)
    assert (a.code, a.iters, a.batch, a.p, a.cn_type, a.factor, a.p0) == ("ghp882", "64", 10000, 0.01, "boxplus-phi", 0.8, 0.3)
    a = bench.parse_args(["--config", "qldpc_1270"])
    assert (a.code, a.batch, a.p0) == ("ghp1270", 10000, 0.3) and bench.parse_args([]).p0 == 0.05
    a = bench.parse_args(["--config", "c2"])  # configs[1]: BP4-64 alone
    assert (a.code, a.iters, a.batch, a.p) == ("ghp882", "64", 65536, 0.01)
    a = bench.parse_args(["--config", "c4"])
    assert (a.code, a.iters, a.batch) == ("ghp1270", "64,64", 32768) and 8 * a.batch == 262144
    a = bench.parse_args(["--config", "c5", "--batch", "128"])
    assert (a.code, a.iters, a.batch) == ("ghp1270", "10", 128) and 8 * bench.CONFIGS["c5"]["batch"] == 131072
    n, m, E = 882, 882, 5292
    e1, l1 = bench.bp4_transcendentals_per_codeword(n, m, E, 1, True)
    e0, l0 = bench.bp4_transcendentals_per_codeword(n, m, E, 0, True)
    assert (e1 - e0, l1 - l0) == (16 * n, 28 * n)       # shared form: 4 + 4 in the qubit update, 12 + 24 in the check update, per qubit
    e1, l1 = bench.bp4_transcendentals_per_codeword(n, m, E, 1, False)
    assert (e1 - e0, l1 - l0) == (20 * n, 32 * n)       # literal form: one log-sum-exp per edge
    assert (e0, l0) == (4 * n + E + m, 4 * n + 2 * (E + m))  # cal_logit
    e1, l1 = bench.bp4_transcendentals_per_codeword(n, m, E, 1, True, "boxplus")
    assert (e1 - e0, l1 - l0) == (4 * n, 4 * n + E)      # 'boxplus': the tanh is a rational, the atanh one log1p per edge (decoding_q.py:313-363)
    e1, l1 = bench.bp4_transcendentals_per_codeword(n, m, E, 1, True, "minsum")
    assert (e1 - e0, l1 - l0) == (4 * n, 4 * n)
    assert abs(bench.HW_TRANSCENDENTAL_PEAK - 1.966e13) < 1e10
    n, m, E = 1270, 1270, 7620
    per_it = bench.gnnbp4_flops_per_codeword(n, m, E, 2) - bench.gnnbp4_flops_per_codeword(n, m, E, 1)
    assert abs(per_it / 1e6 - 87.5) < 0.05
    f10 = bench.gnnbp4_flops_per_codeword(n, m, E, 10)
    assert abs(f10 / 1e9 - 0.83) < 0.005 and f10 == 10 * (E * 2 * 2400 + n * 2 * 3200) + 9 * (E * 2 * 2400 + m * 2 * 2440)
    assert bench.gnnbp4_flops_per_codeword_factored(n, m, E, 10) < f10
    # 131 072 codewords on 8 GPUs at the f32 peak: the survey's >= 0.087 s
    assert abs(f10 * 131072 / (8 * bench.GNN_PEAK_TFLOPS * 1e12) - 0.087) < 0.001


@pytest.mark.gpu
@pytest.mark.parametrize("config,batch", [("c4", 1024), ("c5", 256)])
def test_bench_lines_of_the_other_two_configs(config, batch):
    """`bench.py --config c4` ([[1270,28]] (64, G, 64)) and `--config c5` (GNN_BP4, 10 iterations) print the same contract line: roofline
    (c5: against the f32 MFMA peak in the reference's FLOPs) and cpu_baseline, GPU == oracle on the sampled codewords; at a batch
    that has no PMC entry `--require-roofline` turns the null fraction into exit code 5 AFTER the line has been printed."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "2", "--warmup", "1", "--batch", str(batch),
                          "--cpu-sample", "32", "--no-extras", "--require-roofline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 5 and "--require-roofline" in res.stderr, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert "[[1270,28]]" in d["metric"] and d["unit"] == "codewords/s" and d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == batch and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2 * batch / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"] and d["counts"]["samples"] == 2 * batch
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["frac"] is None and "no entry" in r["traffic_source"] and r["launches_timed"] == 2 and r["avg_launch_ms"] > 0
    assert c["kind"] == "port" and c["value"] > 0 and c["gpu_matches_oracle_bit_exact"] is True and "32 codewords" in c["sample"]
    assert d["cpu_baseline_tf_like"]["value"] > 0
    fa = d["reassociated_forms"]["forms_agreement"]
    assert fa["samples"] == batch and d["literal_forms"]["value"] == d["value"] and d["config"]["gnn_association"] == "literal"
    if config == "c5":
        # untrained (seeded) weights leave marginals within 1e-5 of the argmin boundary: the decisions of such qubits may flip under the
        # 1e-6 rounding difference of the two associations; none may flip beyond the LLR tolerance
        assert fa["decisions_differ_beyond_llr_tolerance"] == 0
        assert fa["max_decision_margin_where_they_differ"] <= 2e-4
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and d["config"]["gnn_bp4_iters"] == 10
        assert r["algorithmic_flops_per_launch"] == batch * (10 * (7620 * 2 * 2400 + 1270 * 2 * 3200) + 9 * (7620 * 2 * 2400 + 1270 * 2 * 2440))
        tf = r["algorithmic_flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12
        assert abs(r["reference_tflops"] - tf) < 1e-6 * tf and fa["max_abs_dllr"] <= 1e-4
    else:
        assert fa["decisions_differ"] <= 1  # a measured rate on this window of 1 024 samples at p = 0.01, not a guarantee
        assert r["bound"] == "valu" and d["config"]["bp_iters"] == [64, 64] and "configs[3]" in d["config"]["workload"]
        assert r["later_decoders_avg_launch_ms"] > 0 and r["gnn"]["launches_timed"] == 2


@pytest.mark.gpu
def test_bench_lines_of_the_bp4_only_configs():
    """`--config c1` (BASELINE configs[0]: BP4-32 on 256 codewords at p = 0.05 — at the configuration's own shape, so its PMC entry
    applies and --require-roofline passes) and `--config c2` (configs[1]: BP4-64 alone; here at a batch without an entry): one decoder
    launch per step, no feedback-GNN object in the roofline, the literal-forms leg and the per-sample agreement like the headline."""
    import bench
    for extra, key, cn in (([], "bp4_ghp882_it32_B256_boxplus", "boxplus"), (["--cn-type", "boxplus-phi"], "bp4_ghp882_it32_B256", "boxplus-phi")):
        # configs[0] as the reference constructs it (cn_type='boxplus', normalization_factor=0.625: decoding_q.py:18-22), then the
        # QLDPC.ipynb cell 11 helper's 'boxplus-phi' variant of the same case
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "4", "--warmup", "1", "--cpu-sample", "32",
                              "--no-extras", "--require-roofline"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
        d = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
        r = d["roofline"]
        ent, _ = bench.pmc_entry("bp4", key)
        assert res.returncode == (0 if ent else 5), res.stderr[-2000:]
        assert d["metric"] == "decoded codewords/sec, [[882,24]] BP4 32 iters" and "configs[0]" in d["config"]["workload"]
        assert d["config"]["bp_iters"] == [32] and d["config"]["p"] == 0.05 and d["config"]["batch_per_gpu"] == 256
        assert d["config"]["cn_type"] == cn and d["config"]["normalization_factor"] == 0.625 and f"cn_type={cn}" in d["config"]["workload"]
        assert f"bp4_kernel<{cn}>" in r["kernel"]
        assert r["gnn"] is None and r["launches_timed"] == 4 and r["later_decoders_avg_launch_ms"] is None
        assert (r["frac"] is not None and 0 < r["frac"] <= 1) if ent else r["frac"] is None
        fa = d["reassociated_forms"]["forms_agreement"]
        assert d["cpu_baseline"]["gpu_matches_oracle_bit_exact"] is True and fa["samples"] == 256
        assert fa["p"] == 0.05  # the agreement of the opt-in forms with the default (literal) ones at the configuration's own p
        assert d["cpu_baseline_tf_like"]["value"] > 0
        assert d["counts"]["samples"] == 4 * 256 and d["literal_forms"]["value"] > 0
        assert d["dist"]["world_size"] == 1 and d["dist"]["ranks"][0]["device_index"] == 0 and "value_is" in d
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "2", "--warmup", "1", "--batch", "2048",
                          "--cpu-sample", "32", "--no-extras", "--no-build"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
    assert d["metric"] == "decoded codewords/sec, [[882,24]] BP4 64 iters" and "BP4-64 alone" in d["config"]["workload"]
    assert d["roofline"]["gnn"] is None and d["roofline"]["frac"] is None and d["reassociated_forms"]["forms_agreement"]["decisions_differ"] <= 1
    assert d["cpu_baseline"]["gpu_matches_oracle_bit_exact"] is True


@pytest.mark.gpu
def test_bench_line_of_a_published_workload():
    """`--config n882_3r`: the workload of the reference's published 10.9 k codewords/s (examples/n882.ipynb cell 2: (64, G, 16, G, 16, G,
    16), batch_size 5 000, p = 0.05, n882.py:39,56-66) — one stream and two streams timed in the same run, literal forms and the
    per-sample agreement at p = 0.05, GPU == oracle on the sampled codewords."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "n882_3r", "--steps", "3", "--warmup", "1", "--cpu-sample", "32",
                          "--no-extras"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][0])
    assert d["config"]["bp_iters"] == [64, 16, 16, 16] and d["config"]["batch_per_gpu"] == 5000 and d["config"]["p"] == 0.05
    assert "3 x (feedback-GNN + BP4-16)" in d["metric"] and "n882.ipynb" in d["config"]["workload"] and d["vs_baseline"] is None
    assert d["counts"]["samples"] == 3 * 5000 and d["config"]["streams"] == 1
    t = d["two_streams"]
    assert t["streams"] == 2 and t["value"] > 0 and abs(t["value"] - 3 * 5000 / (t["ms_per_step"] * 3e-3)) < 1e-6 * t["value"]
    fa = d["reassociated_forms"]["forms_agreement"]
    assert d["literal_forms"]["value"] == d["value"] and fa["samples"] == 5000 and fa["p"] == 0.05
    r = d["roofline"]
    assert r["launches_timed"] == 3 and r["later_decoders_avg_launch_ms"] > 0 and r["gnn"]["launches_timed"] == 9
    assert "streaming VALU" in r["gnn"]["kernel"] and d["config"]["gnn_kernel"] == "streaming VALU"  # 5 000 >= 4 096 codewords
    assert d["cpu_baseline"]["gpu_matches_oracle_bit_exact"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("config,batch", [("c4", 512), ("c5", 128)])
def test_other_configs_shard_over_two_ranks(config, batch):
    """`--config c4 | c5 --gpus 2` (two ranks sharing the test box's GPU over gloo): the counters equal ONE process over the same 2 B
    global samples per step — sharding by global sample index, also for the GNN_BP4 line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FGNN_BENCH_BACKEND"] = "gloo"
    common = ["--config", config, "--steps", "2", "--warmup", "1", "--p", "0.13", "--cpu-sample", "0", "--no-extras"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", str(batch)] + common,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.lstrip().startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * batch and len(d["per_rank_ms"]) == 2
    assert d["counts"]["samples"] == 2 * 2 * batch and d["literal_forms"]["value"] == d["value"] and "reassociated_forms" not in d
    # the sharding proof of the multi-GPU line (round 6): rank r's timed batches are blocks r of every 2 consecutive blocks of `batch`
    sh, rows = d["dist"]["sharding"], d["dist"]["ranks"]
    assert [r["timed_samples"] for r in rows] == [[2 * batch, 2 * batch + 4 * batch - batch], [3 * batch, 6 * batch]]
    assert sh["timed_region_samples"] == [2 * batch, 6 * batch] and sh["batches"] == 4 and sh["ranges_tile_the_region_without_overlap"] is True
    assert sh["sum_of_rank_counts_equals_all_reduced"] is True and sh["sum_of_rank_counts"] == d["counts"]
    assert sum(r["own_counts"]["samples"] for r in rows) == 4 * batch and all(r["own_counts"]["samples"] == 2 * batch for r in rows)
    res1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(2 * batch), "--no-build"] + common,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res1.returncode == 0, res1.stderr[-2000:]
    d1 = json.loads([l for l in res1.stdout.splitlines() if l.lstrip().startswith("{")][0])
    assert d1["counts"] == d["counts"] and d["counts"]["flagged"] > 0


def test_roofline_counts_are_fingerprinted_and_the_fraction_is_a_fraction():
    """profiles/traffic.json carries the fingerprint of the kernel sources its PMC counts were measured on; bench.pmc_entry hands a
    count out only while the tree still hashes to it (else None + the reason).  With the counts of the current tree and the launch
    times the committed bench record of this round reports, the VALU-issue fraction is <= 1 (it is a fraction of something) and the
    streaming-model figure is labelled an effective bandwidth."""
    import bench
    from feedback_gnn_amd import _lib
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    # every BASELINE config's dominant kernel has an entry (round 4): no driver-runnable line prints a null fraction for want of one
    for key in ("bp4_ghp882_it32_B256", "bp4_ghp882_it64_B65536", "gnn_ghp882_B65536", "bp4_ghp1270_it64_B32768", "gnn_ghp1270_B32768",
                "gnnbp4_ghp1270_it10_B16384"):
        assert key in tj, key
    for kind, key in (("bp4", "bp4_ghp882_it64_B65536"), ("gnn", "gnn_ghp882_B65536"), ("bp4", "bp4_ghp1270_it64_B32768"),
                      ("gnn", "gnn_ghp1270_B32768"), ("gnnbp4", "gnnbp4_ghp1270_it10_B16384")):
        assert len(tj[key]["csrc_sha256"]) == 64 and len(tj[key]["lib_sha256"]) == 64
        ent, why = bench.pmc_entry(kind, key)
        if tj[key]["csrc_sha256"] == _lib.source_fingerprint(kind):
            assert ent is not None and ent["valu_wave_insts_per_launch"] > 0 and "kernel sources unchanged" in why
        else:
            assert ent is None and "other kernel sources" in why
    ent, why = bench.pmc_entry("bp4", "bp4_ghp882_it64_B123")
    assert ent is None and "no entry" in why
    # the fraction the counts give at the measured launch time of the same profile run: a fraction
    e = tj["bp4_ghp882_it64_B65536"]
    frac = e["valu_wave_insts_per_launch"] / (e["avg_ms_under_pmc"] * 1e-3) / 1e9 / bench.VALU_PEAK_GINST
    assert 0.5 < frac <= 1.0, frac
    assert e["hbm_bytes_per_launch"] / (e["avg_ms_under_pmc"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS < 0.05


def test_sharding_report_accepts_the_partition_and_names_every_way_it_can_fail():
    """bench.sharding_report (the `dist.sharding` object of a multi-GPU line; rank 0 exits 8 when it is not ok): rank r of `world`
    decodes block r of every `world` consecutive blocks of B samples (Sandwich_BP_GNN_Evaluation_Model._take_samples) — W warm-up
    steps, then K timed ones.  Eight ranks as the first SCALE run will have them; then a gap, an overlap, a rank off by one block, a
    short batch, a missing rank and counters that do not add up."""
    import bench
    W, K, world, B = 5, 20, 8, 65536
    ranges = [[[(W + k) * world * B + r * B, (W + k) * world * B + (r + 1) * B] for k in range(K)] for r in range(world)]
    rows = [[r, ranges[r][0][0], ranges[r][-1][1], 3 + r, 2 + r, K * B] for r in range(world)]
    total = [sum(x[3] for x in rows), sum(x[4] for x in rows), world * K * B]
    rep, ok = bench.sharding_report(ranges, rows, W, K, world, B, total)
    assert ok and rep["ranges_tile_the_region_without_overlap"] and rep["sum_of_rank_counts_equals_all_reduced"]
    assert rep["timed_region_samples"] == [W * world * B, (W + K) * world * B] and rep["batches"] == world * K
    assert rep["sum_of_rank_counts"] == {"flagged": total[0], "block_errors": total[1], "samples": total[2]}
    import copy
    def broken(edit, counts=total, rws=rows):
        rg = copy.deepcopy(ranges)
        edit(rg)
        return bench.sharding_report(rg, rws, W, K, world, B, counts)
    rep, ok = broken(lambda rg: rg[3].__setitem__(7, [rg[3][7][0] + B, rg[3][7][1] + B]))        # rank 3 decodes rank 4's block once: overlap + gap
    assert not ok and not rep["ranges_tile_the_region_without_overlap"] and rep["sum_of_rank_counts_equals_all_reduced"]
    rep, ok = broken(lambda rg: [r.__setitem__(0, [r[0][0] - world * B, r[0][1] - world * B]) for r in rg])  # everybody starts one step early
    assert not ok and not rep["ranges_tile_the_region_without_overlap"]
    rep, ok = broken(lambda rg: rg[0].__setitem__(0, [rg[0][0][0], rg[0][0][1] - 1]))              # a short batch
    assert not ok
    rep, ok = broken(lambda rg: rg.pop())                                                          # a rank is missing
    assert not ok and rep["batches"] == (world - 1) * K
    rep, ok = broken(lambda rg: rg.__setitem__(1, copy.deepcopy(rg[0])))                          # two ranks on the same samples
    assert not ok and not rep["ranges_tile_the_region_without_overlap"]
    rep, ok = broken(lambda rg: None, counts=[total[0] + 1, total[1], total[2]])                  # a counter lost or double-counted in the reduction
    assert not ok and rep["ranges_tile_the_region_without_overlap"] and not rep["sum_of_rank_counts_equals_all_reduced"]
    # one process is the world-1 case of the same rule
    rep, ok = bench.sharding_report([[[2048, 4096], [4096, 6144]]], [[0, 2048, 6144, 1, 1, 4096]], 1, 2, 1, 2048, [1, 1, 4096])
    assert ok and rep["batches"] == 2


def test_gnn_flops_are_the_survey_figures():
    """13.4 MFLOP ([[882,24]]) / 19.3 MFLOP ([[1270,28]]) per feedback-GNN pass and codeword."""
    import bench
    assert bench.gnn_flops_per_codeword(882, 5292) == 13406400
    assert bench.gnn_flops_per_codeword(1270, 7620) == 19304000
    # factored association: per qubit and side 3*40 + 3*40 + 40*20 multiply-adds, then the unchanged embed MLP
    assert bench.gnn_flops_per_codeword_factored(882, 5292) == 882 * 2 * (2 * (120 + 120 + 800) + 43 * 40 + 120) == 6914880


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
def test_bench_three_ranks_equal_one_process_over_the_same_global_samples():
    """More ranks than two, and not a power of two: `python bench.py --gpus 3` (three ranks sharing the test box's one GPU over gloo —
    the box allows at most six GPU processes of ours at a time and this pytest process is one of them, so the count stays well below;
    the EIGHT-rank rendezvous / sharding / reduction is covered without a GPU by tests/test_launch.py::test_eight_ranks_*) must report
    n_gpus 3, three per-rank step times, a global batch of 3 B, and counters equal to ONE process decoding the same 3 B global
    samples per step."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FGNN_BENCH_BACKEND"] = "gloo"
    Bq, N = 1024, 3
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--steps", "2", "--warmup", "1",
                          "--batch", str(Bq), "--p", "0.1", "--cpu-sample", "0", "--no-extras"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == N and d["config"]["global_batch"] == N * Bq and d["counts"]["samples"] == 2 * N * Bq
    assert len(d["per_rank_ms"]) == N and abs(max(d["per_rank_ms"]) - d["ms_per_step"]) < 1e-6 * d["ms_per_step"]
    assert abs(d["value"] - 2 * N * Bq / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # the CPU legs belong to the single-GPU line
    # the pre-flight map (n882.py:9-25: one process per GPU id): world size and backend as torch.distributed reports them, one row per
    # rank gathered by all-gather — here three ranks share cuda:0 over gloo, which the line must SAY (under RCCL that is exit code 7)
    ds = d["dist"]
    assert ds["world_size"] == N and ds["backend"] == "gloo" and [r["rank"] for r in ds["ranks"]] == [0, 1, 2]
    assert [r["local_rank"] for r in ds["ranks"]] == [0, 1, 2] and {r["device_index"] for r in ds["ranks"]} == {0}
    assert len({r["pid"] for r in ds["ranks"]}) == N and all("MI3" in r["device_name"] or r["device_name"] for r in ds["ranks"])
    assert ds["one_distinct_device_per_rank"] is False
    # every rank's global sample range of the timed region and its own counters; rank 0 has checked that the per-step ranges tile
    # [W * world * B, (W + K) * world * B) without overlap and that the ranks' counters add up to the all-reduced ones (else: exit 8)
    assert [r["timed_samples"] for r in ds["ranks"]] == [[(N + k) * Bq, (2 * N + k + 1) * Bq] for k in range(N)]
    sh = ds["sharding"]
    assert sh["timed_region_samples"] == [N * Bq, 3 * N * Bq] and sh["batches"] == 2 * N and sh["ranges_tile_the_region_without_overlap"] is True
    assert sh["sum_of_rank_counts_equals_all_reduced"] is True and sh["sum_of_rank_counts"] == d["counts"]
    assert all(r["own_counts"]["samples"] == 2 * Bq for r in ds["ranks"]) and len({r["own_counts"]["flagged"] for r in ds["ranks"]}) > 1
    res1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", str(N * Bq),
                           "--p", "0.1", "--cpu-sample", "0", "--no-extras", "--no-build"], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res1.returncode == 0, res1.stderr[-2000:]
    d1 = json.loads([l for l in res1.stdout.splitlines() if l.strip()][0])
    assert d1["counts"] == d["counts"] and d["counts"]["block_errors"] > 0


@pytest.mark.gpu
def test_bench_rank_failure_is_a_nonzero_exit_not_a_hang():
    """A rank that cannot get its GPU (nccl backend, more ranks than devices, started by a launcher so that bench.py's own
    up-front refusal does not apply) must leave with a non-zero code from its own fresh process, and the job must end."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with one GPU")
    from feedback_gnn_amd.launch import spawn_ranks
    env_keep = {k: os.environ.get(k) for k in ("FGNN_BENCH_BACKEND",)}
    os.environ.pop("FGNN_BENCH_BACKEND", None)
    try:
        codes, out0 = spawn_ranks(os.path.join(ROOT, "bench.py"), ["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "256",
                                                                  "--cpu-sample", "0", "--no-extras", "--no-build"], 2, capture_rank0=True)
    finally:
        for k, v in env_keep.items():
            if v is not None:
                os.environ[k] = v
    assert codes[1] not in (0, None) and codes[0] != 0, codes
    assert not [l for l in (out0 or "").splitlines() if l.lstrip().startswith("{")]  # no bench line from a broken job


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
                           "--p", "0.1", "--cpu-sample", "0", "--no-extras", "--no-build", "--no-literal"], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res1.returncode == 0, res1.stderr[-2000:]
    d1 = json.loads([l for l in res1.stdout.splitlines() if l.strip()][0])
    assert d1["counts"] == d["counts"] and d["counts"]["block_errors"] > 0
    # --no-literal / --no-other-forms (profiled runs): no other-forms region, no forms_agreement decode — and the line says so with a null
    assert d1["reassociated_forms"] is None and d1["literal_forms"]["value"] == d1["value"] and d["literal_forms"]["value"] > 0


@pytest.mark.gpu
def test_bench_under_the_drivers_torchrun_line():
    """The driver's own launch line for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` — with two ranks sharing the test box's GPU over gloo: RANK / LOCAL_RANK / WORLD_SIZE come
    from the launcher, rank 0 prints the one line, and N = 1 under the same launcher runs without a process group."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FGNN_BENCH_BACKEND"] = "gloo"
    for n in (1, 2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                              "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2",
                              "--warmup", "1", "--batch", "2048", "--p", "0.1", "--cpu-sample", "0", "--no-extras", "--no-build"],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [l for l in res.stdout.splitlines() if l.lstrip().startswith("{")]
        assert len(lines) == 1, res.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and len(d["per_rank_ms"]) == n and d["config"]["global_batch"] == n * 2048
        assert d["counts"]["samples"] == 2 * n * 2048 and d["scaling"] == "weak"
