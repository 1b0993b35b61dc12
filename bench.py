#!/usr/bin/env python3
"""bench.py — decoded codewords/s of the BP4 + feedback-GNN sandwich (and of GNN_BP4) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c4|c5]   (N > 1: starts N ranks itself, one per GPU, before touching a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--config` picks the BASELINE.json configuration (per-GPU shard of it; default c3):
  c3  configs[2]  [[882,24]]  sandwich BP4-64 + feedback GNN + BP4-16, 65 536 codewords per GPU and step (n882.py:45-66) — the headline
  c4  configs[3]  [[1270,28]] sandwich BP4-64 + feedback GNN + BP4-64, 262 144 / 8 = 32 768 codewords per GPU and step (n1270.py:37,57)
  c5  configs[4]  [[1270,28]] GNN_BP4, 10 iterations, seeded weights, 131 072 / 8 = 16 384 codewords per GPU and step (gnn.py:383-423)
  c1  configs[0]  [[882,24]]  BP4-32 alone, 256 codewords, p = 0.05, AS THE REFERENCE CONSTRUCTS IT: cn_type='boxplus', normalization_factor
                  0.625 (decoding_q.py:18-22); `--cn-type boxplus-phi` is the QLDPC.ipynb cell 11 helper's variant.  A latency figure.
  c2  configs[1]  [[882,24]]  BP4-64 alone, 65 536 codewords (the first launch of c3)
  qldpc_882 / qldpc_1270   plain BP4-64 as examples/QLDPC.ipynb cell 12 runs it (boxplus-phi, factor 0.8, p0 = 0.3, batch 10 000, p = 0.01)
  osd_bp4_minsum   BP4 min-sum, 120 iterations, factor 0.8, 50 000 samples at p = 0.09 (examples/OSD.ipynb cell 6, "bp Elapsed time")
  n1270_5r / n1270_coarse   n1270.ipynb cells 4 and 9 (five rounds at p = 0.08; the coarse GNN at p = 0.02)
  n882_3r / n882_5r / n1270_3r   the workloads of the reference's only published timings (one RTX 4090, TF-XLA):
                  (64, G, 16) x 3 or 5 feedback rounds, batch_size 5 000, p = 0.05 / 0.07 (n882.py:13,39,56-66, n1270.py:57-70); timed on one
                  stream (`value`) and with consecutive batches alternating between two HIP streams (`two_streams`) in the same run.
`--cn-type` / `--factor` set the check-node rule and normalization factor of every decoder (default: the configuration's).

One "step" = one Monte-Carlo batch through the whole hot path, everything on device: Philox depolarizing noise -> syndromes -> decoder
(c3 / c4: BP4 -> flag update -> feedback GNN (trained weights) -> BP4 -> masked merge; c5: GNN_BP4) -> residual check -> counters, at
p = 0.01, p0 = 0.05.  Every sample goes through every stage and every iteration with NO data-dependent shortcut (no compaction, no
early exit, the exact saturation shortcut of the product default switched OFF): the operation count per codeword is a constant, like
the reference's fixed dataflow; the shortcut / compaction variants (bit-identical outputs) are reported under `extras`.  Batches are
sharded over ranks by global sample index with no data-path collective; the three counters are all-reduced once at the end ("weak"
scaling: per-GPU batch fixed).

WHICH ARITHMETIC IS TIMED.  `value` is measured on the library's default operation sequence, which since round 6 is the reference's
formulas term by term: one reduce_logsumexp per edge in the qubit update (decoding_q.py:254-273) and one 40 -> 20 Dense per edge in the
feedback GNN (feedback_gnn.py:175-184 / gnn.py:573-610) — `config` names them (`gnn_association: literal`, `bp4_qubit_update_lse: per edge
(literal)`), and `literal_forms` repeats the figure under the key earlier rounds used.  The two OPT-IN re-associations (include/fgnn.h
options 4 and 5: the log-sum-exp term shared per qubit side, the GNNs' Dense layers factored; the same real-number functions, statistically
the same decoder, NOT the reference's operation sequence) are timed in the same run on single-GPU runs and reported under
`extras.reassociated_forms` {value, ms_per_step, forms_agreement} next to the opt-in hardware-transcendental variant; `forms_agreement`
decodes the first timed batch under both and reports, per sample, how many final decisions differ and how far the marginals are apart
(north-star bar: decisions identical, LLRs within 1e-4 — which the re-associated forms do NOT meet on every sample: DESIGN.md §3).
FGNN_BENCH_BP4_LSE=shared / FGNN_BENCH_GNN_ORDER=factored make the re-associated forms the timed headline (A/B and profiling runs only;
`value_is` then says so).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against the bound it is actually subject to.  c3 / c4: the
64-iteration BP4 launch, bound by VALU instruction issue — it keeps every message in LDS for all iterations, so HBM sees only its
inputs and outputs (0.4 % of peak) and the MFMA pipe nothing; `achieved` = VALU wave-instructions per launch / the launch's average
duration, measured live with HIP events recorded on the launch stream around every launch of the timed region (fgnn_profile_*);
`peak` = 1 024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; `frac` = achieved / peak <= 1.  That is issue UTILISATION of the
library's own instruction stream; next to it stands the algorithmic reading: `transcendental_evals_per_launch` (every exp and log the
fixed dataflow evaluates: 20 + 32 per qubit-iteration in the literal form) / launch time against the chip's quarter-rate hardware
transcendental rate (1 024 SIMDs x 64 lanes x 2.4 GHz / 8 = 1.97e13 /s) = `frac_of_hw_transcendental_rate` — what a bit-inexact
v_exp_f32 / v_log_f32 implementation is bounded by.  c5: the GNN_BP4 launch against the f32 MFMA (= f32 vector) peak in the reference's
FLOPs.  The instruction counts / HBM bytes per launch are properties of the compiled kernel at a shape (fixed dataflow:
no data-dependent branch), taken from the rocprofv3 --pmc passes summarised in profiles/traffic.json, and quoted only while the kernel
sources still hash to what that profile was measured on (otherwise `traffic` / `frac` are null with a reason; `--require-roofline`
turns a null `roofline.frac` into a non-zero exit).  `effective_bandwidth_frac` is the SURVEY.md §8(d) contract figure (the reference's
streaming dataflow: 16E+12n+4m bytes per codeword-iteration + 4E+24n+2n+4m epilogue) / launch time / 8 TB/s — an EFFECTIVE bandwidth
that exceeds 1 because the messages never travel; `hbm_frac` = measured HBM bytes (`traffic`, rocprofv3 FETCH_SIZE x2 + WRITE_SIZE) /
launch time / 8 TB/s.  `roofline.gnn` prices the feedback-GNN launch the same way.

`dist` is the multi-GPU pre-flight: world size and backend as torch.distributed reports them and one row per rank (LOCAL_RANK -> device
index, name, uuid, pid) gathered by all-gather; a rank that is not on cuda:LOCAL_RANK under RCCL exits 6, ranks sharing a device exit 7.

`cpu_baseline` times the oracle (a C port of the reference arithmetic, OpenMP over codewords) on the host cores on a bounded sample of
the same workload; `cpu_baseline_tf_like` times an op-for-op restatement of how the reference executes on a host (batch-minor [E,B]
tensors, one framework op at a time: oracle/torch_cpu_baseline.py; c5: the batched-matmul NumPy restatement oracle/numpy_ref.py) —
TensorFlow itself can run on neither box.  Both CPU legs run BEFORE the GPU is touched, in a child interpreter whose thread pools end with
it; a 0.25 s settle phase (an elementwise torch loop) precedes the W warm-up steps; the GPU's decisions on the sampled codewords are then checked
against the oracle's bit for bit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# same guide: 256 CUs x 4 SIMD-32, max clock 2.4 GHz, a wave64 VALU instruction issues over 2 cycles
VALU_PEAK_GINST = 1024 * 2.4e9 / 2 / 1e9  # 1 228.8 G wave-instructions/s
# hardware transcendentals (v_exp_f32 / v_log_f32) issue at a quarter of that: 8 cycles per wave64 = 64 results
HW_TRANSCENDENTAL_PEAK = 1024 * 64 * 2.4e9 / 8  # 1.966e13 evaluations/s
GNN_PEAK_TFLOPS = 157.3  # same guide: f32 MFMA (= f32 vector) peak

CONFIGS = {
    # configs[0] as the reference CONSTRUCTS it: QLDPCBPDecoder's defaults cn_type='boxplus', normalization_factor=0.625 (decoding_q.py:18-22);
    # `--cn-type boxplus-phi` is the QLDPC.ipynb cell 11 helper's variant of the same case
    "c1": dict(code="ghp882", iters="32", batch=256, p=0.05, cn_type="boxplus", factor=0.625,
               baseline="configs[0]: the reference's own CPU-runnable case"),
    "c2": dict(code="ghp882", iters="64", batch=65536, baseline="configs[1]: BP4 alone"),
    "c3": dict(code="ghp882", iters="64,16", batch=65536, baseline="configs[2]"),
    "c4": dict(code="ghp1270", iters="64,64", batch=32768, baseline="configs[3], per-GPU shard 262 144 / 8"),
    "c5": dict(code="ghp1270", iters="10", batch=16384, baseline="configs[4], per-GPU shard 131 072 / 8"),
    # the workloads the reference's only published timings were taken on (RTX 4090, TF-XLA, batch_size 5 000)
    "n882_3r": dict(code="ghp882", iters="64,16,16,16", batch=5000, p=0.05,
                    baseline="none — the reference's published workload examples/n882.ipynb cell 2 (n882.py:45-66 with nG = 3), 10.9 k cw/s on an RTX 4090"),
    "n882_5r": dict(code="ghp882", iters="64,16,16,16,16,16", batch=5000, p=0.05,
                    baseline="none — the reference's published workload n882.py:13,45-66 (nG = 5; examples/n882.ipynb cell 3), 7.50 k cw/s on an RTX 4090"),
    "n1270_3r": dict(code="ghp1270", iters="64,16,16,16", batch=5000, p=0.07,
                     baseline="none — the reference's published workload examples/n1270.ipynb cell 2 (n1270.py:57-70 with nG = 3), 6.39 k cw/s on an RTX 4090"),
    "n1270_5r": dict(code="ghp1270", iters="64,16,16,16,16,16", batch=5000, p=0.08,
                     baseline="none — the reference's published workload examples/n1270.ipynb cell 4 (nG = 5), 4.46 k cw/s on an RTX 4090"),
    "n1270_coarse": dict(code="ghp1270", iters="64,16", batch=5000, p=0.02, weights="feedback_GNN_n1270_k28_wt_10_60_iter_16_16.npz",
                         baseline="none — the reference's published workload examples/n1270.ipynb cell 9: (64, G_coarse, 16), 10.9 k cw/s on an RTX 4090"),
    # BP4 min-sum as examples/OSD.ipynb cell 6 times it ("bp Elapsed time": noise + syndromes + 120 min-sum iterations on 50 000 samples)
    "osd_bp4_minsum": dict(code="ghp882", iters="120", batch=50000, p=0.09, cn_type="minsum", factor=0.8, p0=0.09,
                           baseline="none — the reference's published timing examples/OSD.ipynb cell 6, 50 000 / 3.96 s = 12.6 k cw/s on an RTX 4090"),
    # plain BP4 as examples/QLDPC.ipynb cell 12 runs it (helper `define_code`: 64 iterations, boxplus-phi, factor 0.8, p0 = 0.3, batch 10 000)
    "qldpc_882": dict(code="ghp882", iters="64", batch=10000, p=0.01, cn_type="boxplus-phi", factor=0.8, p0=0.3,
                      baseline="none — the reference's published workload examples/QLDPC.ipynb cell 12, table GHP_n882_k24, row p = 0.01: 29.7 k cw/s on an RTX 4090"),
    "qldpc_1270": dict(code="ghp1270", iters="64", batch=10000, p=0.01, cn_type="boxplus-phi", factor=0.8, p0=0.3,
                       baseline="none — the reference's published workload examples/QLDPC.ipynb cell 12, table GHP_n1270_k28, row p = 0.01: 17.3 k cw/s on an RTX 4090"),
}
PUBLISHED_CONFIGS = ("n882_3r", "n882_5r", "n1270_3r", "n1270_5r", "n1270_coarse", "osd_bp4_minsum", "qldpc_882", "qldpc_1270")
CN_TYPES = ("boxplus", "boxplus-phi", "minsum")
PROF_TAG_GNN, PROF_TAG_GNNBP4 = -1, -2  # fgnn_profile_read tags (include/fgnn.h)


def algorithmic_bytes_per_codeword(n, m, E, iters):
    """f32 message-streaming model of the reference's dataflow."""
    per_iter = 16 * E + 12 * n + 4 * m
    epilogue = 4 * E + 12 * n + 12 * n + 2 * n + 4 * m
    return per_iter * iters + epilogue


def bp4_transcendentals_per_codeword(n, m, E, iters, shared_lse=True, cn_type="boxplus-phi"):
    """(exp, log) evaluations of one fixed-dataflow BP4 decode (boxplus-phi) per codeword.  Per iteration: the qubit update — two
    softplus per qubit (exp + log1p each, decoding_q.py:265,270) and one log-sum-exp (exp + log) per edge (:266,:271), or per qubit and
    side in the shared form — and the check update — two phi per edge (decoding_q.py:405-429), a phi being one exp, one log1p and one
    log (:372-373).  Epilogue (cal_logit, :455-471): two softplus and two log-sum-exp per qubit, one phi per row entry and one per row
    of the two soft-syndrome row sets (stage-one: hz and hx, E entries, m rows).  [[882,24]], shared form: 16 exp + 28 log per
    qubit-iteration."""
    lse = 2 * n if shared_lse else E
    # check update: 'boxplus-phi' two phi per edge; 'boxplus' one tanh (a rational: no exp / log) and one atanh (one log1p) per edge
    # (decoding_q.py:313-363); 'minsum' none (:539-644)
    cn_exp, cn_log = {"boxplus-phi": (2 * E, 4 * E), "boxplus": (0, E), "minsum": (0, 0)}[cn_type]
    exp_it, log_it = 2 * n + lse + cn_exp, 2 * n + lse + cn_log
    exp_ep, log_ep = 4 * n + (E + m), 4 * n + 2 * (E + m)
    return exp_it * iters + exp_ep, log_it * iters + log_ep


def gnn_flops_per_codeword(n, E):
    """E*2*(4*40+40*20) + n*2*(43*40+40*3) = 13.4 MFLOP ([[882,24]]) / 19.3 MFLOP ([[1270,28]])."""
    return E * 2 * (4 * 40 + 40 * 20) + n * 2 * (43 * 40 + 40 * 3)


def gnn_flops_per_codeword_factored(n, E):
    """What the factored association (FGNN_OPT_GNN_FACTORED) executes: per qubit and side 3*40 (X/Y/Z part of the first Dense)
    + deg*40 (one fma per edge and hidden unit) + 40*20 (ONE last Dense) multiply-adds, then the same embed MLP."""
    return 2 * (2 * n * (3 * 40 + 40 * 20) + E * 40) + n * 2 * (43 * 40 + 40 * 3)


def gnnbp4_flops_per_codeword(n, m, E, iters, D=20, H=40):
    """SURVEY.md §8(d), GNN_BP4 with embed 20 / hidden 40 / message width 20: per iteration the qubit side E*2*(2D*H + H*D) +
    n*2*(3D*H + H*D) and the check side E*2*(2D*H + H*D) + m*2*((2D+1)*H + H*D) = 87.5 MFLOP on [[1270,28]]; `iters` qubit updates
    and `iters` - 1 check updates (gnn.py:414-415) = 0.83 GFLOP per codeword for 10 iterations.  (The kernel also runs the check
    update that precedes the first iteration, gnn.py:400-401: the contract figure under-counts what is executed by one check side.)"""
    vn = E * 2 * (2 * D * H + H * D) + n * 2 * (3 * D * H + H * D)
    cn = E * 2 * (2 * D * H + H * D) + m * 2 * ((2 * D + 1) * H + H * D)
    return iters * vn + (iters - 1) * cn


def gnnbp4_flops_per_codeword_factored(n, m, E, iters, D=20, H=40):
    """What the factored association executes: per receiving node and side D*H (own half of the first Dense, once) + deg*D*H (the
    neighbour halves) + H*D (ONE last Dense) multiply-adds, then the unchanged embed MLPs; `iters` check updates (see above)."""
    vn = 2 * (2 * n * (D * H + H * D) + E * D * H) + n * 2 * (3 * D * H + H * D)
    cn = 2 * (m * (D * H + H * D) + E * D * H) + m * 2 * ((2 * D + 1) * H + H * D)
    return iters * vn + iters * cn


def pmc_key(kind, code_name, B, iters=None, cn_type="boxplus-phi", reassociated=False):
    """Key of profiles/traffic.json: kernel kind, code, launch shape, check-node rule when it is not 'boxplus-phi', and the FORM the PMC
    pass ran: no suffix = the library default (the reference's formulas term by term), `_shared` / `_factored` = the opt-in re-association
    of that kernel (tools/refresh_traffic.sh configurations c3r / c4r)."""
    key = f"{kind}_{code_name}" + (f"_it{iters}" if iters is not None else "") + f"_B{B}"
    if kind == "bp4" and cn_type != "boxplus-phi":
        key += f"_{cn_type}"
    if reassociated:
        key += "_shared" if kind == "bp4" else "_factored"
    return key


def pmc_entry(kind, key):
    """The offline rocprofv3 counts of one kernel at one shape from profiles/traffic.json, or (None, reason) when there is none
    or when the kernel's sources have changed since they were taken (a count of some other build is not a measurement of this one)."""
    from feedback_gnn_amd import _lib
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        ent = json.load(open(tpath)).get(key)
    except Exception as e:
        return None, f"profiles/traffic.json unreadable: {e}"
    if not ent:
        return None, f"profiles/traffic.json has no entry {key}"
    fp = _lib.source_fingerprint(kind)
    if ent.get("csrc_sha256") != fp:
        return None, (f"profiles/traffic.json[{key}] was measured on other kernel sources (csrc_sha256 {str(ent.get('csrc_sha256'))[:12]} "
                      f"!= {fp[:12]} of this tree): re-run tools/refresh_traffic.sh")
    ent = dict(ent)
    ent["library_is_the_profiled_binary"] = ent.get("lib_sha256") == _lib.library_sha256()
    return ent, (f"profiles/traffic.json (offline rocprofv3 --pmc passes, {ent.get('taken_at')}, kernel sources unchanged since: "
                 f"csrc_sha256 {fp[:12]}); counts are per launch of this shape, not measured in this run")


def sharding_report(step_ranges, rank_rows, W, K, world, B, reduced_counts):
    """The proof a multi-GPU line carries that the batch was sharded as DESIGN.md §6 says (host-only, unit-tested on CPU):
    `step_ranges[r]` = the K half-open global sample ranges rank r decoded in the timed region, `rank_rows[r]` = [rank, first, last,
    flagged, block_errors, samples] of that rank BEFORE the all-reduce, `reduced_counts` = the three all-reduced counters.  The world * K
    ranges must tile [W * world * B, (W + K) * world * B) without gap or overlap, each of length B, and the ranks' own counters must
    add up to the reduced ones.  Returns (the `dist.sharding` object, ok)."""
    flat = sorted((int(a), int(b)) for per_rank in step_ranges for a, b in per_rank)
    lo, hi = W * world * B, (W + K) * world * B
    tiles = (len(flat) == world * K and flat[0][0] == lo and flat[-1][1] == hi
             and all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1)) and all(b - a == B for a, b in flat))
    sums = [sum(int(r[3 + j]) for r in rank_rows) for j in range(3)]
    adds_up = sums == [int(v) for v in reduced_counts]
    rep = {"timed_region_samples": [lo, hi], "batches": len(flat), "ranges_tile_the_region_without_overlap": bool(tiles),
           "sum_of_rank_counts": {"flagged": sums[0], "block_errors": sums[1], "samples": sums[2]},
           "sum_of_rank_counts_equals_all_reduced": bool(adds_up)}
    return rep, bool(tiles and adds_up)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration: c3 = configs[2] (headline), c4 = configs[3] shard, c5 = configs[4] shard (GNN_BP4); "
                         "c1 = configs[0] (BP4-32, 256 codewords, p = 0.05, 'boxplus' 0.625), c2 = configs[1] (BP4-64 alone); n882_3r / n882_5r / "
                         "n1270_3r = the workloads of the reference's published timings (batch 5 000)")
    ap.add_argument("--batch", type=int, default=None, help="codewords per GPU per step (default: the configuration's)")
    ap.add_argument("--p", type=float, default=None, help="depolarizing probability (default: the configuration's, 0.01 unless it names one)")
    ap.add_argument("--code", default=None, choices=["ghp882", "ghp1270"], help="default: the configuration's")
    ap.add_argument("--iters", default=None, help="BP iterations per stage (c5: GNN_BP4 iterations); default: the configuration's")
    ap.add_argument("--cn-type", default=None, choices=CN_TYPES,
                    help="check-node rule of every decoder of the sandwich (decoding_q.py:18: the class default is 'boxplus'; the reference's "
                         "scripts and notebooks construct 'boxplus-phi', n882.py:61, QLDPC.ipynb cell 11); default: the configuration's "
                         "(c1: 'boxplus', every other: 'boxplus-phi')")
    ap.add_argument("--factor", type=float, default=None,
                    help="normalization_factor of every decoder (decoding_q.py:22: class default 0.625; n882.py:58-59: 1.0); default: the "
                         "configuration's (c1: 0.625, every other: 1.0)")
    ap.add_argument("--p0", type=float, default=None,
                    help="the p0 the channel LLR log(3(1-p0)/p0) of the first decoder is formed from (feedback_gnn.py:265,311-312: 0.05; "
                         "QLDPC.ipynb cell 12: 0.3); default: the configuration's")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams consecutive batches alternate between (Sandwich_BP_GNN_Evaluation_Model(streams=)); the published-workload "
                         "configurations time 1 and 2 in the same run")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="codewords for the CPU baseline (0 = skip, -1 = sized from a probe to ~12 s of CPU work)")
    ap.add_argument("--cpu-baseline", default="both", choices=["both", "port", "torch", "none"],
                    help="which CPU legs to time: the C port of the reference arithmetic, the TF-CPU-shaped restatement, or both")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-build", action="store_true",
                    help="do not run make: only check that the libraries exist (profiled runs, child ranks)")
    ap.add_argument("--no-literal", "--no-other-forms", dest="no_literal", action="store_true",
                    help="skip the timing of the other forms (the opt-in re-associations on a default run) and the forms_agreement decode "
                         "(profiled runs: only the launches of the warm-up and of the timed region reach the trace)")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed GPU work (an elementwise torch loop, none of this library's kernels) before the warm-up steps, in ms of wall time")
    ap.add_argument("--cpu-legs-to", default=None, help=argparse.SUPPRESS)  # internal: run the CPU legs only and pickle them to this path
    ap.add_argument("--require-roofline", action="store_true",
                    help="exit non-zero (after printing the line) when roofline.frac is null: no offline PMC counts for this shape, "
                         "or counts measured on other kernel sources")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    args.batch = cfg["batch"] if args.batch is None else args.batch
    args.code = cfg["code"] if args.code is None else args.code
    args.iters = cfg["iters"] if args.iters is None else args.iters
    args.p = cfg.get("p", 0.01) if args.p is None else args.p
    args.cn_type = cfg.get("cn_type", "boxplus-phi") if args.cn_type is None else args.cn_type
    args.factor = cfg.get("factor", 1.0) if args.factor is None else args.factor
    args.p0 = cfg.get("p0", 0.05) if args.p0 is None else args.p0
    if args.streams < 1:
        ap.error("--streams must be >= 1")
    return args


def build_once(no_build):
    """Compile (or, with --no-build, just locate) the native libraries BEFORE this process makes any GPU call: make and hipcc
    run as children of a process that has not initialised the GPU.  Ranks started by a launcher serialise on a file lock;
    all but the first find everything up to date."""
    import fcntl
    import __graft_entry__ as entry
    from feedback_gnn_amd import _lib
    if no_build:
        if not os.path.exists(_lib.LIB_PATH):
            raise SystemExit(f"--no-build: {_lib.LIB_PATH} is missing")
        return
    with open(os.path.join(ROOT, ".bench_build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            entry.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks (one process per GPU, the way the reference
    pins one process per GPU id, /root/reference n882.py:9,15-21) and relay rank 0's JSON line.  This parent never touches a
    GPU: devices are counted and the build is run by child interpreters, and the ranks are fresh interpreters."""
    import subprocess
    from feedback_gnn_amd.launch import spawn_ranks, visible_gpus  # host-only module: no GPU library is loaded in this parent
    backend = os.environ.get("FGNN_BENCH_BACKEND", "nccl")
    ndev = visible_gpus() if backend == "nccl" else 1  # counted in a child interpreter
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible "
                         "(FGNN_BENCH_BACKEND=gloo lets ranks share a device for a self-test)")
    if not args.no_build:  # in a child interpreter: this parent loads no GPU library at all
        rc = subprocess.call([sys.executable, "-c", "import __graft_entry__ as e; e.build()"], cwd=ROOT)
        if rc != 0:
            raise SystemExit("bench.py: build failed")
    child_argv = [a for a in argv if a != "--no-build"] + ["--no-build"]
    codes, out0 = spawn_ranks(__file__, child_argv, args.gpus, capture_rank0=True)
    # rank 0's stdout carries the ONE JSON line; anything else a library printed there (gloo's rendezvous banner) goes to stderr
    for line in out0.splitlines():
        is_json = line.lstrip().startswith("{") and line.rstrip().endswith("}")
        (sys.stdout if is_json else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit(f"bench.py: ranks failed (rank, exit code): {bad}")


def make_code(name):
    import numpy as np
    import feedback_gnn_amd as F
    if name == "ghp882":
        return (F.create_QC_GHP_codes(63, F.create_cyclic_permuting_matrix(7, [27, 54, 0]), [0, 1, 6]),
                "feedback_GNN_n882_k24_wt_4_60_iter_64_16_mixed.npz")
    return (F.create_QC_GHP_codes(127, np.array([[0, -1, 51, 52, -1], [-1, 0, -1, 111, 20], [0, -1, 98, -1, 122],
                                                  [0, 80, -1, 119, -1], [-1, 0, 5, -1, 106]]), [0, 1, 7], name="GHP_n1270_k28"),
            "feedback_GNN_n1270_k28_wt_10_80_iter_64_16_mixed.npz")


def gnnbp4_seeded_weights(seed=0):
    """The 30 arrays of GNN_BP4(num_embed_dims=20, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, use_bias=True) as Keras
    would initialise them (glorot-uniform kernels, ones biases, gnn.py:47-50) from a seeded generator — the reference ships no trained
    GNN_BP4 weights; `_llr_inv_embed` gets a glorot kernel too (Keras: zeros, gnn.py:249) so that the marginals and
    decisions the run is checked on are not constants."""
    import numpy as np
    from feedback_gnn_amd.graph import GNNBP4_SHAPES
    rng = np.random.RandomState(seed)
    w = []
    for shp in GNNBP4_SHAPES:
        if len(shp) == 1:
            w.append(np.ones(shp, np.float32))
        else:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            w.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    return w


def llr_const(p0):
    import numpy as np
    p0 = np.float32(p0)
    return float(np.log(np.float32(3.0) * (np.float32(1.0) - p0) / p0, dtype=np.float32))  # feedback_gnn.py:311-312


def _cpu_model():
    try:
        return [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return "unknown"


def cpu_legs(args, code, wname, iters, seed, factored):
    """Both CPU baselines of the sandwich (c3 / c4), on host cores only — this runs BEFORE the process touches the GPU.  Returns the
    JSON objects and the oracle's decisions / flags on the sample (checked against the GPU's afterwards)."""
    import numpy as np
    from feedback_gnn_amd.weights_io import read_weight_list
    from oracle.oracle import OracleGraph, host_cpu_share, num_threads, set_num_threads
    out, check = {}, None
    # threads = the CPUs this process is really granted (affinity capped by the cgroup quota), unless OMP_NUM_THREADS says otherwise
    share = host_cpu_share()
    if "OMP_NUM_THREADS" not in os.environ:
        set_num_threads(share)
    L0 = llr_const(args.p0)
    w = read_weight_list(wname)
    cpu_model = _cpu_model()
    shared = os.environ.get("FGNN_BENCH_BP4_LSE", "literal") == "shared"
    # the checker restates the very forms the GPU run is configured with (both restatements exist in oracle/fgnn_oracle.c)
    og = OracleGraph(code, forms="reassociated" if (factored and shared) else "literal")
    og.set_gnn_order(factored)
    og.set_vn_shared_lse(shared)
    nl = len(iters)
    fcs, cts = [args.factor] * nl, [args.cn_type] * nl  # every decoder of the sandwich is constructed alike (n882.py:61-62)
    if args.cpu_baseline in ("both", "port"):
        S = args.cpu_sample
        if S < 0:  # probe: 2 codewords per thread, then size the sample for ~12 s
            probe = 2 * num_threads()
            ex, ez = og.pauli_noise(seed, args.p, 0, probe)
            sx, sz = og.syndrome(ex, ez)
            og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), L0, factors=fcs, cn_types=cts)  # also warms the thread pool
            t = time.perf_counter()
            og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), L0, factors=fcs, cn_types=cts)
            rate = probe / (time.perf_counter() - t)
            S = int(min(32768, max(512, 12.0 * rate)))
            S -= S % 64
        ex, ez = og.pauli_noise(seed, args.p, 0, 8)
        sx, sz = og.syndrome(ex, ez)
        og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), L0, factors=fcs, cn_types=cts)  # warm the thread pool
        t = time.perf_counter()
        ex, ez = og.pauli_noise(seed, args.p, 0, S)
        sx, sz = og.syndrome(ex, ez)
        o = og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), L0, factors=fcs, cn_types=cts)
        _, _, fl = og.residual(ex, ez, o["x_hat"], o["z_hat"])
        t_cpu = time.perf_counter() - t
        out["cpu_baseline"] = {"value": S / t_cpu, "unit": "codewords/s", "cores": num_threads(), "host_cpu_share": share,
                               "host_hw_threads": os.cpu_count(), "kind": "port",
                               "sample": f"{S} codewords of the same workload (same Philox samples 0..{S - 1}), "
                                         f"oracle/fgnn_oracle.c with OpenMP over codewords, {t_cpu:.1f} s on {cpu_model}",
                               "note": "the repo's own C port of the reference arithmetic (one codeword per thread, state in L1): a far "
                                       "better CPU program than the reference's TensorFlow graph, which can run on neither box "
                                       "(SURVEY.md §8c); see cpu_baseline_tf_like for the reference-shaped execution"}
        check = (S, o, fl)
    if args.cpu_baseline in ("both", "torch") and args.cn_type == "minsum":
        pass  # oracle/torch_cpu_baseline.py restates the 'boxplus-phi' and 'boxplus' rules only: no TF-shaped leg for min-sum
    elif args.cpu_baseline in ("both", "torch"):
        import torch
        from oracle import torch_cpu_baseline as T
        nthr = num_threads()  # the same thread count as the C port
        torch.set_num_threads(nthr)
        torch.set_flush_denormal(True)  # TensorFlow's CPU thread pools flush denormals too (switched back off after the timing)
        tg = T.Graph(code)
        # chunks of 512 codewords (one chunk = the batch the [E,B] tensors are built for) until ~8 s have passed: the wall time stays
        # bounded whatever the host is doing, and the rate is all codewords / all time
        chunk = 512 if args.cpu_sample < 0 else max(8, min(args.cpu_sample, 512))
        budget_s, max_chunks = (8.0, 8) if args.cpu_sample < 0 else (1e9, max(1, min(args.cpu_sample, 2048) // chunk))
        ex, ez = og.pauli_noise(seed, args.p, 0, 32)
        sx, sz = og.syndrome(ex, ez)
        T.sandwich_decode(tg, w, sx, sz, [2] * nl, L0, fcs, cts)  # warm-up
        St, t_t, xs, zs, sxs, szs, chunk_s = 0, 0.0, [], [], [], [], []
        while len(xs) < max_chunks and t_t < budget_s:
            ex, ez = og.pauli_noise(seed, args.p, St, chunk)
            sx, sz = og.syndrome(ex, ez)
            t = time.perf_counter()
            xh, zh = T.sandwich_decode(tg, w, sx, sz, iters, L0, fcs, cts)
            chunk_s.append(time.perf_counter() - t)
            t_t += chunk_s[-1]
            St += chunk
            xs.append(xh); zs.append(zh); sxs.append(sx); szs.append(sz)
        xh, zh, sx, sz = np.concatenate(xs), np.concatenate(zs), np.concatenate(sxs), np.concatenate(szs)
        torch.set_flush_denormal(False)
        ref = og.sandwich_decode(sx, sz, iters, [w] * (nl - 1), L0, factors=fcs, cn_types=cts)
        hx, hz = np.asarray(code.hx, dtype=np.int64), np.asarray(code.hz, dtype=np.int64)

        def solved(x, z):
            return ~(((x.astype(np.int64) @ hz.T) % 2 != sz).any(1) | ((z.astype(np.int64) @ hx.T) % 2 != sx).any(1))

        both = solved(ref["x_hat"], ref["z_hat"]) & solved(xh, zh)
        same = (ref["x_hat"] == xh).all(1) & (ref["z_hat"] == zh).all(1)
        out["cpu_baseline_tf_like"] = {
            "value": St / t_t, "unit": "codewords/s", "cores": nthr, "host_cpu_share": share, "kind": "port",
            "best_chunk_value": chunk / min(chunk_s),  # the box's CPU share is contended: the fastest 512-codeword chunk next to the mean
            "sample": f"{St} codewords of the same workload (Philox samples 0..{St - 1}), oracle/torch_cpu_baseline.py: the "
                      f"reference's op structure (decoding_q.py:732-767, feedback_gnn.py:161-188) on batch-minor [E,B] float32 torch "
                      f"CPU tensors, one op at a time, {nthr} intra-op threads, {t_t:.1f} s on {cpu_model}",
            "decisions_identical_to_oracle_on_samples_both_decode": float(same[both].mean()) if both.any() else None,
            "samples_both_decode": int(both.sum())}
    return out, check


def cpu_legs_c5(args, code, weights, num_iter, seed, factored):
    """CPU baselines of GNN_BP4 (c5): the C port (og_gnn_bp4, OpenMP over codewords) and the batched-matmul NumPy restatement
    (oracle/numpy_ref.py: Dense = matmul over [B, E, 40] tensors, how a framework executes gnn.py:573-751 on a host)."""
    import numpy as np
    from oracle.oracle import OracleGraph, host_cpu_share, num_threads, set_num_threads
    out, check = {}, None
    share = host_cpu_share()
    if "OMP_NUM_THREADS" not in os.environ:
        set_num_threads(share)
    cpu_model = _cpu_model()
    og = OracleGraph(code, forms="reassociated" if factored else "literal")
    og.set_gnn_order(factored)
    if args.cpu_baseline in ("both", "port"):
        S = args.cpu_sample
        if S < 0:
            probe = 2 * num_threads()
            ex, ez = og.pauli_noise(seed, args.p, 0, probe)
            sx, sz = og.syndrome(ex, ez)
            og.gnn_bp4(weights, sx, sz, num_iter)
            t = time.perf_counter()
            og.gnn_bp4(weights, sx, sz, num_iter)
            rate = probe / (time.perf_counter() - t)
            S = int(min(8192, max(2 * num_threads(), 12.0 * rate)))
            S -= S % max(1, num_threads())
        ex, ez = og.pauli_noise(seed, args.p, 0, S)
        sx, sz = og.syndrome(ex, ez)
        og.gnn_bp4(weights, sx[:8], sz[:8], 1)
        t = time.perf_counter()
        o = og.gnn_bp4(weights, sx, sz, num_iter)
        _, _, fl = og.residual(ex, ez, o["x_hat"], o["z_hat"])
        t_cpu = time.perf_counter() - t
        out["cpu_baseline"] = {"value": S / t_cpu, "unit": "codewords/s", "cores": num_threads(), "host_cpu_share": share,
                               "host_hw_threads": os.cpu_count(), "kind": "port",
                               "sample": f"{S} codewords of the same workload (same Philox samples 0..{S - 1}), "
                                         f"oracle/fgnn_oracle.c og_gnn_bp4 with OpenMP over codewords, {t_cpu:.1f} s on {cpu_model}",
                               "note": "the repo's own C port of gnn.py:383-423 (the reference's call raises as shipped and ships no "
                                       "weights, SURVEY §8 a17: repaired semantics, seeded weights)"}
        check = (S, o, fl)
    if args.cpu_baseline in ("both", "torch"):
        from oracle import numpy_ref as NR
        chunk = 32 if args.cpu_sample < 0 else max(4, min(args.cpu_sample, 32))
        budget_s, max_chunks = (8.0, 8) if args.cpu_sample < 0 else (1e9, max(1, min(args.cpu_sample, 256) // chunk))
        St, t_t = 0, 0.0
        ex, ez = og.pauli_noise(seed, args.p, 0, 4)
        sx, sz = og.syndrome(ex, ez)
        NR.gnn_bp4(code, weights, sx, sz, 1)
        n_chunks = 0
        while n_chunks < max_chunks and t_t < budget_s:
            ex, ez = og.pauli_noise(seed, args.p, St, chunk)
            sx, sz = og.syndrome(ex, ez)
            t = time.perf_counter()
            NR.gnn_bp4(code, weights, sx, sz, num_iter)
            t_t += time.perf_counter() - t
            St += chunk
            n_chunks += 1
        out["cpu_baseline_tf_like"] = {
            "value": St / t_t, "unit": "codewords/s", "cores": share, "host_cpu_share": share, "kind": "port",
            "sample": f"{St} codewords of the same workload (Philox samples 0..{St - 1}), oracle/numpy_ref.py gnn_bp4: the reference's op "
                      f"structure (gnn.py:573-610, 714-751) as batched float32 NumPy matmuls over [B,E,.] tensors, {t_t:.1f} s on {cpu_model}"}
    return out, check


def gnnbp4_roofline(code_name, dims, launches, B, num_iter, factored):
    """`roofline` of the c5 line: the GNN_BP4 launch in the reference's FLOPs against the f32 MFMA (= f32 vector) peak, with the offline PMC
    counts of this kernel at this shape (which pipe the time went to) while the kernel sources still hash to what was profiled."""
    import numpy as np
    n, m, E = dims
    dom = [ms for ms, it, b in launches if it == PROF_TAG_GNNBP4 and b == B]
    dom_ms = float(np.mean(dom)) if dom else None
    flops = gnnbp4_flops_per_codeword(n, m, E, num_iter) * B
    tf = flops / (dom_ms * 1e-3) / 1e12 if dom_ms else None
    ent, tsrc = pmc_entry("gnnbp4", pmc_key("gnnbp4", code_name, B, num_iter, reassociated=factored))
    traffic = ent.get("hbm_bytes_per_launch") if ent else None
    vi = ent.get("valu_wave_insts_per_launch") if ent else None
    mi = ent.get("mfma_insts_per_launch") if ent else None
    simd_cycles = 1024 * 2.4e9 * dom_ms * 1e-3 if dom_ms else None
    return {"bound": "valu" if (ent and not mi) else "mfma",
            "kernel": f"GNN_BP4 kernel, {num_iter} iterations, B={B}",
            "achieved": tf, "peak": GNN_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": (tf / GNN_PEAK_TFLOPS if tf else None) if ent else None,
            "reference_tflops": tf, "reference_tflops_frac_of_f32_peak": tf / GNN_PEAK_TFLOPS if tf else None,
            "algorithmic_flops_per_launch": flops,
            "executed_flops_per_launch": gnnbp4_flops_per_codeword_factored(n, m, E, num_iter) * B if factored else None,
            "traffic": traffic, "traffic_source": tsrc,
            "mfma_insts_per_launch": mi, "valu_wave_insts_per_launch": vi,
            "fp32_lane_busy_frac": (mi * 32 + vi * 2) / simd_cycles if (simd_cycles and mi is not None and vi is not None) else None,
            "mfma_pipe_busy_frac": mi * 32 / simd_cycles if (simd_cycles and mi is not None) else None,
            "library_is_the_profiled_binary": ent.get("library_is_the_profiled_binary") if ent else None,
            "avg_launch_ms": dom_ms, "launches_timed": len(dom),
            "hbm_frac": traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if (traffic and dom_ms) else None,
            "hbm_peak_GBs": HBM_PEAK_GBS,
            "note": "achieved = the reference's algorithm in FLOPs (SURVEY §8d: 0.83 GFLOP per [[1270,28]] codeword for 10 "
                    "iterations) / this run's HIP-event launch time, against the 157.3 TFLOP/s f32 MFMA (= f32 vector) peak; "
                    "frac is quoted only while profiles/traffic.json holds PMC counts of this kernel's sources at this shape "
                    "(they say which pipe the time went to: fp32_lane_busy_frac = (32 cycles x MFMAs + 2 cycles x VALU "
                    "wave-instructions) / (1024 SIMDs x 2.4 GHz x time)); executed_flops_per_launch = what the factored "
                    "association executes for the same function"}


def feedback_gnn_roofline(code_name, dims, launches, B, factored, stream):
    """`roofline.gnn` of the c3 / c4 lines: the feedback-GNN launch, priced against VALU issue (streaming kernel, the default) or against
    the f32 MFMA peak in the reference's FLOPs (MFMA-tile kernel)."""
    import numpy as np
    n, _, E = dims
    gnn = [ms for ms, it, b in launches if it == PROF_TAG_GNN and b == B]
    gnn_ms = float(np.mean(gnn)) if gnn else None
    gnn_flops = gnn_flops_per_codeword(n, E) * B
    gnn_exec = (gnn_flops_per_codeword_factored(n, E) if factored else gnn_flops_per_codeword(n, E)) * B
    gnn_tf = gnn_flops / (gnn_ms * 1e-3) / 1e12 if gnn_ms else None
    gent, gsrc = pmc_entry("gnn", pmc_key("gnn", code_name, B, reassociated=factored))
    gvi = gent.get("valu_wave_insts_per_launch") if gent else None
    entry_is_of_the_mfma_kernel = bool(gent and gent.get("mfma_insts_per_launch"))  # the streaming kernel issues no MFMA
    if gent and entry_is_of_the_mfma_kernel == stream:
        gent, gsrc, gvi = None, "profiles/traffic.json holds the counts of the other feedback-GNN kernel (MFMA tiles vs streaming VALU)", None
    common = {"avg_launch_ms": gnn_ms, "launches_timed": len(gnn), "algorithmic_flops_per_launch": gnn_flops,
              "executed_flops_per_launch": gnn_exec,
              "reference_tflops": gnn_tf, "reference_tflops_frac_of_f32_peak": gnn_tf / GNN_PEAK_TFLOPS if gnn_tf else None,
              "executed_frac": gnn_exec / (gnn_ms * 1e-3) / 1e12 / GNN_PEAK_TFLOPS if gnn_ms else None,
              "traffic": gent.get("hbm_bytes_per_launch") if gent else None,
              "mfma_insts_per_launch": gent.get("mfma_insts_per_launch") if gent else None,
              # MFMA-busy of this launch: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs-worth of cycles is what the counter sums over) — 0 on the
              # streaming VALU kernel, which issues no MFMA at all (an f32 MFMA shares the FP32 lanes on gfx950: DESIGN.md §4.2)
              "mfma_busy_cycles_per_launch": gent.get("mfma_busy_cycles") if gent else None,
              "mfma_busy_frac": (gent.get("mfma_busy_cycles") / (1024 * 2.4e9 * gnn_ms * 1e-3)
                                 if (gent and gent.get("mfma_busy_cycles") is not None and gnn_ms) else None),
              "valu_wave_insts_per_launch": gvi, "traffic_source": gsrc}
    if stream:
        # the default: factored association on the streaming VALU kernel (no MFMA: on gfx950 an f32 MFMA has the f32 VALU's rate and
        # only pads the 40 / 20 / 3-row layers to 16-row tiles).  Priced like the BP4 kernel: VALU wave-instructions per second.
        g_ach = gvi / (gnn_ms * 1e-3) / 1e9 if (gvi and gnn_ms) else None
        return dict({"bound": "valu", "kernel": f"feedback-GNN streaming VALU kernel ({'factored' if factored else 'literal'} association), B={B}",
                     "achieved": g_ach, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                     "frac": g_ach / VALU_PEAK_GINST if g_ach else None}, **common,
                    note="frac = SQ_INSTS_VALU per launch (offline PMC pass, source-fingerprinted) / this run's launch time / "
                         "(1024 SIMDs x 2.4 GHz / 2); `reference_tflops` prices the reference's algorithm (SURVEY §8d: 13.4 MFLOP per "
                         "[[882,24]] codeword, one 40->20 Dense per EDGE) per second, `executed_frac` the FLOPs the factored "
                         "association executes, both against the 157.3 TFLOP/s f32 peak")
    return dict({"bound": "mfma", "kernel": f"feedback-GNN MFMA-tile kernel ({'factored' if factored else 'literal'} association), B={B}",
                 "achieved": gnn_tf, "peak": GNN_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": gnn_tf / GNN_PEAK_TFLOPS if gnn_tf else None}, **common,
                note="`achieved` prices the reference's algorithm (SURVEY §8d: 13.4 MFLOP per [[882,24]] codeword, one 40->20 "
                     "Dense per EDGE) against the f32 MFMA peak; the factored association executes `executed_flops_per_launch` "
                     "(one 40->20 Dense per qubit and side) for the same function")


def sandwich_roofline(code_name, dims, launches, launches_per_step, B, iters, factored, stream, shared_lse, cn_type="boxplus-phi"):
    """`roofline` of the c3 / c4 lines: the first decoder's BP4 launch against VALU issue (utilisation) and against the hardware
    transcendental rate (algorithmic efficiency), the SURVEY §8d streaming figure as an effective bandwidth, measured HBM bytes."""
    import numpy as np
    n, m, E = dims
    # launches of a step arrive in order: BP4 (first decoder), then (feedback GNN, BP4) per further layer
    per_step = [launches[i:i + launches_per_step] for i in range(0, len(launches), launches_per_step)]
    dom = [s[0][0] for s in per_step if len(s) == launches_per_step and s[0][1] == iters[0] and s[0][2] == B]
    dom_ms = float(np.mean(dom)) if dom else None
    later = [ms for s in per_step if len(s) == launches_per_step for (ms, it, b) in s[2::2]]
    alg_bytes = algorithmic_bytes_per_codeword(n, m, E, iters[0]) * B
    eff_gbs = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else None
    # VALU instruction counts and HBM bytes per launch are NOT measured by this run: they come from the rocprofv3 PMC passes of
    # tools/refresh_traffic.sh at the same shape, guarded by the fingerprint of the kernel sources (pmc_entry)
    # one entry per check-node rule (a template argument of the kernel); the normalization factor is a runtime multiplier
    ent, tsrc = pmc_entry("bp4", pmc_key("bp4", code_name, B, iters[0], cn_type, reassociated=shared_lse))
    traffic = ent.get("hbm_bytes_per_launch") if ent else None
    vi = ent.get("valu_wave_insts_per_launch") if ent else None
    achieved = vi / (dom_ms * 1e-3) / 1e9 if (vi and dom_ms) else None
    n_exp, n_log = bp4_transcendentals_per_codeword(n, m, E, iters[0], shared_lse, cn_type)
    trans = (n_exp + n_log) * B
    return {"bound": "valu",
            "kernel": f"bp4_kernel<{cn_type}>, first decoder (constant channel LLR), {iters[0]} iterations, B={B}",
            "achieved": achieved, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
            "frac": achieved / VALU_PEAK_GINST if achieved else None,
            "traffic": traffic, "traffic_source": tsrc,
            "valu_wave_insts_per_launch": vi,
            # the BP4 kernel issues no MFMA (north_star: "MFMA used only for the dense per-node MLP of the feedback GNN"): SQ_INSTS_MFMA of the
            # same PMC pass, and the MFMA pipe's busy cycles over the launch's SIMD cycles
            "mfma_insts_per_launch": ent.get("mfma_insts_per_launch") if ent else None,
            "mfma_busy_frac": (ent.get("mfma_busy_cycles") / (1024 * 2.4e9 * dom_ms * 1e-3)
                               if (ent and ent.get("mfma_busy_cycles") is not None and dom_ms) else None),
            "library_is_the_profiled_binary": ent.get("library_is_the_profiled_binary") if ent else None,
            "avg_launch_ms": dom_ms, "launches_timed": len(dom),
            "later_decoders_avg_launch_ms": float(np.mean(later)) if later else None,
            "transcendental_evals_per_launch": trans,
            "transcendental_evals_per_codeword": {"exp": n_exp, "log": n_log},
            "hw_transcendental_peak_per_s": HW_TRANSCENDENTAL_PEAK,
            "frac_of_hw_transcendental_rate": trans / (dom_ms * 1e-3) / HW_TRANSCENDENTAL_PEAK if dom_ms else None,
            "valu_insts_per_transcendental": vi * 64 / trans if vi else None,
            "hbm_frac": traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if (traffic and dom_ms) else None,
            "effective_bandwidth_frac": eff_gbs / HBM_PEAK_GBS if eff_gbs else None,
            "effective_bandwidth_GBs": eff_gbs, "hbm_peak_GBs": HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg_bytes,
            # the same figures in the shape of the task contract's roofline object (bound "hbm"): SURVEY 8(d) bytes per launch / the launch's
            # HIP-event duration against the 8 TB/s peak — an EFFECTIVE bandwidth (frac > 1: the messages never travel); traffic = measured bytes
            "contract_hbm": {"bound": "hbm", "achieved": eff_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": eff_gbs / HBM_PEAK_GBS if eff_gbs else None, "traffic": traffic, "effective": True},
            "gnn": feedback_gnn_roofline(code_name, dims, launches, B, factored, stream) if len(iters) > 1 else None,
            "note": "bound = VALU issue: all messages stay in LDS for the 64 iterations, the kernel issues the exp/log "
                    "instruction streams of fgnn_math.h (DESIGN.md §4.1).  frac = SQ_INSTS_VALU per launch (offline PMC pass, "
                    "source-fingerprinted) / this run's HIP-event launch time / (1024 SIMDs x 2.4 GHz / 2): issue UTILISATION of "
                    "the library's own instruction stream, not algorithmic efficiency.  frac_of_hw_transcendental_rate = "
                    "every exp and log of the fixed dataflow (transcendental_evals_per_launch) / launch time / the chip's "
                    "quarter-rate v_exp_f32 / v_log_f32 rate: the software routines that make CPU and GPU bit-equal spend "
                    "valu_insts_per_transcendental lane-instructions per evaluation where the hardware unit would spend one "
                    "quarter-rate instruction.  effective_bandwidth_frac = SURVEY §8d streaming-model bytes / time / 8 TB/s (can "
                    "exceed 1: nothing streams); hbm_frac = measured HBM bytes / time / 8 TB/s"}


def gnnbp4_forms_agreement(g, wdev, sx, sz, num_iter, workspace):
    """GNN_BP4 on the same syndromes under the opt-in factored and the literal (default) association, compared per sample.
    make_hard_decision = argmin over (0, X, Z, Y) (gnn.py:359-367): a qubit whose two smallest candidates are closer than twice the
    LLR tolerance may legitimately decide either way under a 1e-6 perturbation — with untrained (seeded) weights the marginals of
    many qubits sit that close to the boundary.  A differing decision BEYOND the tolerance would be a defect."""
    import torch
    prev = g.gnn_factored
    res = []
    for f in (True, False):
        g.set_gnn_factored(f)
        res.append(g.gnn_bp4_decode(wdev, sx, sz, num_iter, return_logits=False, workspace=workspace))
    g.set_gnn_factored(prev)
    a, b = res
    d = (a["llr"] - b["llr"]).abs().flatten(1).max(1).values
    X, Y, Z = a["llr"][:, 0], a["llr"][:, 1], a["llr"][:, 2]
    cand = torch.stack([torch.zeros_like(X), X, Z, Y], -1).sort(-1).values
    margin = cand[..., 1] - cand[..., 0]
    qdiff = (a["x_hat"] != b["x_hat"]) | (a["z_hat"] != b["z_hat"])
    return {"samples": int(sx.shape[0]), "decisions_differ": int(qdiff.any(1).sum()),
            "decisions_differ_beyond_llr_tolerance": int((qdiff & (margin > 2e-4)).any(1).sum()),
            "max_decision_margin_where_they_differ": float(margin[qdiff].max()) if bool(qdiff.any()) else 0.0,
            "max_abs_dllr": float(d.max()), "samples_gt_1e_4": int((d > 1e-4).sum())}


def sandwich_extras(g, model, code, decs, G, iters, B, p, seed, cn_type="boxplus-phi", factor=1.0, p0=0.05):
    """`extras` of a single-GPU c3 / c4 run: the variants that are NOT the headline — BP4 alone (configs[1]), the product default (exact
    shortcuts, compaction: identical outputs) and the opt-in hardware-transcendental BP4 with its measured distance from the exact kernel."""
    import numpy as np
    import torch
    import feedback_gnn_amd as F
    L0 = model._llr_const(p)
    ex, ez = g.pauli_noise(seed, p, 0, B)
    sx, sz = g.syndrome(ex, ez)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    t_bp = timed(lambda: g.bp4_decode(sx, sz, iters[0], cn_type, factor, llr_const=L0))
    g.set_saturation_shortcut(True)
    t_bp_s = timed(lambda: g.bp4_decode(sx, sz, iters[0], cn_type, factor, llr_const=L0))
    cs = torch.zeros(3, dtype=torch.int64, device="cuda")
    t_s = timed(lambda: model.mc_step(B, p, cs))
    g.set_saturation_shortcut(False)
    model_c = F.Sandwich_BP_GNN_Evaluation_Model(code, decs, [G] * (len(iters) - 1), num_layers=len(iters), p0=p0, seed=seed, compact=True)
    cc = torch.zeros(3, dtype=torch.int64, device="cuda")
    t_c = timed(lambda: model_c.mc_step(B, p, cc))
    g.set_saturation_shortcut(True)
    t_cs = timed(lambda: model_c.mc_step(B, p, cc))
    g.set_saturation_shortcut(False)
    # the headline's fixed dataflow with consecutive batches alternating between two HIP streams (own workspaces, shared atomic counters)
    model_2s = F.Sandwich_BP_GNN_Evaluation_Model(code, decs, [G] * (len(iters) - 1), num_layers=len(iters), p0=p0, seed=seed, streams=2)
    c2 = torch.zeros(3, dtype=torch.int64, device="cuda")

    def two_stream_steps():
        for _ in range(2):
            model_2s.mc_step(B, p, c2)
        model_2s.join()

    t_2s = timed(two_stream_steps) / 2
    hw_info = None
    if cn_type == "boxplus-phi":  # the hardware-transcendental variant exists for the phi rule only
        # OPT-IN variant, never the headline: the phi rule on v_exp_f32 / v_log_f32 (FGNN_OPT_HW_TRANSCENDENTALS), and how
        # far its results are from the exact kernel's on this very batch — the measured price of bit-exactness
        exact = g.bp4_decode(sx, sz, iters[0], cn_type, factor, llr_const=L0)
        g.set_hw_transcendentals(True)
        t_hw = timed(lambda: g.bp4_decode(sx, sz, iters[0], cn_type, factor, llr_const=L0))
        hw = g.bp4_decode(sx, sz, iters[0], cn_type, factor, llr_const=L0)
        g.set_hw_transcendentals(False)
        same_dec = ((exact["x_hat"] == hw["x_hat"]).all(1) & (exact["z_hat"] == hw["z_hat"]).all(1))
        ones = torch.ones(B, dtype=torch.uint8, device="cuda")
        conv_e = g.flag_update(exact["x_hat"], exact["z_hat"], sx, sz, ones.clone()) == 0
        conv_h = g.flag_update(hw["x_hat"], hw["z_hat"], sx, sz, ones.clone()) == 0
        both = conv_e & conv_h
        dl = (exact["llr"] - hw["llr"]).abs().flatten(1).max(1).values
        nb = max(int(both.sum()), 1)
        # decisions that differ by a stabilizer (difference in the row space of hx / hz <=> zero syndrome under hx_perp / hz_perp)
        hxp = torch.from_numpy(np.asarray(code.hx_perp)).to("cuda").float()
        hzp = torch.from_numpy(np.asarray(code.hz_perp)).to("cuda").float()
        dxb, dzb = (exact["x_hat"] ^ hw["x_hat"])[both].float(), (exact["z_hat"] ^ hw["z_hat"])[both].float()
        same_class = ~(((dxb @ hxp.t()) % 2).bool().any(1) | ((dzb @ hzp.t()) % 2).bool().any(1))
        hw_info = {"cw_per_s": B / t_hw, "speedup_vs_exact_kernel": t_bp / t_hw,
                   "samples": B, "converged_exact": int(conv_e.sum()), "converged_hw": int(conv_h.sum()),
                   "identical_decisions_all_samples": float(same_dec.float().mean()),
                   "identical_decisions_on_samples_both_converge": float(same_dec[both].float().mean()) if int(both.sum()) else None,
                   "same_correction_class_on_samples_both_converge": float(same_class.double().mean()) if int(both.sum()) else None,
                   "max_abs_llr_diff_le_1e-4_on_samples_both_converge": float((dl[both] <= 1e-4).sum()) / nb,
                   "median_abs_llr_diff_on_samples_both_converge": float(dl[both].median()) if int(both.sum()) else None}
    return {"bp4_only_cw_per_s (configs[1])": B / t_bp,
            "bp4_only_HW_TRANSCENDENTALS_opt_in_NOT_bit_exact (v_exp_f32/v_log_f32, phi clip points pinned, fixed dataflow)": hw_info,
            "bp4_only_product_default_cw_per_s (exact saturation shortcut + fixed-point exit, identical outputs)": B / t_bp_s,
            "sandwich_product_default_cw_per_s (exact saturation shortcut + fixed-point exit, identical outputs)": B / t_s,
            "sandwich_compacted_cw_per_s (feedback rounds only on flagged samples, same outputs)": B / t_c,
            "sandwich_compacted_product_default_cw_per_s (all exact optimisations, same outputs)": B / t_cs,
            "sandwich_two_streams_cw_per_s (fixed dataflow, Sandwich_BP_GNN_Evaluation_Model(streams=2): consecutive batches alternate between two "
            "HIP streams; same samples, same counters)": B / t_2s}


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this pool (before any GPU runtime loads); a caller's setting wins
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    build_once(args.no_build)  # before the first GPU call of this process

    import numpy as np
    import torch

    SEED = 0x5EED
    is_c5 = args.config == "c5"
    iters = [int(x) for x in args.iters.split(",")]
    code, wname = make_code(args.code)
    wname = CONFIGS[args.config].get("weights", wname)  # (n1270_coarse: the coarse GNN of n1270.ipynb cell 9)
    # the library default = the reference's association (one Dense per edge); FGNN_BENCH_GNN_ORDER=factored times the opt-in re-association
    factored = os.environ.get("FGNN_BENCH_GNN_ORDER", "literal") == "factored"
    # both associations run on the streaming VALU kernel by default (the library's choice at these batch sizes);
    # FGNN_BENCH_GNN_KERNEL=mfma times the MFMA-tile kernel
    stream = os.environ.get("FGNN_BENCH_GNN_KERNEL", "stream") != "mfma"
    shared_lse = os.environ.get("FGNN_BENCH_BP4_LSE", "literal") == "shared"    # likewise for the qubit update's log-sum-exp term
    c5_weights = gnnbp4_seeded_weights(0) if is_c5 else None

    # ---- CPU baselines first (rank 0 of a single-GPU run), in a CHILD interpreter: its OpenMP / torch intra-op / BLAS thread pools end
    # with it, so the process that times the GPU never shares its 16 granted CPUs with pools that are still winding down (measured on
    # c1, whose step is 0.14 ms of which the host's launch calls are most: 0.8-1.2 ms per step for the first ~100 ms after in-process
    # CPU legs).  The child never touches the GPU; this parent has not initialised it yet.
    cpu_out, cpu_check = {}, None
    if args.cpu_legs_to:
        if is_c5:
            res = cpu_legs_c5(args, code, c5_weights, iters[0], SEED, factored)
        else:
            res = cpu_legs(args, code, wname, iters, SEED, factored)
        import pickle
        with open(args.cpu_legs_to, "wb") as f:
            pickle.dump(res, f)
        return
    if rank == 0 and world == 1 and args.cpu_sample != 0 and args.cpu_baseline != "none":
        import pickle
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "cpu_legs.pkl")
            try:
                rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--cpu-legs-to", path, "--no-build"], cwd=ROOT,
                                     stdout=sys.stderr)  # this process' stdout carries the ONE JSON line and nothing else
                if rc != 0:
                    raise RuntimeError(f"exit code {rc}")
                with open(path, "rb") as f:
                    cpu_out, cpu_check = pickle.load(f)
            except Exception as e:  # never lose the line to the isolation: time the legs in this process, as rounds 1-4 did
                sys.stderr.write(f"bench.py: the CPU-baseline child failed ({e}); running the CPU legs in-process\n")
                if is_c5:
                    cpu_out, cpu_check = cpu_legs_c5(args, code, c5_weights, iters[0], SEED, factored)
                else:
                    cpu_out, cpu_check = cpu_legs(args, code, wname, iters, SEED, factored)

    # one process per GPU; FGNN_BENCH_BACKEND=gloo (self-test of the multi-process flow on a 1-GPU box) lets several
    # ranks share a device and reduces through host memory
    backend = os.environ.get("FGNN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1 or (backend == "nccl" and local_rank >= ndev):
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, {ndev} visible")
    dev_index = local_rank % ndev if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if backend == "nccl":  # RCCL: communicator bound to this rank's device, created eagerly so that a failure shows here
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:  # this process is a fresh child (launch_ranks / torchrun): report and leave with a non-zero code
            sys.stderr.write(f"bench.py: rank {rank}: init_process_group({backend}) failed: {e}\n")
            sys.exit(3)

    # ---- pre-flight of the multi-GPU run (the reference pins one process per GPU id, n882.py:9-25): every rank reports the device it
    # really sits on; a rank that is not on cuda:LOCAL_RANK under RCCL leaves with its own non-zero code, and rank 0 prints the map
    if backend == "nccl" and torch.cuda.current_device() != local_rank:
        sys.stderr.write(f"bench.py: rank {rank}: on cuda:{torch.cuda.current_device()}, expected cuda:{local_rank}\n")
        sys.exit(6)
    props = torch.cuda.get_device_properties(dev_index)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
          "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    distinct = len({(r["device_uuid"], r["pci_bus_id"], r["device_index"]) for r in ranks_info}) == world
    dist_info = {"world_size": dist.get_world_size() if dist is not None else 1,
                 "backend": dist.get_backend() if dist is not None else None,
                 "collective_library": ("RCCL (torch.distributed backend 'nccl' on ROCm)" if backend == "nccl" else backend) if dist is not None else None,
                 "ranks": ranks_info, "one_distinct_device_per_rank": distinct}
    if dist is not None and backend == "nccl" and not distinct:
        sys.stderr.write(f"bench.py: rank {rank}: the {world} ranks do not sit on {world} distinct devices: {ranks_info}\n")
        sys.exit(7)

    def allreduce(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        dist.all_reduce(c, op=op)
        return c.to(t.device)

    def allgather(t):
        if backend == "nccl":
            out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t)
            return out
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return torch.stack(parts).to(t.device)

    import feedback_gnn_amd as F
    from feedback_gnn_amd._lib import lib
    from feedback_gnn_amd.graph import GnnBp4Weights
    lib()  # fail loudly if the HIP extension is missing

    B, K, W = args.batch, args.steps, args.warmup
    sample_log = []  # [first, last) of the global Philox sample range of every step this rank has run, in order
    if is_c5:
        from feedback_gnn_amd.graph import TannerGraph
        g = TannerGraph(code)
        g.set_gnn_factored(factored)
        wdev = GnnBp4Weights(c5_weights, g.device)
        ws = torch.empty(lib().fgnn_gnnbp4_weights_workspace_bytes(g.handle, wdev.handle, B), dtype=torch.uint8, device=g.device)
        state = {"next": 0}

        def c5_decode(first):
            ex, ez = g.pauli_noise(SEED, args.p, first, B)
            sx, sz = g.syndrome(ex, ez)
            o = g.gnn_bp4_decode(wdev, sx, sz, iters[0], return_logits=False, workspace=ws)
            o["noise_x"], o["noise_z"] = ex, ez
            return o

        def step(counts):
            first = state["next"] + rank * B
            state["next"] += world * B
            sample_log.append((first, first + B))
            o = c5_decode(first)
            _, _, flags = g.residual(o["noise_x"], o["noise_z"], o["x_hat"], o["z_hat"], want_arrays=False)
            g.count_flags(flags, counts)

        launches_per_step = 1
        model = decs = G = None
    else:
        decs = [F.QLDPCBPDecoder(code=code, num_iter=iters[0], normalization_factor=args.factor, cn_type=args.cn_type, stage_one=True)]
        g = decs[0].graph
        g.set_gnn_factored(factored)
        g.set_gnn_stream(stream)
        # which kernel that selects: the library streams launches of 4 096 codewords or more, smaller ones run on the MFMA tiles
        # (fgnn_gnn.hip `stream_pays`: FGNN_OPT_GNN_STREAM = 1) — the labels and the roofline below name the kernel that runs
        stream = stream and B >= 4096
        g.set_bp4_shared_lse(shared_lse)
        for it in iters[1:]:
            decs.append(F.QLDPCBPDecoder(code=code, num_iter=it, normalization_factor=args.factor, cn_type=args.cn_type,
                                         stage_one=True, graph=g))
        G = F.Feedback_GNN(code=code, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean",
                           activation="tanh", use_bias=True, graph=g)
        F.load_weights(G, wname)
        def make_model(streams):
            return F.Sandwich_BP_GNN_Evaluation_Model(code, decs, [G] * (len(iters) - 1), num_layers=len(iters), p0=args.p0, seed=SEED,
                                                      rank=rank, world_size=world, streams=streams)

        model = make_model(args.streams)

        def step(counts):
            first = model.next_sample_range(B)[0]
            sample_log.append((first, first + B))
            model.mc_step(B, args.p, counts)

        launches_per_step = 2 * len(iters) - 1  # BP4 launches + feedback-GNN launches
    # Headline = the reference's fixed dataflow: every exp/log of every iteration is evaluated.  The product
    # default (exact wave-uniform shortcut for saturated nodes, same bits) is timed separately under `extras`.
    g.set_saturation_shortcut(False)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps, profile, step=step):
        """EXACTLY `steps` steps between barrier + synchronize on both sides (the device-wide synchronize also waits for a model's side
        streams); (max-over-ranks seconds, this rank's seconds, launches)."""
        counts = torch.zeros(3, dtype=torch.int64, device="cuda")
        if profile:
            g.profile_enable(steps * launches_per_step)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(counts)
        sync()
        own = time.perf_counter() - t0
        launches = g.profile_read() if profile else []
        if profile:
            g.profile_enable(0)
        elapsed = own
        if dist is not None:
            elapsed = float(allreduce(torch.tensor([own], dtype=torch.float64, device="cuda"), dist.ReduceOp.MAX).item())
        return elapsed, own, launches, counts

    try:
        # Settle phase (untimed, before the W warm-up steps, off the sample stream): the CPU legs above leave the GPU idle for tens of
        # seconds and the host's thread pools (OpenMP, torch intra-op) winding down; a configuration whose step is 0.15 ms (c1) would
        # otherwise spend its whole warm-up AND timed region inside that transient (measured: 1.16 instead of 0.14 ms per step).  GPU work
        # for at least --settle-ms of wall time.
        # The settle work is an elementwise torch kernel, NOT one of this library's kernels: a rocprofv3 --stats summary of this command then
        # averages each of our kernel symbols over the warm-up and timed launches only.
        t_settle = time.perf_counter()
        sa = torch.ones(1 << 20, dtype=torch.float32, device=g.device)  # 4 MB: keeps the queue busy without heating the HBM
        while time.perf_counter() - t_settle < args.settle_ms * 1e-3:
            for _ in range(32):
                sa.mul_(1.0)
            torch.cuda.synchronize()
        del sa
        warm = torch.zeros(3, dtype=torch.int64, device="cuda")
        for _ in range(W):
            step(warm)
        n_before = len(sample_log)
        elapsed, own_elapsed, launches, counts = timed_region(K, True)
        timed_ranges = sample_log[n_before:]
        own_counts = [int(v) for v in counts.cpu().numpy()]  # this rank's counters before the all-reduce
        per_rank_ms = [own_elapsed / K * 1e3]
        rank_rows = [[rank, timed_ranges[0][0], timed_ranges[-1][1]] + own_counts]
        if dist is not None:
            per_rank_ms = [float(v) for v in allgather(torch.tensor([own_elapsed / K * 1e3], dtype=torch.float64, device="cuda")).flatten()]
            # every rank's timed sample range and its own counters (int64: [rank, first, last, flagged, block_errors, samples]), by the same
            # all-gather path as the step times, and the per-step ranges, so that rank 0 can prove the sharding (below)
            rank_rows = allgather(torch.tensor(rank_rows[0], dtype=torch.int64, device="cuda")).cpu().numpy().tolist()
            step_ranges = allgather(torch.tensor(timed_ranges, dtype=torch.int64, device="cuda")).cpu().numpy().tolist()
            counts = allreduce(counts, dist.ReduceOp.SUM)
        else:
            step_ranges = [[list(r) for r in timed_ranges]]
        # ---- the same step under the OTHER forms (single-GPU runs): the opt-in re-associations when the headline is the library
        # default, the literal forms when an A/B run made the re-associated ones the headline; same bracket, same sample stream ----
        other_forms = None
        headline_is_literal = not factored and (is_c5 or not shared_lse)
        if world == 1 and not args.no_literal and (headline_is_literal or factored or shared_lse):
            to_reassociated = headline_is_literal
            g.set_gnn_factored(to_reassociated)
            if not is_c5:
                g.set_bp4_shared_lse(to_reassociated)
            step(torch.zeros(3, dtype=torch.int64, device="cuda"))  # untimed: first launch of these kernel variants
            o_elapsed, _, _, _ = timed_region(K, False)
            g.set_gnn_factored(factored)
            if not is_c5:
                g.set_bp4_shared_lse(shared_lse)
            other_forms = {"value": world * B * K / o_elapsed, "unit": "codewords/s", "ms_per_step": o_elapsed / K * 1e3, "steps": K,
                           "forms": "re-associated (opt-in)" if to_reassociated else "literal",
                           "what": ("the same step, same sample stream, with the two OPT-IN re-associations on: "
                                    + ("GNN_BP4 message MLP factored (FGNN_OPT_GNN_FACTORED = 1)" if is_c5 else
                                       "the qubit update's log-sum-exp term once per qubit and side (FGNN_OPT_BP4_SHARED_LSE = 1) and the "
                                       "feedback GNN's Dense layers factored (FGNN_OPT_GNN_FACTORED = 1)")
                                    + " — the same real-number functions, statistically the same decoder, NOT the reference's float32 "
                                      "operation sequence (see forms_agreement)") if to_reassociated else
                                   ("the same step, same sample stream, with the reference's formulas term by term (the library default): "
                                    + ("GNN_BP4 message MLP once per edge (gnn.py:573-610, 714-751)" if is_c5 else
                                       "one log-sum-exp per edge in the qubit update (decoding_q.py:254-273) and one 40 -> 20 Dense per edge "
                                       "in the feedback GNN (feedback_gnn.py:175-184)"))}
        # ---- the reference's published workloads run 5 000-codeword batches (n882.py:39): one such batch fills a fraction of the chip,
        # so the same step is also timed with consecutive batches alternating between two HIP streams (same samples, same counters)
        other_streams = None
        if args.config in PUBLISHED_CONFIGS and not is_c5:
            ns = 1 if args.streams > 1 else 2
            model_o = make_model(ns)

            def step_o(counts):
                model_o.mc_step(B, args.p, counts)

            for _ in range(max(W, 2)):
                step_o(torch.zeros(3, dtype=torch.int64, device="cuda"))
            o_elapsed, _, _, o_counts = timed_region(K, False, step_o)
            other_streams = {"streams": ns, "value": world * B * K / o_elapsed, "unit": "codewords/s", "ms_per_step": o_elapsed / K * 1e3,
                             "steps": K, "what": f"the same step on Sandwich_BP_GNN_Evaluation_Model(streams={ns}): "
                                                 + ("consecutive batches alternate between two HIP streams (own workspaces, shared atomic "
                                                    "counters), fixed dataflow, same Philox samples" if ns == 2 else "one stream")}
    except Exception as e:
        if dist is None:
            raise
        sys.stderr.write(f"bench.py: rank {rank}: {type(e).__name__}: {e}\n")
        sys.exit(4)  # a failed collective / launch ends this (fresh child) rank with a non-zero code; the launcher ends the others
    cnt = counts.cpu().numpy()
    total_cw = world * B * K
    value = total_cw / elapsed

    out = None
    if rank == 0:
        n, m, E = g.n, g.m_x + g.m_z, g.E_x + g.E_z
        info = g.info()
        common = {"value": value, "unit": "codewords/s", "n_gpus": world, "steps": K, "warmup": W,
                  "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": "f32", "data": "synthetic"}
        counts_obj = {"flagged": int(cnt[0]), "block_errors": int(cnt[1]), "samples": int(cnt[2])}
        if is_c5:
            out = dict({"metric": "decoded codewords/sec, [[1270,28]] GNN_BP4 full-GNN decoder, 10 iterations"}, **common, **{
                "config": {"workload": f"{code.name} GNN_BP4(num_embed_dims=20, num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, "
                                       f"num_iter={iters[0]}, mean, tanh, use_bias) with seeded glorot weights (the reference ships none), "
                                       f"depolarizing p={args.p}, noise+syndrome+decode+residual+count on device (BASELINE.json {CONFIGS['c5']['baseline']})",
                           "code": code.name, "batch_per_gpu": B, "global_batch": world * B, "gnn_bp4_iters": iters[0], "p": args.p,
                           "parallelism": f"batch-sharded x{world}, no data-path collective", "seed": SEED,
                           "gnn_association": "factored" if factored else "literal"},
                "per_rank_ms": per_rank_ms,
                "roofline": gnnbp4_roofline(args.code, (n, m, E), launches, B, iters[0], factored),
                "counts": counts_obj})
        else:
            cfg = CONFIGS[args.config]
            is_cfg_shape = args.code == cfg["code"] and args.iters == cfg["iters"]
            nk = "[[882,24]]" if args.code == "ghp882" else "[[1270,28]]"
            metric = (f"decoded codewords/sec, {nk} BP4 {iters[0]} iters" if len(iters) == 1 else
                      f"decoded codewords/sec, {nk} BP4-{iters[0]} + {len(iters) - 1} x (feedback-GNN + BP4-{iters[1]})" if len(iters) > 2 else
                      "decoded codewords/sec, [[882,24]] 64-iter BP4 + feedback-GNN" if args.code == "ghp882" else
                      "decoded codewords/sec, [[1270,28]] BP4 64+64 iters w/ feedback-GNN")
            ref_construction = ("the reference's constructor defaults, decoding_q.py:18-22" if (args.cn_type, args.factor) == ("boxplus", 0.625) else
                                "as the reference's scripts construct it, n882.py:56-62" if (args.cn_type, args.factor) == ("boxplus-phi", 1.0) else
                                "the QLDPC.ipynb cell 11 helper's construction" if (args.cn_type, args.factor) == ("boxplus-phi", 0.625) else
                                "as QLDPC.ipynb cell 12 constructs it" if (args.cn_type, args.factor, args.p0) == ("boxplus-phi", 0.8, 0.3) else
                                "construction given on the command line")
            out = dict({"metric": metric}, **common, **{
                "config": {"workload": (f"{code.name} BP4-{iters[0]} alone (one QLDPCBPDecoder launch per step), " if len(iters) == 1 else
                                        f"{code.name} sandwich BP4-{'+'.join(map(str, iters))} with {len(iters) - 1} feedback-GNN "
                                        f"pass(es), trained weights {wname}, ") +
                                       f"cn_type={args.cn_type}, normalization_factor={args.factor} ({ref_construction}), p0={args.p0}, "
                                       f"depolarizing p={args.p}, "
                                       f"noise+syndrome+decode+residual+count on device (BASELINE.json "
                                       f"{cfg['baseline'] if is_cfg_shape else 'shape given on the command line'})",
                           "code": code.name, "batch_per_gpu": B, "global_batch": world * B, "bp_iters": iters, "p": args.p,
                           "cn_type": args.cn_type, "normalization_factor": args.factor, "p0": args.p0, "streams": args.streams,
                           "parallelism": f"batch-sharded x{world}, no data-path collective",
                           "threads_per_codeword": info["threads_per_codeword"], "seed": SEED,
                           "gnn_association": "factored" if factored else "literal",
                           "gnn_kernel": "streaming VALU" if stream else "MFMA tiles",
                           "bp4_qubit_update_lse": "shared per qubit side" if shared_lse else "per edge (literal)"},
                "per_rank_ms": per_rank_ms,
                "roofline": sandwich_roofline(args.code, (n, m, E), launches, launches_per_step, B, iters, factored, stream, shared_lse,
                                              args.cn_type),
                "counts": counts_obj})
        headline = {"value": value, "ms_per_step": elapsed / K * 1e3, "unit": "codewords/s", "steps": K}
        if headline_is_literal:
            out["literal_forms"] = dict(headline, what="the headline of this run IS the literal forms: the reference's formulas term by term are "
                                                       "the library default (`value` repeated under the key of earlier rounds)")
        else:  # an A/B run with FGNN_BENCH_BP4_LSE=shared / FGNN_BENCH_GNN_ORDER=factored
            out["literal_forms"] = other_forms
        if args.streams > 1 and not is_c5:
            # launches of consecutive batches overlap on the chip, so a HIP-event bracket around one launch also covers its neighbour's
            # work: the per-launch durations (and every fraction priced with them) are not this kernel's own — say so instead of quoting them
            r = out["roofline"]
            for k in ("frac", "achieved", "frac_of_hw_transcendental_rate", "effective_bandwidth_frac", "effective_bandwidth_GBs", "hbm_frac"):
                r[k] = None
            r["contract_hbm"]["achieved"] = r["contract_hbm"]["frac"] = None
            r["traffic_source"] = (f"--streams {args.streams}: per-launch HIP-event durations overlap between streams; the roofline fractions are "
                                   "quoted for one-stream runs only")
            if r.get("gnn"):
                r["gnn"]["frac"] = r["gnn"]["achieved"] = None
        out.update(cpu_out)
        out["dist"] = dist_info
        if other_streams is not None:
            out["two_streams" if other_streams["streams"] == 2 else "one_stream"] = other_streams
        out["value_is"] = ("the reference's formulas term by term (literal forms), the library's default operation sequence: one log-sum-exp per edge "
                           "(decoding_q.py:254-273), one Dense per edge (feedback_gnn.py:175-184 / gnn.py:573-610); extras.reassociated_forms is the "
                           "same step under the two opt-in re-associations" if headline_is_literal else
                           "AN A/B RUN, not the library default: the opt-in re-associated forms named in `config` (FGNN_BENCH_BP4_LSE / "
                           "FGNN_BENCH_GNN_ORDER); literal_forms.value is the same step with the reference's formulas term by term")
        # ---- multi-GPU proof of the sharding (no 8-GPU node needed to test it: gloo world 2-3 in tests/test_bench_contract.py): every rank's
        # timed sample range and own counters; the per-step ranges of all ranks must tile [W*world*B, (W+K)*world*B) without overlap, and the
        # ranks' own counters must add up to the all-reduced ones
        for row in out["dist"]["ranks"]:
            rr = [r for r in rank_rows if r[0] == row["rank"]][0]
            row["timed_samples"] = [int(rr[1]), int(rr[2])]
            row["own_counts"] = {"flagged": int(rr[3]), "block_errors": int(rr[4]), "samples": int(rr[5])}
        out["dist"]["sharding"], sharding_ok = sharding_report(step_ranges, rank_rows, W, K, world, B, cnt)

    fa = None
    if rank == 0 and world == 1 and not args.no_literal:
        # ---- per-sample agreement of the opt-in re-associated forms with the literal ones on the first batch of the timed region ----
        first_timed = W * world * B  # rank 0's first timed batch starts at global sample W * world * B
        ex, ez = g.pauli_noise(SEED, args.p, first_timed, B)
        sx, sz = g.syndrome(ex, ez)
        if is_c5:
            fa = gnnbp4_forms_agreement(g, wdev, sx, sz, iters[0], ws)
        else:
            fa = g.forms_agreement(sx, sz, iters, [G.device_weights] * (len(iters) - 1), llr_const(args.p0),
                                   factors=[args.factor] * len(iters), cn_types=[args.cn_type] * len(iters))
        fa["p"] = args.p
        fa["first_sample"] = first_timed
        fa["what"] = ("the first timed batch of rank 0 decoded under the opt-in re-associated forms and under the literal forms (both this "
                      "library's kernels, each bit-equal to the oracle's restatement of its form): samples whose final decisions differ, max over "
                      "samples of max |dLLR| of the last decoder's marginals, samples beyond the north-star tolerance 1e-4"
                      + ("; decisions_differ_beyond_llr_tolerance = samples with a qubit that decides differently although its two best "
                         "candidates of argmin(0, X, Z, Y) are more than 2e-4 apart (the seeded, untrained weights leave many marginals within "
                         "1e-5 of the decision boundary, where a 1e-6 rounding difference flips the argmin)" if is_c5 else "; *_solved = over the samples neither form leaves flagged (a sample BP does not converge on is "
                                          "chaotic under any change of float32 rounding); first_decoder = the BP4-64 launch alone"))

    # ---- extras and the equality check of the CPU sample: rank 0, single-GPU runs only, bounded time, all GPU work ----
    if rank == 0 and world == 1:
        if cpu_check is not None:
            S, o, fl = cpu_check
            if is_c5:
                ex, ez = g.pauli_noise(SEED, args.p, 0, S)
                sx, sz = g.syndrome(ex, ez)
                d = g.gnn_bp4_decode(wdev, sx, sz, iters[0], return_logits=False)
                d["noise_x"], d["noise_z"] = ex, ez
            else:
                m2 = F.Sandwich_BP_GNN_Evaluation_Model(code, decs, [G] * (len(iters) - 1), num_layers=len(iters), p0=args.p0, seed=SEED)
                d = m2.decode(S, args.p, first_sample=0)
            _, _, gfl = g.residual(d["noise_x"], d["noise_z"], d["x_hat"], d["z_hat"], want_arrays=False)
            same = bool(np.array_equal(fl, gfl.cpu().numpy()) and np.array_equal(o["x_hat"], d["x_hat"].cpu().numpy())
                        and np.array_equal(o["z_hat"], d["z_hat"].cpu().numpy()))
            if is_c5:
                same = same and bool(np.array_equal(o["llr"], d["llr"].cpu().numpy()))
            out["cpu_baseline"]["gpu_matches_oracle_bit_exact"] = same
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
        if "cpu_baseline_tf_like" in out:
            out["speedup_vs_cpu_tf_like"] = value / out["cpu_baseline_tf_like"]["value"]
        if not args.no_extras and not is_c5:
            out["extras"] = sandwich_extras(g, model, code, decs, G, iters, B, args.p, SEED, args.cn_type, args.factor, args.p0)
        elif not args.no_extras:
            out["extras"] = {}
        if "extras" in out:
            # the opt-in re-associated forms (NOT the reference's operation sequence) next to the other opt-in variants, with their agreement
            if headline_is_literal and other_forms is not None:
                out["extras"]["reassociated_forms"] = dict(other_forms, forms_agreement=fa, speedup_vs_value=other_forms["value"] / value)
            elif fa is not None:
                out["extras"]["forms_agreement (this A/B run's re-associated headline vs the literal forms)"] = fa
        elif other_forms is not None and headline_is_literal:  # --no-extras: keep the figure, under its own key
            out["reassociated_forms"] = dict(other_forms, forms_agreement=fa)
        elif args.no_literal:
            out["reassociated_forms"] = None  # --no-literal / --no-other-forms: not timed, and the line says so
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        # rank 0 alone ran forms_agreement after the timed regions: the others wait for it here, so that no rank tears its
        # communicator down while a peer is still inside the job
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()
    if rank == 0 and not sharding_ok:
        sys.stderr.write(f"bench.py: the ranks' timed sample ranges do not tile the timed region or their counters do not add up: {out['dist']['sharding']}\n")
        sys.exit(8)
    if rank == 0 and args.require_roofline and out["roofline"].get("frac") is None:
        sys.stderr.write(f"bench.py: --require-roofline: roofline.frac is null ({out['roofline'].get('traffic_source')})\n")
        sys.exit(5)


if __name__ == "__main__":
    main()
