/* fgnn.h — C ABI of the MI355X-native BP4 + feedback-GNN decoder (libfgnn_hip.so).
 *
 * The reference (gongaa/Feedback-GNN, a Sionna/TensorFlow fork) has no FFI: its boundary is the
 * Keras-Layer call contract.  This header is the C boundary placed directly beneath the Python
 * classes that keep the reference's names (feedback_gnn_amd/decoding_q.py, feedback_gnn.py).
 * Each entry point cites the reference code it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C types only; every data pointer is a DEVICE pointer (HIP) unless marked "host";
 *   - the caller owns all buffers; the library owns only the immutable tables inside fgnn_graph /
 *     fgnn_weights; no allocation, no synchronisation on the hot path: all work is enqueued on
 *     `stream` (a hipStream_t passed as void*);
 *   - every per-codeword array is codeword-major ("batch first"), contiguous:
 *       llr      float32 [B,3,n]   planes (x,y,z): L_E = log(P(I)/P(E)), positive = no error
 *       synd_x   uint8   [B,m_x]   hx * noise_z mod 2          synd_z uint8 [B,m_z]  hz * noise_x mod 2
 *       x_hat    uint8   [B,n]     z_hat uint8 [B,n]
 *       messages float32 [B,E]     canonical edge order = sorted by (qubit, check)
 *   - return value 0 = OK, negative = error; text via fgnn_last_error().
 */
#ifndef FGNN_H
#define FGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fgnn_graph fgnn_graph;     /* Tanner graphs + row sets of one CSS code, on one device */
typedef struct fgnn_weights fgnn_weights; /* one feedback GNN's 3 923 parameters, on one device */
typedef struct fgnn_gnnbp4_weights fgnn_gnnbp4_weights; /* parameters of one GNN_BP4 decoder, on one device */

enum { FGNN_CN_BOXPLUS = 0, FGNN_CN_BOXPLUS_PHI = 1, FGNN_CN_MINSUM = 2 }; /* decoding_q.py:97-107 */
enum { FGNN_ROWS_X_LOGIT = 0, FGNN_ROWS_Z_LOGIT = 1, FGNN_ROWS_HX_PERP = 2, FGNN_ROWS_HZ_PERP = 3, FGNN_ROWS_LX = 4, FGNN_ROWS_LZ = 5 };

enum {
    FGNN_OK = 0,
    FGNN_ERR_ARG = -1,     /* shape / value rejected (the Python shim raises ValueError) */
    FGNN_ERR_HIP = -2,     /* a HIP runtime call failed */
    FGNN_ERR_STATE = -3,   /* e.g. row set missing */
    FGNN_ERR_NOMEM = -4
};

const char* fgnn_last_error(void);
int fgnn_version(void); /* 2 = this header (round 3: trace entry point, general GNN_BP4, options 4 and 5 default on) */

/* QLDPCBPDecoder.__init__ edge tables, decoding_q.py:53-94: built here from COO lists of hx and hz
 * (host pointers, any order).  `device` = HIP device ordinal. */
int fgnn_graph_create(int n, int m_x, int m_z, int nnz_x, const int32_t* chk_x, const int32_t* var_x, int nnz_z,
                      const int32_t* chk_z, const int32_t* var_z, int device, fgnn_graph** out);
void fgnn_graph_destroy(fgnn_graph* g);

/* Extra binary row sets as COO (host pointers):
 *   FGNN_ROWS_X_LOGIT / Z_LOGIT : pcm_x_perp / pcm_z_perp of decoding_q.py:33-37,93-94
 *                                 (hz / hx when stage_one or stage_two, else hx_perp / hz_perp);
 *   FGNN_ROWS_HX_PERP / HZ_PERP : code.hx_perp / code.hz_perp for the residual check,
 *                                 feedback_gnn.py:352-353;
 *   FGNN_ROWS_LX / LZ           : code.lx / code.lz, the logical rows appended to the soft syndromes of
 *                                 GNN_BP4.cal_logit, gnn.py:305-313. */
int fgnn_graph_set_rows(fgnn_graph* g, int which, int rows, int nnz, const int32_t* row, const int32_t* col);

/* Launch geometry of the LDS-resident kernels: threads per codeword and codewords per workgroup.
 * 0 = keep the built-in heuristic.  (No reference equivalent: XLA picks its own launch shapes.) */
int fgnn_graph_set_launch(fgnn_graph* g, int threads_per_codeword, int codewords_per_block);
/* Options.  FGNN_OPT_SATURATION_SHORTCUT (default 1): the BP4 kernels (compile-time and runtime degrees) skip the exp/log evaluation
 * of a wave whose 64 nodes are all saturated (|v->c| >= 16.635532 at a check; totals beyond the softplus threshold
 * and 20 apart at a qubit), writing the values those evaluations produce bit for bit (phi(clip max) = 0,
 * phi(clip min), log(1) = 0); and a decode from zero messages and one constant channel LLR (the first decoder of a sandwich) on a
 * degree-regular graph starts from the closed form of its first iteration (every qubit sends the same value, a check's outputs
 * differ by the syndrome sign alone: the same float operations, evaluated once per thread).  Results are identical with 0 and 1; 0 evaluates every transcendental like the
 * reference's fixed dataflow (decoding_q.py:732-767) and is what bench.py's headline number uses.
 * FGNN_OPT_FIXED_POINT_EXIT (default 1, effective only with the shortcut on, boxplus-phi, one codeword per workgroup,
 * at most 32 edges per qubit):
 * when two consecutive check-node phases were all-saturated and no c->v sign changed in between, the messages are at a
 * bit-exact fixed point of the (deterministic) iteration map, every remaining iteration is the identity, and the
 * workgroup leaves the loop.  Results are identical with 0 and 1.
 * FGNN_OPT_HW_TRANSCENDENTALS (default 0; opt-in, NOT bit-exact, never used by a parity test or by bench.py's headline):
 * boxplus-phi decodes evaluate exp / log on the hardware's v_exp_f32 / v_log_f32 units instead of the shared float32 routines
 * of fgnn_math.h, in the fixed dataflow (the two exact options above are proofs about fgnn_math.h and are ignored while this is
 * set).  Same TensorFlow op structure (softplus thresholds, max-shifted log-sum-exp, _phi clip; _phi's two clip points pinned to
 * 0 and 16.635532, the values the reference's saturation known answer fixes), ~1 ulp per elementary function, but bits that no
 * CPU oracle reproduces: on non-converged samples decisions may differ from the default path's (chaotic
 * transients).  bench.py reports its rate and its measured agreement with the exact kernel under `extras`.
 * FGNN_OPT_GNN_FACTORED and FGNN_OPT_BP4_SHARED_LSE (both default 0 since round 6; OPT-IN re-associations of the reference's own formulas, same
 * real-number functions, NOT the reference's float32 operation sequence).  The library's default evaluates the reference's formulas term
 * by term — one 40 -> 20 Dense per edge (feedback_gnn.py:175-184), one reduce_logsumexp per edge (decoding_q.py:254-273) — and that is what
 * bench.py's `value`, smoke() and every default parity test run.
 * FGNN_OPT_GNN_FACTORED = 1: the feedback GNN's message MLP and mean (feedback_gnn.py:175-184) in the factored association:
 * [g, X, Y, Z] W1 + b1 = g W1[0,:] + ([X, Y, Z] W1[1:4,:] + b1) with the bracket formed once per qubit and side, and
 * mean_e(h_e W2 + b2) = (sum_e h_e) W2 / deg + b2 with ONE 40 -> 20 Dense per qubit and side instead of one per edge.  float32 results
 * differ from the literal association's by rounding only (measured <= 5e-7 on GNN outputs of magnitude 0.2 .. 2.7).  Applies to the kernels
 * of the shipped architecture (fgnn_weights_create); the runtime-shaped kernel (fgnn_weights_create_general, any reduce_op) always runs the
 * literal order.
 * FGNN_OPT_BP4_SHARED_LSE = 1: the variable-node update (decoding_q.py:254-273) evaluates, on every hx edge e of a qubit,
 * reduce_logsumexp([-(Z - mu_e), -(Y - mu_e)]) = max(.,.) + log(1 + exp(-|(Z - mu_e) - (Y - mu_e)|)), and the last term's argument is
 * Z - Y for all of the qubit's edges: with the option on it is formed once per qubit and side from the unshifted totals (the per-edge
 * max term stays as it is) — 4 instead of 8 exp/log pairs per qubit and iteration.  A v->c message moves by the rounding of two
 * subtractions, and BP's transient amplifies that like any other float32 rounding.
 * What a caller who turns both on gets, against the default, per sample through the whole sandwich (these are MEASURED RATES, not
 * guarantees; bench.py prints the same comparison for its timed batch under extras.reassociated_forms.forms_agreement):
 *     north-star bar = decisions identical and LLRs within 1e-4.  [[882,24]] (64, G, 16), p = 0.01, 65 536 samples per window: 0 samples
 *     with a different decision in most windows, but NOT in all: the driver's round-5 batch (global samples 327 680 - 393 215) has one
 *     SOLVED sample at 1.03e-4, and 8.4 M samples hold 3 differing decisions and 40 solved samples beyond 1e-4 (up to 101)
 *     (profiles/r4_forms_agreement_8M.json); p = 0.02: 40 / 8.4 M decisions differ; p = 0.03: 2 of 65 536; 0.05: 74 (0.11 %); 0.06: 344;
 *     0.08: 2 746 (4.2 %); 0.10: 9 611 (14.7 %) end on a different — equally valid or equally failed — estimate.
 *     [[1270,28]] (64, G, 64): p = 0.01: 6 / 8.4 M; 0.03: 4 of 65 536; 0.05: 126; 0.06: 426; 0.08: 4 539; 0.10: 15 177 (23 %).
 *     BASELINE configs[0] as the reference constructs it ('boxplus', 0.625, p = 0.05, 256 samples): 0 decisions differ but 250 of the 256
 *     solved samples are beyond 1e-4 (max 0.575): the 'boxplus' rule does not saturate its messages, so rounding differences persist.
 *     Published workloads (batch 5 000 / 50 000): n1270_3r (p = 0.07) 144 of 5 000 decisions differ; osd_bp4_minsum (p = 0.09) 1 497 of 50 000.
 * So the re-associated forms are the reference's decoder only STATISTICALLY: BP4-64 decodes the same number of samples (24 M compared
 * at p = 0.06 .. 0.10 on both codes: differences within 1.8 sigma, both signs, profiles/r3j_bp4_shared_lse_ab.txt), paired block-error
 * counts on 40 M samples agree (profiles/r3v_bp4_lse_forms_mcnemar.json) and the 77 published rows of the nine sandwich curves land on
 * the same z-scores (max |z| 1.89, profiles/r3k_published_curves_bp4_shared_lse.json).  They buy ~1.15x on the sandwich step.  The oracle
 * restates both forms (og_graph_set_gnn_order, og_graph_set_vn_shared_lse); the kernels equal it bit for bit in either.
 * FGNN_OPT_GNN_STREAM (default 1): which kernel runs the feedback GNN (either association) on a graph with 3, 4 or 5 checks per qubit and
 * side — the streaming VALU kernel (one lane per qubit, weights as scalar operands; the literal association's 40 -> 20 Dense per edge as
 * v_pk_fma_f32 on scalar weight pairs) or the MFMA-tile kernel ((3,3) only; other degrees: the runtime-degree kernel).  0: never the
 * streaming kernel; 1: wherever it is the faster one (from 4 096 codewords per launch on; smaller launches are latency-bound and
 * quicker on the MFMA tiles); 2: always.  The same float operations in the same
 * order: results are bit-identical; the option exists for A/B timing and tests.  No effect on irregular graphs.
 * Value 2 ALSO moves fgnn_gnnbp4_decode on a (3,3,6)-regular graph from its MFMA-tile kernel to its streaming packed-FMA kernel — the
 * tested second implementation of the same float operations, bit-identical and SLOWER (180 ms against 134 ms per 16 384 x 10
 * iterations of [[1270,28]], profiles/r4_gnnbp4_stream_ab.txt); values 0 and 1 leave GNN_BP4 on the MFMA tiles.  A caller who wants
 * "always" for the feedback GNN and times GNN_BP4 on the same graph handle should set the option back to 1 around GNN_BP4 calls. */
enum { FGNN_OPT_SATURATION_SHORTCUT = 1, FGNN_OPT_FIXED_POINT_EXIT = 2, FGNN_OPT_HW_TRANSCENDENTALS = 3, FGNN_OPT_GNN_FACTORED = 4,
       FGNN_OPT_BP4_SHARED_LSE = 5, FGNN_OPT_GNN_STREAM = 6 };
int fgnn_graph_set_option(fgnn_graph* g, int option, int value);
/* Testing hook: on != 0 forces the runtime-degree (CSR) kernel even on a degree-regular graph. */
int fgnn_graph_force_generic(fgnn_graph* g, int on);
/* info[0..9] = n, m_x, m_z, E_x, E_z, threads_per_codeword, codewords_per_block, lds_bytes_per_block,
 *              regular(0/1), device */
int fgnn_graph_info(const fgnn_graph* g, int32_t info[16]);
/* canonical (qubit, check)-sorted edge lists, host output: chk[E_s], var[E_s] for side 0 (hx) / 1 (hz) */
int fgnn_graph_edges(const fgnn_graph* g, int side, int32_t* chk, int32_t* var);

/* Per-launch timing of the BP4 and feedback-GNN kernels (no reference equivalent; sim_ber only has wall-clock per point,
 * misc.py:639,696): HIP events are recorded on the launch stream right before and after every BP4 / feedback-GNN launch,
 * standalone or inside fgnn_sandwich_decode, up to max_launches (0 disables).  fgnn_profile_read waits for the
 * last one and returns ms / tag / B per launch in host arrays of length cap; tag = num_iter of a BP4 launch,
 * -1 for a feedback-GNN launch, -2 for a GNN_BP4 launch (fgnn_gnnbp4_decode). */
int fgnn_profile_enable(fgnn_graph* g, int max_launches);
int fgnn_profile_read(fgnn_graph* g, float* ms, int32_t* iters, int32_t* batch, int cap, int32_t* count);

/* QLDPCBPDecoder.call, decoding_q.py:661-797 (flooding BP4, num_iter iterations, then marginals,
 * hard decision :783-790 and soft syndrome cal_logit :455-471).
 *   llr_ch       [B,3,n] or NULL: all three channel LLRs = llr_const (feedback_gnn.py:311-313)
 *   msg_init_*   [B,E_x]/[B,E_z] or NULL (= zeros, :726-727)        msg_out_* optional
 *   llr_out      [B,3,n] marginals (llrx, llry, llrz of :777)       x_hat,z_hat [B,n]
 *   x_logit      [B,rows(X_LOGIT)]  z_logit [B,rows(Z_LOGIT)]  (either may be NULL)
 */
int fgnn_bp4_decode(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                    float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* msg_init_x,
                    const float* msg_init_z, float* llr_out, uint8_t* x_hat, uint8_t* z_hat, float* x_logit,
                    float* z_logit, float* msg_out_x, float* msg_out_z, void* stream);

/* The `trainable` / `stage_two` return mode of QLDPCBPDecoder.call (decoding_q.py:730, 743-746, 779-781, 794-795) in ONE launch:
 * the soft syndromes cal_logit (:455-471) of the marginals after 0, 1, ..., num_iter iterations,
 *   x_logit_trace [num_iter+1, B, rows(X_LOGIT)],  z_logit_trace [num_iter+1, B, rows(Z_LOGIT)]
 * (the reference's llr_hat[2k] / llr_hat[2k+1], transposed), plus the usual marginals and decisions of the last iteration.
 * tape_x [num_iter+1, B, E_x] / tape_z [num_iter+1, B, E_z] (both or neither): the c->v messages before iteration k (slot num_iter:
 * after the last) — the tape fgnn_bp4_backward reads.  Fixed dataflow on the shared float32 routines; every value equals what
 * num_iter + 1 chained fgnn_bp4_decode calls (0, 1, 1, ... iterations through msg_init / msg_out) produce, bit for bit. */
int fgnn_bp4_decode_trace(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                          float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* msg_init_x,
                          const float* msg_init_z, float* llr_out, uint8_t* x_hat, uint8_t* z_hat, float* x_logit_trace,
                          float* z_logit_trace, float* tape_x, float* tape_z, void* stream);

/* load_weights, gnn.py:774-791: 12 host arrays in the reference's file order
 * [W_out(40,3), b_out(3), Wx1(4,40), bx1(40), Wx2(40,20), bx2(20), Wz1, bz1, Wz2, bz2, We(43,40), be(40)]. */
int fgnn_weights_create(const float* const host_arrays[12], int device, fgnn_weights** out);
void fgnn_weights_destroy(fgnn_weights* w);

/* Feedback_GNN with other hyper-parameters than the shipped weights' (constructor feedback_gnn.py:21-28, build :110-128):
 * num_msg_dims D <= 32, num_hidden_units H <= 96, num_mlp_layers L in 1..4, reduce_op (:139-148), the activation of the
 * hidden layers, use_bias.  host_arrays = Layer.get_weights() order: _llr_inv_embed [(L>1 ? H : 2D+3), 3] (+ bias),
 * vn_msg_mlp_x: L Dense layers 4 -> H -> ... -> D, vn_msg_mlp_z: likewise, vn_embed_mlp: L-1 Dense layers 2D+3 -> H -> ... -> H;
 * every kernel [in, out] row-major, a bias array after each kernel iff use_bias.  num_arrays must equal
 * (3L) * (1 + use_bias).  The handle is accepted wherever fgnn_weights is (fgnn_feedback_gnn, fgnn_sandwich_decode); it runs
 * a runtime-shaped VALU kernel (the MFMA kernel is specialised for D=20, H=40, L=2, mean, tanh, bias). */
typedef struct {
    int num_msg_dims, num_hidden_units, num_mlp_layers;
    int reduce_op;   /* FGNN_REDUCE_* */
    int activation;  /* FGNN_ACT_* */
    int use_bias;
} fgnn_gnn_config;
enum { FGNN_REDUCE_SUM = 0, FGNN_REDUCE_MEAN = 1, FGNN_REDUCE_MAX = 2, FGNN_REDUCE_MIN = 3 };
enum { FGNN_ACT_LINEAR = 0, FGNN_ACT_TANH = 1, FGNN_ACT_RELU = 2, FGNN_ACT_SIGMOID = 3 };
int fgnn_weights_create_general(const fgnn_gnn_config* cfg, const float* const* host_arrays, int num_arrays, int device,
                                fgnn_weights** out);

/* Feedback_GNN.call, feedback_gnn.py:161-188 (reduce_op="mean", activation="tanh", 2-layer MLPs):
 *   llr [B,3,n] = h_vn planes (llrx, llry, llrz);  logit_hx [B,m_x], logit_hz [B,m_z] soft syndromes
 *   of the hx / hz rows;  out [B,3,n] = new channel LLRs for the next BP run. */
int fgnn_feedback_gnn(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                      const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B, float* out,
                      void* stream);

/* Pauli.call (non-wt branch), sionna/channel/pauli.py:98-108 with px=pz=2p/3, py=p/3
 * (feedback_gnn.py:298): Philox4x32-10 stream keyed by (seed, first_sample + b). */
int fgnn_pauli_noise(uint64_t seed, float p, uint64_t first_sample, int B, int n, uint8_t* noise_x, uint8_t* noise_z,
                     void* stream);
/* Pauli.call (non-wt branch) for ANY triple, sionna/channel/pauli.py:98-108: noise_x = u < px, noise_z = (u >= px - py) & (u < (px + pz) - py)
 * in float32, u the same Philox uniform as fgnn_pauli_noise (which is this call with px = pz = 2p/3, py = p/3 formed in float32).  Like
 * the reference it validates nothing but NaNs: X, Y, Z are disjoint events of probability px - py, py, pz - py when
 * 0 <= py <= min(px, pz) and px + pz - py <= 1. */
int fgnn_pauli_noise_xyz(uint64_t seed, float px, float py, float pz, uint64_t first_sample, int B, int n, uint8_t* noise_x,
                         uint8_t* noise_z, void* stream);
/* fgnn_pauli_noise with the stream position read on the DEVICE: sample b is keyed by (*first_sample_dev + offset + b).  For Monte-Carlo loops
 * captured in a hipGraph (the reference's `while` loop of misc.py:636-738 with no host in it): the graph owns an 8-byte device counter,
 * every noise launch of the captured loop reads it, and the graph's last node advances it, so each replay draws the next samples. */
int fgnn_pauli_noise_dev(uint64_t seed, float p, const uint64_t* first_sample_dev, uint64_t offset, int B, int n, uint8_t* noise_x,
                         uint8_t* noise_z, void* stream);
/* Pauli.call with wt=True, pauli.py:80-97 (training-set harvesting, Generate_dataset.ipynb): exactly `wt` qubits per sample
 * carry an error, X / Y / Z with probability 1/3 each; positions by a Philox-driven partial Fisher-Yates shuffle. */
int fgnn_pauli_noise_wt(uint64_t seed, int wt, uint64_t first_sample, int B, int n, uint8_t* noise_x, uint8_t* noise_z, void* stream);
/* syndrome_x = hx noise_z, syndrome_z = hz noise_x (mod 2), feedback_gnn.py:305-309. */
int fgnn_syndrome(const fgnn_graph* g, const uint8_t* noise_x, const uint8_t* noise_z, int B, uint8_t* synd_x,
                  uint8_t* synd_z, void* stream);
/* errors[b] &= (syndrome of (x_hat,z_hat) != (synd_z, synd_x)), feedback_gnn.py:324-330. */
int fgnn_flag_update(const fgnn_graph* g, const uint8_t* x_hat, const uint8_t* z_hat, const uint8_t* synd_x,
                     const uint8_t* synd_z, int B, uint8_t* errors, void* stream);
/* x_hat[b], z_hat[b] <- x_upd[b], z_upd[b] where errors[b], feedback_gnn.py:339-340. */
int fgnn_merge(const uint8_t* errors, const uint8_t* x_upd, const uint8_t* z_upd, int B, int n, uint8_t* x_hat,
               uint8_t* z_hat, void* stream);
/* Residual check, feedback_gnn.py:343-361: s_hat [B,m_z+m_x] = [hz xd ; hx zd], ls_hat
 * [B,rows(hx_perp)+rows(hz_perp)] (either may be NULL), flags[b] bit0 = any(s_hat), bit1 = any(ls_hat)
 * (= what count_block_errors sees, sionna/utils/metrics.py:221-223 via misc.py:649-651). */
int fgnn_residual(const fgnn_graph* g, const uint8_t* noise_x, const uint8_t* noise_z, const uint8_t* x_hat,
                  const uint8_t* z_hat, int B, uint8_t* s_hat, uint8_t* ls_hat, uint8_t* flags, void* stream);
/* Same with the two row sets chosen by the caller: ls_hat = [rows_x . xd ; rows_z . zd], e.g. (FGNN_ROWS_LZ, FGNN_ROWS_LX) for
 * BP4_OSD_Model.call, bp_osd.py:184-188. */
int fgnn_residual_rows(const fgnn_graph* g, int rows_x, int rows_z, const uint8_t* noise_x, const uint8_t* noise_z,
                       const uint8_t* x_hat, const uint8_t* z_hat, int B, uint8_t* s_hat, uint8_t* ls_hat, uint8_t* flags, void* stream);
/* Bit-packed decisions for the multi-GPU gather (no reference equivalent: the reference runs one process per GPU id and
 * never collects decisions, n882.py:9,15-21): packed[b] = the 2n bits [x_hat[b,:] | z_hat[b,:]], most significant bit first
 * (numpy.packbits order), ceil(2n/8) bytes per codeword; fgnn_unpack_decisions is the inverse.  Runs on the current device. */
int fgnn_pack_decisions(const uint8_t* x_hat, const uint8_t* z_hat, int B, int n, uint8_t* packed, void* stream);
int fgnn_unpack_decisions(const uint8_t* packed, int B, int n, uint8_t* x_hat, uint8_t* z_hat, void* stream);
/* counts[0] += #flagged, counts[1] += #block errors, counts[2] += B (device uint64[3]), misc.py:649-669. */
int fgnn_count_flags(const uint8_t* flags, int B, uint64_t* counts, void* stream);
/* The same for num_batches consecutive batches of `batch` samples whose flags lie back to back in flags[num_batches * batch] (the
 * batches were decoded as one launch): ring[3 j .. 3 j + 2] = the cumulative counters after batch j, counts = ring of the last batch —
 * what num_batches calls of fgnn_count_flags would have produced one after the other, so the stopping rule of sim_ber
 * (misc.py:700-738) can be applied batch by batch afterwards.  scratch: device uint32[2 * num_batches].  Current device. */
int fgnn_count_flags_batches(const uint8_t* flags, int num_batches, int batch, uint64_t* counts, uint64_t* ring, uint32_t* scratch,
                             void* stream);

/* Sandwich_BP_GNN_Evaluation_Model.call, feedback_gnn.py:293-361, from the syndromes on:
 * decoder 0, then for i = 1..num_layers-1: flag update, GNN i-1, decoder i, masked merge.
 * iters/factors/cn_types: host arrays [num_layers]; weights: host array of num_layers-1 handles.
 * Needs the logit row sets to be (hz, hx) ("stage_one").  compact != 0 runs the GNN/BP rounds only on
 * the samples still in `errors` (same x_hat/z_hat, fewer FLOPs; llr_final then holds the last marginals
 * computed for each sample).  rounds[b] (optional) = rounds in which sample b was still in `errors`. */
size_t fgnn_sandwich_workspace_bytes(const fgnn_graph* g, int B);
int fgnn_sandwich_decode(const fgnn_graph* g, int num_layers, const int32_t* iters, const float* factors,
                         const int32_t* cn_types, const fgnn_weights* const* weights, float llr_const,
                         const uint8_t* synd_x, const uint8_t* synd_z, int B, int compact, uint8_t* x_hat,
                         uint8_t* z_hat, float* llr_final, uint8_t* rounds, void* workspace, size_t ws_bytes,
                         void* stream);

/* Binary syndrome BP — LDPCBPDecoder.call with is_syndrome=True, sionna/fec/ldpc/decoding.py:874-1048 (SURVEY.md §8f
 * rank 1).  The parity-check matrix is side 0 (hx) of the graph.  llr_ch [B,n] are LOGITS as in the reference (clipped
 * to +-20, sign flipped on entry and exit, :918-920,:940,:1031) or NULL (= llr_const everywhere); synd [B,m_x] or NULL
 * (all-zero syndrome = ordinary decoding).  soft_out [B,n] = output logits, hard_out [B,n] = (0 < logit) (:1033-1034). */
int fgnn_bp2_decode(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                    float llr_const, const uint8_t* synd, int B, float* soft_out, uint8_t* hard_out, void* stream);
/* BinarySymmetricChannel on the all-zero word, BP_BSC_Model.call feedback_gnn.py:213-214: noise = u < p (Philox stream). */
int fgnn_bsc_noise(uint64_t seed, float p, uint64_t first_sample, int B, int n, uint8_t* noise, void* stream);

/* OSD-0 post-processing of BP failures — OSD0_Decoder.call / find_mrb, sionna/fec/ldpc/bp_osd.py:14-77, as driven by
 * BP4_OSD_Model.call_osd (:138-157).  fgnn_graph_set_basis installs code.pivot_hx (side 0) or
 * code.pivot_hz (side 1) (host array): the rows hx[pivot_hx] form the full-rank matrix of :147-150.  fgnn_osd0 solves, for
 * every listed sample, H_basis e = syndrome on the most reliable independent columns and writes e_hat[b,:]:
 * side 0 -> z_hat from hx and osd_llrz, side 1 -> x_hat from hz and osd_llrx (:125-131), computed from the BP4 marginals
 * marg [B,3,n], or from llr_bin [B,n] if given (BP2_OSD_Model).  synd [B,m_side] is the FULL syndrome (reduced
 * internally, :144-145).  index (device int32[nact]) selects the samples to process, NULL = all B.
 * Ties in the reliability sort keep qubit order (tf.argsort leaves them unspecified). */
int fgnn_graph_set_basis(fgnn_graph* g, int side, int rank, const int32_t* pivot_rows);
int fgnn_osd0(const fgnn_graph* g, int side, const float* marg, const float* llr_bin, const uint8_t* synd, int B,
              const int32_t* index, int nact, uint8_t* e_hat, void* stream);
/* index[0..*count) = ids b with (mask[b] & bit) != 0 (tf.where(err), bp_osd.py:166-171); *count (device int32) must be
 * zero on entry; ids of one 256-sample block are ascending, blocks arrive in any order. */
int fgnn_compact(const uint8_t* mask, int bit, int B, int32_t* index, int32_t* count, void* stream);

/* GNN_BP4 (the syndrome-only "full GNN" decoder), sionna/fec/ldpc/gnn.py:71-423 with UpdateCNEmbeddings (:426-610)
 * and UpdateVNEmbeddings (:612-751): num_mlp_layers=2, tanh, mean, use_bias, num_embed_dims=20, num_hidden_units=40.
 * Weights: 30 host arrays — cn_msg_x, cn_msg_z, cn_embed_x, cn_embed_z, vn_msg_x, vn_msg_z, vn_embed, each
 * {W1[in,40], b1[40], W2[40,20], b2[20]} with in = 40, 40, 41, 41, 40, 40, 60, then llr_inv {W[20,3], b[3]}.
 * Outputs: x_hat/z_hat [B,n] (make_hard_decision :359-367), llr_out [B,3,n] = last embed_to_llr (:283-289),
 * x_logit_all [num_iter,B,m_z+rows(lz)], z_logit_all [num_iter,B,m_x+rows(lx)] = llr_hat of :409 (may be NULL).
 * The reference's call raises as shipped (:408 unpacks 5 of cal_logit's 4 values); that line is repaired. */
int fgnn_gnnbp4_weights_create(const float* const host_arrays[30], int num_embed_dims, int num_hidden_units, int device,
                               fgnn_gnnbp4_weights** out);
void fgnn_gnnbp4_weights_destroy(fgnn_gnnbp4_weights* w);
size_t fgnn_gnnbp4_workspace_bytes(const fgnn_graph* g, int B);
/* GNN_BP4 with any other constructor setting the reference's classes accept (gnn.py:131-207; UpdateCNEmbeddings :494-610,
 * UpdateVNEmbeddings :640-751): num_embed_dims D <= 32, num_hidden_units H <= 96, num_mlp_layers L in 1..4, reduce_op (:556-568), the
 * activation of the hidden layers, use_bias, use_attributes with node_attribute_dims / msg_attribute_dims <= 16 (trainable per-node
 * and per-edge vectors concatenated to the MLP inputs, :584-599, :724-746).  num_msg_dims has no effect in the reference
 * (`units[-1] = num_embed_dims` rewrites the list the message MLPs share, :548, :690); clip_llr_to, input_embed and the syndrome_embed
 * layers are never used by call.  host_arrays: for each MLP in the order cn_msg_x, cn_msg_z, cn_embed_x, cn_embed_z, vn_msg_x,
 * vn_msg_z, vn_embed its L Dense layers {W[in,out] (, b[out])}, then _llr_inv_embed {W[D,3] (, b[3])}, then with attributes
 * cn_node_x [m_x,An], cn_node_z [m_z,An], cn_msg_x [E_x,Am], cn_msg_z [E_z,Am], vn_node [n,An], vn_msg_x [E_x,Am], vn_msg_z [E_z,Am]
 * with edge rows in the reference's order np.where(pcm) (check-major); MLP inputs [h_from | h_to | msg attr], [m | node attr | h_to |
 * logit], [m_x | m_z | node attr | h_to].  The handle is accepted by fgnn_gnnbp4_decode and runs a runtime-shaped VALU kernel in the
 * literal association (the MFMA kernel is specialised for D = 20, H = 40, L = 2, mean, tanh, bias, no attributes); the workspace
 * it needs is fgnn_gnnbp4_weights_workspace_bytes (which also serves handles of fgnn_gnnbp4_weights_create). */
typedef struct {
    int num_embed_dims, num_hidden_units, num_mlp_layers;
    int reduce_op;   /* FGNN_REDUCE_* */
    int activation;  /* FGNN_ACT_* */
    int use_bias, use_attributes, node_attribute_dims, msg_attribute_dims;
} fgnn_gnnbp4_config;
int fgnn_gnnbp4_weights_create_general(const fgnn_graph* g, const fgnn_gnnbp4_config* cfg, const float* const* host_arrays,
                                       int num_arrays, fgnn_gnnbp4_weights** out);
size_t fgnn_gnnbp4_weights_workspace_bytes(const fgnn_graph* g, const fgnn_gnnbp4_weights* w, int B);
int fgnn_gnnbp4_decode(const fgnn_graph* g, const fgnn_gnnbp4_weights* w, int num_iter, const uint8_t* synd_x,
                       const uint8_t* synd_z, int B, uint8_t* x_hat, uint8_t* z_hat, float* llr_out, float* x_logit_all,
                       float* z_logit_all, void* workspace, size_t ws_bytes, void* stream);

/* ---- training: reverse pass of Second_Stage_GNN_BP_Model (feedback_gnn.py:423-463, train_n882.ipynb cell 7) -------------
 * The reference differentiates GNN -> 16 BP4 iterations (boxplus-phi, stage_two soft syndromes per iteration,
 * decoding_q.py:768-775) -> BCE with tf.GradientTape.  The two entry points below are that chain rule by hand.
 *
 * fgnn_bp4_backward: tape_x/tape_z [T+1,B,E_s] are the c->v messages before iteration k (k = T: after the last one), i.e.
 * the msg_out of T chained one-iteration fgnn_bp4_decode calls, slot 0 all zero.  grad_x_logit [T+1,B,rows0] /
 * grad_z_logit [T+1,B,rows1] are d loss / d (soft syndromes after k iterations) (either may be NULL); has_grad [T+1]
 * (device, u8) flags the k at which a gradient enters.  Output grad_llr_ch [B,3,n] = d loss / d llr_ch.  */
int fgnn_bp4_backward(const fgnn_graph* g, int num_iter, float normalization_factor, const float* llr_ch,
                      const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* tape_x, const float* tape_z,
                      const float* grad_x_logit, const float* grad_z_logit, const uint8_t* has_grad, float* grad_llr_ch,
                      void* stream);

/* fgnn_feedback_gnn_backward: inputs as fgnn_feedback_gnn plus grad_out [B,3,n] = d loss / d (GNN output).  Leaves the
 * (activation, delta) pair of every Dense layer in device memory — node_in [B,n,44] (= [mean_x|mean_z|X,Y,Z|0]),
 * node_h2 / node_d2 [B,n,40], and per side s (0 = hx, 1 = hz) edge_feat[s] [B,E_s,4], edge_h1[s] / edge_d1[s] [B,E_s,40],
 * edge_dm[s] [B,E_s,20] — so that each weight gradient is one plain GEMM activation^T * delta and each bias gradient a
 * column sum (Keras order of get_weights(): Wout = node_h2^T G, We = node_in[:, :43]^T node_d2, W1_s = edge_feat^T edge_d1,
 * W2_s = edge_h1^T edge_dm). */
int fgnn_feedback_gnn_backward(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                               const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B,
                               const float* grad_out, float* node_in, float* node_h2, float* node_d2,
                               float* const edge_feat[2], float* const edge_h1[2], float* const edge_d1[2],
                               float* const edge_dm[2], void* stream);
/* The same for weights made by fgnn_weights_create_general (any constructor setting of Feedback_GNN: the reference trains whatever
 * feedback_gnn.py:21-128 built, :423-463).  For Dense layer li in EXECUTION order — [0, L) vn_msg_mlp_x, [L, 2L) vn_msg_mlp_z,
 * [2L, 3L-1) vn_embed_mlp, 3L-1 _llr_inv_embed; L = num_mlp_layers, num_layers = 3L — the kernel leaves acts[li] [rows, K_li] = the
 * layer's input and deltas[li] [rows, J_li] = d loss / d (its pre-activation), rows = B*E_x / B*E_z (message MLPs, canonical edge
 * order) or B*n; the caller forms W_li grad = acts^T deltas and b_li grad = column sums.  max / min reduce: the gradient goes to the
 * edges attaining the extremum, shared equally among ties. */
int fgnn_feedback_gnn_backward_general(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                                       const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B,
                                       const float* grad_out, float* const* acts, float* const* deltas, int num_layers, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FGNN_H */
