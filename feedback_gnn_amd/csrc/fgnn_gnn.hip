// fgnn_gnn.hip — the feedback GNN: one CN->VN message-passing layer that turns the marginals and
// soft syndromes of a BP run into the channel LLRs of the next one.
//
// Replaces Feedback_GNN.call of /root/reference sionna/fec/ldpc/feedback_gnn.py:161-188 (with
// reduce_msg :130-150 and MLP.call, gnn.py:63-69) for the configuration the reference trains and
// ships: num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean",
// activation="tanh", use_bias=True (n882.py:45-51).
//
// The reference materialises [bs,E,4], [bs,E,40] and [bs,E,20] tensors in HBM (27 GB at
// bs=65 536).  Here nothing but the [B,3,n] input/output and the two soft-syndrome vectors touches HBM.
//
// Two kernels:
//  * gnn_mfma_kernel<DV> (degree-regular graphs, the benchmark codes): one wave owns 16 qubits at a time
//    and runs every Dense layer as v_mfma_f32_16x16x4_f32 on TRANSPOSED problems (H^T = W^T F^T), so that
//    the accumulator layout of one layer (column = qubit on the lane, rows in the 4 registers) is
//    directly the B operand of the next layer — no LDS round trip, no cross-lane moves.  The weight
//    matrices live in registers as A operands for the whole kernel; their rows are permuted at upload
//    so that k-step s of the next layer finds hidden units 4s..4s+3 on lane groups 0..3.
//  * gnn_kernel (any graph): one thread per qubit, weights by scalar loads, plain v_fma chains.
//
// Arithmetic order = oracle/fgnn_oracle.c (gnn_edge_side / gnn_one): every dot product is an fmaf
// chain in ascending k starting from 0, then + bias; edge messages are summed in ascending check
// order and divided by the degree.
#include <cstring>

#include "fgnn_internal.h"
#include "fgnn_math.h"
#include "fgnn_pk.h"

namespace {

constexpr int HID = 40;
constexpr int MSG = 20;

struct GnnArgs {
    int B, tpc, cpb, lds_per_cw;
    const float* llr;       // [B,3,n]
    const float* logit_hx;  // [B,m_x]
    const float* logit_hz;  // [B,m_z]
    const uint8_t* synd_x;
    const uint8_t* synd_z;
    float* out;             // [B,3,n]
    const int* index;       // optional: workgroup slot -> sample
    int nsplit;             // MFMA kernel: a codeword's tiles are dealt to nsplit groups of four waves (small batches: latency)
};

// vn_msg_mlp_{x,z} on one edge: feature [g, X, Y, Z] -> Dense(40,tanh) -> Dense(20)  (:175-181)
__device__ __forceinline__ void edge_mlp(float gv, float X, float Y, float Z, scalar_fp w1t, scalar_fp b1, scalar_fp w2, scalar_fp b2,
                                         float (&msg)[MSG])
{
#pragma unroll
    for (int i = 0; i < MSG; ++i) msg[i] = 0.0f;
#pragma unroll 2
    for (int j = 0; j < HID; ++j) {
        scalar_fp r = w1t + j * 4;
        float a = 0.0f;
        a = FG_FMA(gv, r[0], a);
        a = FG_FMA(X, r[1], a);
        a = FG_FMA(Y, r[2], a);
        a = FG_FMA(Z, r[3], a);
        const float h = fg_tanh(a + b1[j]);
        scalar_fp r2 = w2 + j * MSG;
#pragma unroll
        for (int i = 0; i < MSG; ++i) msg[i] = FG_FMA(h, r2[i], msg[i]);
    }
#pragma unroll
    for (int i = 0; i < MSG; ++i) msg[i] = msg[i] + b2[i];
}

// mean over the qubit's edges of one side (:183-184, reduce_msg :139-141)
__device__ __forceinline__ void side_mean(const GraphDev& g, const int* __restrict__ vptr, int v, const float* gcn,
                                          float X, float Y, float Z, scalar_fp w1t, scalar_fp b1, scalar_fp w2, scalar_fp b2,
                                          float (&mean)[MSG])
{
    const int e0 = vptr[v], e1 = vptr[v + 1];
    float msg[MSG];
#pragma unroll
    for (int i = 0; i < MSG; ++i) mean[i] = 0.0f;
    for (int e = e0; e < e1; ++e) {
        edge_mlp(gcn[g.vchk[e]], X, Y, Z, w1t, b1, w2, b2, msg);
#pragma unroll
        for (int i = 0; i < MSG; ++i) mean[i] = (e == e0) ? msg[i] : mean[i] + msg[i];
    }
    const int deg = e1 - e0;
    if (deg > 0) {
        const float fd = (float)deg;
#pragma unroll
        for (int i = 0; i < MSG; ++i) mean[i] = mean[i] / fd;
    }
}

// The same in the FACTORED association (FGNN_OPT_GNN_FACTORED, oracle: gnn_edge_side_factored): the X/Y/Z part of the first Dense and
// its bias are formed once per side, each edge adds g W1[0,j] with one fma; the hidden activations are summed over the edges and ONE
// last Dense is applied to the sum, then / deg, then + b2.
__device__ __forceinline__ void side_mean_factored(const GraphDev& g, const int* __restrict__ vptr, int v, const float* gcn,
                                                   float X, float Y, float Z, scalar_fp w1t, scalar_fp b1, scalar_fp w2, scalar_fp b2,
                                                   float (&mean)[MSG])
{
    const int e0 = vptr[v], e1 = vptr[v + 1];
#pragma unroll
    for (int i = 0; i < MSG; ++i) mean[i] = 0.0f;
#pragma unroll 2
    for (int j = 0; j < HID; ++j) {
        scalar_fp r = w1t + j * 4;
        float a = 0.0f;
        a = FG_FMA(X, r[1], a);
        a = FG_FMA(Y, r[2], a);
        a = FG_FMA(Z, r[3], a);
        const float pb = a + b1[j];
        float hs = 0.0f;
        for (int e = e0; e < e1; ++e) {
            const float h = fg_tanh(FG_FMA(gcn[g.vchk[e]], r[0], pb));
            hs = (e == e0) ? h : hs + h;
        }
        scalar_fp r2 = w2 + j * MSG;
#pragma unroll
        for (int i = 0; i < MSG; ++i) mean[i] = FG_FMA(hs, r2[i], mean[i]);
    }
    const int deg = e1 - e0;
    if (deg > 0) {
        const float fd = (float)deg;
#pragma unroll
        for (int i = 0; i < MSG; ++i) mean[i] = mean[i] / fd + b2[i];
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA path.  v_mfma_f32_16x16x4_f32: A[16x4] (lane l holds A[l&15][l>>4]), B[4x16] (lane l holds
// B[l>>4][l&15]), C/D[16x16] (lane l, register r holds D[4*(l>>4)+r][l&15]).  The result is a k-ordered
// fmaf chain starting from C (cdna_hip_programming.md §3), i.e. exactly the oracle's dot product.
//
// Row permutation: output row rho = 4q'+r' of tile t carries unit 16t + 4r' + q'.  Then lane group q,
// register r of tile t holds unit 4(4t+r) + q = "element q of k-step s = 4t+r" of the next layer.
// Per-lane operand tables (built in fgnn_weights_create), entry e at lane_tab[e*64 + lane]:
enum {
    T_W1 = 0,    // + side*3 + t          A operand of layer 1 (K=4: g, X, Y, Z)
    T_B1 = 6,    // + side*10 + s         bias of hidden unit 4s+q
    T_W2 = 26,   // + (side*2+u)*10 + s   A operand of layer 2, output tile u, k-step s
    T_B2 = 66,   // + side*5 + i          bias of message 4i+q (i<4) / 16+q (i=4)
    T_WE = 76,   // + t*11 + s            A operand of vn_embed_mlp, k-step s (K = 43 -> 44)
    T_BE = 109,  // + s
    T_WO = 119,  // + s                   A operand of _llr_inv_embed (3 -> 16 rows)
    T_BO = 129,  // + r                   bout[r] on lane group 0
    T_W10 = 132, // + side*10 + s         W1[0, 4s+q]: the check-feature row of layer 1 as a per-lane multiplier (factored order)
    T_COUNT = 152
};

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// CPB codewords per workgroup, four waves each: the 34 KB of operand tables are shared, so LDS no longer caps the CU at four
// workgroups of four waves — with CPB = 4 two 16-wave workgroups fill all 8 wave slots of every SIMD.
#ifndef FGNN_GNN_CPB
#define FGNN_GNN_CPB 4
#endif
#ifdef FGNN_GNN_WAVES
#define FGNN_GNN_OCC __attribute__((amdgpu_waves_per_eu(FGNN_GNN_WAVES, FGNN_GNN_WAVES)))
#else
#define FGNN_GNN_OCC
#endif
template <int DV, int CPB, bool FACT>
__global__ void __launch_bounds__(256 * CPB) FGNN_GNN_OCC gnn_mfma_kernel(GraphDev g, WeightsDev w, GnnArgs a)
{
    extern __shared__ float lds[];
    const int cwl = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int slot = blockIdx.x * CPB + cwl;  // (codeword, part): part p of nsplit takes tiles wave + 4p, wave + 4(p + nsplit), ...
    const int slot_b = slot / a.nsplit, part = slot - slot_b * a.nsplit;
    const bool active = slot_b < a.B;
    const int b = (active && a.index) ? a.index[slot_b] : slot_b;
    // LDS: the per-lane operand tables [T_COUNT][64] (shared by all waves; one conflict-free ds_read_b32 per
    // MFMA keeps ~130 registers free), then per codeword g_x | g_z  (:168-172)
    float* tabs = lds;
    float* gcn = lds + T_COUNT * 64 + cwl * a.lds_per_cw;
    const int n = g.n;
    for (int i = threadIdx.x; i < T_COUNT * 64; i += 256 * CPB) tabs[i] = w.lane_tab[i];
    if (active) {
        for (int c = tid; c < g.m_x; c += 256)
            gcn[c] = a.logit_hx[(size_t)b * g.m_x + c] * ((a.synd_x[(size_t)b * g.m_x + c] & 1) ? -1.0f : 1.0f);
        for (int c = tid; c < g.m_z; c += 256)
            gcn[g.m_x + c] = a.logit_hz[(size_t)b * g.m_z + c] * ((a.synd_z[(size_t)b * g.m_z + c] & 1) ? -1.0f : 1.0f);
    }
    __syncthreads();
    if (!active) return;

    const int l = threadIdx.x & 63, wave = tid >> 6;
    const int j = l & 15, q = l >> 4;
#define TAB(e) tabs[toff + (e) * 64]
    // the asm statements make the table OFFSET opaque per use site: hipcc would otherwise hoist all 132
    // loop-invariant LDS reads back into registers and spill.  (Laundering the pointer instead would drop its
    // LDS address space and turn every read into a flat_load.)
#define FRESH_TAB() int toff = l; asm volatile("" : "+v"(toff))
    const float* in = a.llr + (size_t)b * 3 * n;
    float* out = a.out + (size_t)b * 3 * n;
    const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const int ntiles = (n + 15) >> 4;
    for (int tile = wave + 4 * part; tile < ntiles; tile += 4 * a.nsplit) {
        const int vraw = tile * 16 + j;
        const bool valid = vraw < n;
        const int v = valid ? vraw : n - 1;
        // lane group q feeds feature q of layer 1 (g, X, Y, Z) and element q of the last embed k-step (X, Y, Z, 0)
        const float feat_own = (q == 0) ? 0.0f : in[(q - 1) * n + v];
        const float xyz0 = (q < 3) ? in[q * n + v] : 0.0f;
        float mean[2][5];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ebase = (s2 ? g.E_x : 0) + v * DV;
            const float* gside = gcn + (s2 ? g.m_x : 0);
            float esum[5];
            if constexpr (FACT) {
                // factored association (oracle: gnn_edge_side_factored).  [0, X, Y, Z] W1 through the same three MFMAs (lane group 0
                // feeds 0: the k = 0 step adds +0), + b1 once per side; per edge one fma with the check feature and a tanh; the
                // hidden activations are summed over the DV edges and the 40 -> 20 layer runs ONCE on the sum: 23 MFMAs per side, not 69.
                float Pb[10], Hs[10];
                {
                    FRESH_TAB();
                    f4 d[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) d[t] = mfma4(TAB(T_W1 + s2 * 3 + t), feat_own, zero);
#pragma unroll
                    for (int s = 0; s < 10; ++s) Pb[s] = d[s >> 2][s & 3] + TAB(T_B1 + s2 * 10 + s);
                }
#pragma unroll
                for (int k = 0; k < DV; ++k) {
                    FRESH_TAB();
                    const float gv = gside[g.vchk[ebase + k]];
#pragma unroll
                    for (int s = 0; s < 10; ++s) {
                        const float h = fg_tanh(FG_FMA(gv, TAB(T_W10 + s2 * 10 + s), Pb[s]));
                        Hs[s] = (k == 0) ? h : Hs[s] + h;
                    }
                }
                FRESH_TAB();
                f4 m0 = zero, m1 = zero;
#pragma unroll
                for (int s = 0; s < 10; ++s) {
                    m0 = mfma4(TAB(T_W2 + (s2 * 2 + 0) * 10 + s), Hs[s], m0);
                    m1 = mfma4(TAB(T_W2 + (s2 * 2 + 1) * 10 + s), Hs[s], m1);
                }
                const float ms[5] = {m0[0], m0[1], m0[2], m0[3], m1[0]};
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    mean[s2][i] = (DV == 3 ? fg_div3(ms[i]) : ms[i] / (float)DV) + TAB(T_B2 + s2 * 5 + i);
                continue;
            }
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                FRESH_TAB();
                const float gv = gside[g.vchk[ebase + k]];
                const float F = (q == 0) ? gv : feat_own;  // (:175-178)
                f4 d[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) d[t] = mfma4(TAB(T_W1 + s2 * 3 + t), F, zero);
                float H[10];
#pragma unroll
                for (int s = 0; s < 10; ++s) H[s] = fg_tanh(d[s >> 2][s & 3] + TAB(T_B1 + s2 * 10 + s));
                f4 m0 = zero, m1 = zero;
#pragma unroll
                for (int s = 0; s < 10; ++s) {
                    m0 = mfma4(TAB(T_W2 + (s2 * 2 + 0) * 10 + s), H[s], m0);
                    m1 = mfma4(TAB(T_W2 + (s2 * 2 + 1) * 10 + s), H[s], m1);
                }
                float msg[5] = {m0[0] + TAB(T_B2 + s2 * 5 + 0), m0[1] + TAB(T_B2 + s2 * 5 + 1), m0[2] + TAB(T_B2 + s2 * 5 + 2),
                                m0[3] + TAB(T_B2 + s2 * 5 + 3), m1[0] + TAB(T_B2 + s2 * 5 + 4)};
#pragma unroll
                for (int i = 0; i < 5; ++i) esum[i] = (k == 0) ? msg[i] : esum[i] + msg[i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) mean[s2][i] = DV == 3 ? fg_div3(esum[i]) : esum[i] / (float)DV;  // reduce_mean (:139-141)
        }
        // vn_embed_mlp on [m_x | m_z | X,Y,Z] then _llr_inv_embed  (:186)
        FRESH_TAB();
        f4 e[3] = {zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            const float Bs = (s < 5) ? mean[0][s] : (s < 10) ? mean[1][s - 5] : xyz0;
#pragma unroll
            for (int t = 0; t < 3; ++t) e[t] = mfma4(TAB(T_WE + t * 11 + s), Bs, e[t]);
        }
        f4 o = zero;
#pragma unroll
        for (int s = 0; s < 10; ++s) o = mfma4(TAB(T_WO + s), fg_tanh(e[s >> 2][s & 3] + TAB(T_BE + s)), o);
        if (valid && q == 0) {
            out[v] = o[0] + TAB(T_BO + 0);
            out[n + v] = o[1] + TAB(T_BO + 1);
            out[2 * n + v] = o[2] + TAB(T_BO + 2);
        }
    }
#undef TAB
#undef FRESH_TAB
}

template <bool FACT>
__global__ void __launch_bounds__(1024) gnn_kernel(GraphDev g, WeightsDev w, GnnArgs a)
{
    extern __shared__ float lds[];
    const int cwl = threadIdx.x / a.tpc;
    const int lane = threadIdx.x - cwl * a.tpc;
    const int slot_b = blockIdx.x * a.cpb + cwl;
    const bool active = slot_b < a.B;
    const int b = (active && a.index) ? a.index[slot_b] : slot_b;
    float* gcn = lds + (size_t)cwl * a.lds_per_cw;  // [m_x] g_x then [m_z] g_z  (:168-172)
    const int n = g.n;
    if (active) {
        for (int c = lane; c < g.m_x; c += a.tpc)
            gcn[c] = a.logit_hx[(size_t)b * g.m_x + c] * ((a.synd_x[(size_t)b * g.m_x + c] & 1) ? -1.0f : 1.0f);
        for (int c = lane; c < g.m_z; c += a.tpc)
            gcn[g.m_x + c] = a.logit_hz[(size_t)b * g.m_z + c] * ((a.synd_z[(size_t)b * g.m_z + c] & 1) ? -1.0f : 1.0f);
    }
    __syncthreads();
    if (!active) return;
    const float* in = a.llr + (size_t)b * 3 * n;
    float* out = a.out + (size_t)b * 3 * n;
    for (int v = lane; v < n; v += a.tpc) {
        const float X = in[v], Y = in[n + v], Z = in[2 * n + v];
        float feat[2 * MSG];
        {
            float mean[MSG];
            if constexpr (FACT) side_mean_factored(g, g.vptr_x, v, gcn, X, Y, Z, as_scalar(w.w1t[0]), as_scalar(w.b1[0]), as_scalar(w.w2[0]), as_scalar(w.b2[0]), mean);
            else side_mean(g, g.vptr_x, v, gcn, X, Y, Z, as_scalar(w.w1t[0]), as_scalar(w.b1[0]), as_scalar(w.w2[0]), as_scalar(w.b2[0]), mean);
#pragma unroll
            for (int i = 0; i < MSG; ++i) feat[i] = mean[i];
            // hz slots start at E_x in vchk, check ids are side-local: g_z lives at gcn + m_x
            if constexpr (FACT) side_mean_factored(g, g.vptr_z, v, gcn + g.m_x, X, Y, Z, as_scalar(w.w1t[1]), as_scalar(w.b1[1]), as_scalar(w.w2[1]), as_scalar(w.b2[1]), mean);
            else side_mean(g, g.vptr_z, v, gcn + g.m_x, X, Y, Z, as_scalar(w.w1t[1]), as_scalar(w.b1[1]), as_scalar(w.w2[1]), as_scalar(w.b2[1]), mean);
#pragma unroll
            for (int i = 0; i < MSG; ++i) feat[MSG + i] = mean[i];
        }
        // vn_embed_mlp Dense(40,tanh) on [m_x | m_z | X,Y,Z], then _llr_inv_embed Dense(3)  (:186)
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll 2
        for (int j = 0; j < HID; ++j) {
            scalar_fp r = as_scalar(w.wet) + j * 44;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 2 * MSG; ++k) acc = FG_FMA(feat[k], r[k], acc);
            acc = FG_FMA(X, r[40], acc);
            acc = FG_FMA(Y, r[41], acc);
            acc = FG_FMA(Z, r[42], acc);
            const float h = fg_tanh(acc + as_scalar(w.be)[j]);
            scalar_fp ro = as_scalar(w.wout) + j * 4;
            o0 = FG_FMA(h, ro[0], o0);
            o1 = FG_FMA(h, ro[1], o1);
            o2 = FG_FMA(h, ro[2], o2);
        }
        out[v] = o0 + w.bout[0];
        out[n + v] = o1 + w.bout[1];
        out[2 * n + v] = o2 + w.bout[2];
    }
}


// ---------------------------------------------------------------------------------------------
// Streaming VALU kernel for degree-regular graphs in the FACTORED association (round 3).  On gfx950 an f32 MFMA delivers the same 64
// FLOP/clk/SIMD as v_fma_f32 and runs on the same lanes, so its only effect on this layer stack is the padding of the
// 40-, 20- and 3-row outputs to 16-row tiles (66 % useful).  The factored order needs no hidden VECTOR: a lane (= one qubit) walks the
// 40 hidden units, forms pb = [X,Y,Z] W1[1:4,j] + b1[j], the DV tanh(g_e W1[0,j] + pb), their sum, and streams it into its 20
// message accumulators; the embed MLP likewise streams its 40 units into the 3 outputs.  Every weight is a uniform scalar (s_load ->
// SGPR operand of v_fma), no LDS tables, ~70 VGPRs.  Same operations in the same order as side_mean_factored / the oracle.
// Per-unit weight rows are packed at upload: msg_rows[side][j][32] = W1[0..3][j], b1[j], 0,0,0, W2[j][0..19], 0,0,0,0;
// emb_rows[j][48] = We[0..42][j], be[j], Wout[j][0..2], 0.
// ---------------------------------------------------------------------------------------------
#ifndef FGNN_GNNS_EMB_U
#define FGNN_GNNS_EMB_U 4  // hidden units of the embed MLP per block of the packed form (EMB_U / 2 independent packed chains)
#endif
constexpr int EMB_U = FGNN_GNNS_EMB_U;
constexpr int SROW = 32, EROW = 48, EQUAD = ((43 + 1 + 3) * EMB_U + 31) / 32 * 32;  // 192 floats for blocks of four
static_assert(HID % EMB_U == 0 && EMB_U % 2 == 0, "the embed MLP walks its hidden units in blocks");
// measured (profiles/r3_gnn_stream_ab.txt): a register budget of 6 or 7 waves per SIMD, one hidden unit per loop trip (its DV tanh chains
// interleave) and the LDS reads of the check features waited for BEFORE the unit loop are the best of waves 6 / 7 / 8 x unroll 1 / 2
// x software-pipelined or not; requesting the next unit's row ahead of time (hand-placed s_load / s_waitcnt) was slower, because the
// compiler then serialises the tanh chains; with no weight loads at all (timing probe) the kernel is no faster: it is bound by VALU issue
#ifndef FGNN_GNNS_UNROLL
#define FGNN_GNNS_UNROLL 1
#endif
#ifndef FGNN_GNNS_WAVES
#define FGNN_GNNS_WAVES 7
#endif
#ifndef FGNN_GNNS_EMBPK_FACT
#define FGNN_GNNS_EMBPK_FACT 1
#endif
#ifndef FGNN_GNNS_EMBPK_LIT
#define FGNN_GNNS_EMBPK_LIT 1
#endif
#ifndef FGNN_GNNS_EMB_KG
#define FGNN_GNNS_EMB_KG 4  // embed-MLP weight rows (of four floats) per scalar-load group
#endif
#ifndef FGNN_GNNS_LIT_UNROLL
#define FGNN_GNNS_LIT_UNROLL 1  // measured: 13.4 ms at 1, 13.8 at 2, 13.7 at 4 ([[882,24]] x 65 536)
#endif
#ifndef FGNN_GNNS_LIT_WAVES
#define FGNN_GNNS_LIT_WAVES 5  // the literal association keeps one more 20-wide accumulator alive per lane
#endif

#define FGNN_GNNS_OCC __attribute__((amdgpu_waves_per_eu(FGNN_GNNS_WAVES, FGNN_GNNS_WAVES)))

struct MsgRow {
    float w1[4], b1, w2[MSG];
};
__device__ __forceinline__ MsgRow load_msg_row(scalar_fp r)
{
    MsgRow m;
#pragma unroll
    for (int k = 0; k < 4; ++k) m.w1[k] = r[k];
    m.b1 = r[4];
#pragma unroll
    for (int i = 0; i < MSG; ++i) m.w2[i] = r[8 + i];
    return m;
}

template <int DV>
__device__ __forceinline__ float msg_hidden(const MsgRow& r, const float (&gv)[DV], float X, float Y, float Z)
{
    float p = 0.0f;
    p = FG_FMA(X, r.w1[1], p);
    p = FG_FMA(Y, r.w1[2], p);
    p = FG_FMA(Z, r.w1[3], p);
    const float pb = p + r.b1;
    float hs = fg_tanh(FG_FMA(gv[0], r.w1[0], pb));
#pragma unroll
    for (int k = 1; k < DV; ++k) hs = hs + fg_tanh(FG_FMA(gv[k], r.w1[0], pb));
    return hs;
}
template <int DV>
__device__ __forceinline__ void msg_unit(const MsgRow& r, const float (&gv)[DV], float X, float Y, float Z, float (&acc)[MSG])
{
    const float hs = msg_hidden<DV>(r, gv, X, Y, Z);
#pragma unroll
    for (int i = 0; i < MSG; ++i) acc[i] = FG_FMA(hs, r.w2[i], acc[i]);
}

template <int DV>
__device__ __forceinline__ void side_stream(scalar_fp rows, scalar_fp b2, const float (&gv)[DV],
                                            float X, float Y, float Z, float* __restrict__ feat)
{
    float acc[MSG];
#pragma unroll
    for (int i = 0; i < MSG; ++i) acc[i] = 0.0f;
#pragma unroll FGNN_GNNS_UNROLL
    for (int j = 0; j < HID; ++j) msg_unit<DV>(load_msg_row(rows + j * SROW), gv, X, Y, Z, acc);
#pragma unroll
    for (int i = 0; i < MSG; ++i) feat[i] = (DV == 3 ? fg_div3(acc[i]) : acc[i] / (float)DV) + b2[i];
}

// The same side in the LITERAL association (feedback_gnn.py:175-184 term by term; oracle: gnn_edge_side): one whole message MLP per
// edge — four-term first Dense from 0, + b1, tanh, 40 -> 20 Dense over ascending j from 0, + b2 — then the edges' messages summed in
// ascending check order and divided by their number.  One lane per qubit, weights as scalar operands like side_stream; the edge loop
// is a real loop (unrolled, the compiler interleaves the edges and spills a hundred registers).  Round 4: 12.9 ms against the MFMA
// tiles' 14.7 on [[882,24]] x 65 536 (profiles/r4_gnn_literal_stream_ab.txt); with plain v_fmac for the second Dense it was 15.2.
// Round 6, measured and NOT merged (profiles/r6_literal_kernel_attempts.txt): two hidden units per trip with the four-term first Dense
// and the bias as v_pk_fma_f32 / v_pk_add_f32 on SGPR pairs (57 instead of 62 VALU instructions per two units and edge, two
// independent tanh chains per trip, bit-equal): 13.18 ms against 12.97 ([[1270,28]] x 32 768: 9.42 against 9.25) — the loop waits on
// its scalar weight loads either way and the five half-rate v_fmac it removes were not the bound.
template <int DV>
__device__ __forceinline__ void side_stream_literal(scalar_fp rows, scalar_fp b2, const float (&gv)[DV], float X, float Y, float Z,
                                                    float* __restrict__ feat)
{
    // the 40 -> 20 Dense of an edge as ten v_pk_fma_f32 per hidden unit: two consecutive message elements per instruction, their
    // weights W2[j][2i], W2[j][2i + 1] one SGPR pair (rows are 32 floats, W2 at float 8: 8-byte aligned), the per-lane tanh value
    // broadcast to both halves — each half an IEEE fma in the oracle's order (ascending j from 0)
#pragma unroll 1
    for (int e = 0; e < DV; ++e) {
        float ge = gv[0];
#pragma unroll
        for (int k = 1; k < DV; ++k) ge = e == k ? gv[k] : ge;
        f2 m[MSG / 2];
#pragma unroll
        for (int i = 0; i < MSG / 2; ++i) m[i] = bc2(0.0f);
#pragma unroll FGNN_GNNS_LIT_UNROLL
        for (int j = 0; j < HID; ++j) {
            scalar_fp r = rows + j * SROW;
            scalar_f2p w2 = (scalar_f2p)(r + 8);
            float a = 0.0f;
            a = FG_FMA(ge, r[0], a);
            a = FG_FMA(X, r[1], a);
            a = FG_FMA(Y, r[2], a);
            a = FG_FMA(Z, r[3], a);
            const f2 h = bc_lo(fg_tanh(a + r[4]));
#pragma unroll
            for (int i = 0; i < MSG / 2; ++i) m[i] = pk_fma(h, w2[i], m[i]);
        }
#pragma unroll
        for (int i = 0; i < MSG / 2; ++i) {
            const float m0 = m[i].x + b2[2 * i], m1 = m[i].y + b2[2 * i + 1];
            feat[2 * i] = (e == 0) ? m0 : feat[2 * i] + m0;
            feat[2 * i + 1] = (e == 0) ? m1 : feat[2 * i + 1] + m1;
        }
    }
#pragma unroll
    for (int i = 0; i < MSG; ++i) feat[i] = DV == 3 ? fg_div3(feat[i]) : feat[i] / (float)DV;
}

struct EmbRow {
    float we[2 * MSG + 3], be, wo[3];
};
__device__ __forceinline__ EmbRow load_emb_row(scalar_fp r)
{
    EmbRow m;
#pragma unroll
    for (int k = 0; k < 2 * MSG + 3; ++k) m.we[k] = r[k];
    m.be = r[43];
#pragma unroll
    for (int i = 0; i < 3; ++i) m.wo[i] = r[44 + i];
    return m;
}
__device__ __forceinline__ void emb_unit(const EmbRow& r, const float* __restrict__ feat, float X, float Y, float Z, float (&o)[3])
{
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 2 * MSG; ++k) acc = FG_FMA(feat[k], r.we[k], acc);
    acc = FG_FMA(X, r.we[40], acc);
    acc = FG_FMA(Y, r.we[41], acc);
    acc = FG_FMA(Z, r.we[42], acc);
    const float h = fg_tanh(acc + r.be);
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = FG_FMA(h, r.wo[i], o[i]);
}

// Four hidden units of the embed MLP at once (emb_quads): the 43-term first Dense as v_pk_fma_f32 — two units per instruction, their
// weights We[k][j], We[k][j + 1] one SGPR pair, the per-lane input broadcast to both halves by op_sel; two independent chains.  Each half
// is the IEEE fma of emb_unit in the same order (ascending k from 0), so the bits are those of the scalar-operand form.  Measured
// (round 4, [[882,24]] x 65 536): factored kernel 9.25 -> 9.08 ms, literal 13.48 -> 12.94; blocks of 2, 4 or 8 units and 2 .. 8 weight rows
// per scalar-load group all within 1 % of each other; the SAME packing of the factored association's 40 -> 20 Dense gains nothing
// (9.06 / 9.08 ms): that kernel is bound by the tanh chains' issue slots, and a packed fma costs two.  The inputs are passed through
// an empty asm IN PLACE first: loop-invariant pairs would otherwise have their {x, x} broadcasts hoisted out of the caller's loop as 43
// more live register pairs.
__device__ __forceinline__ void emb_quad(const float* quad, f2 (&fx)[2 * MSG / 2 + 2], float (&o)[3])
{
#pragma unroll
    for (int q = 0; q < 2 * MSG / 2 + 2; ++q) asm volatile("" : "+v"(fx[q]));
    f2 acc[EMB_U / 2];
#pragma unroll
    for (int jp = 0; jp < EMB_U / 2; ++jp) acc[jp] = bc2(0.0f);
    dense_pk<2 * MSG + 3, EMB_U / 2, FGNN_GNNS_EMB_KG, false>(fx, quad, EMB_U, acc);
    scalar_fp r = as_scalar(quad);
    float h[EMB_U];
#pragma unroll
    for (int jp = 0; jp < EMB_U / 2; ++jp) {
        h[2 * jp] = fg_tanh(acc[jp].x + r[43 * EMB_U + 2 * jp]);
        h[2 * jp + 1] = fg_tanh(acc[jp].y + r[43 * EMB_U + 2 * jp + 1]);
    }
#pragma unroll
    for (int u = 0; u < EMB_U; ++u)
#pragma unroll
        for (int i = 0; i < 3; ++i) o[i] = FG_FMA(h[u], r[44 * EMB_U + 3 * u + i], o[i]);
}

template <int DV, bool LITERAL = false, bool EMBPK = false>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(LITERAL ? FGNN_GNNS_LIT_WAVES : FGNN_GNNS_WAVES, LITERAL ? FGNN_GNNS_LIT_WAVES : FGNN_GNNS_WAVES)))
gnn_stream_kernel(GraphDev g, WeightsDev w, GnnArgs a)
{
    extern __shared__ float lds[];
    const int slot_b = blockIdx.x;
    const int b = a.index ? a.index[slot_b] : slot_b;
    float* gcn = lds;  // [m_x] g_x then [m_z] g_z  (:168-172)
    const int n = g.n;
    for (int c = threadIdx.x; c < g.m_x; c += blockDim.x)
        gcn[c] = a.logit_hx[(size_t)b * g.m_x + c] * ((a.synd_x[(size_t)b * g.m_x + c] & 1) ? -1.0f : 1.0f);
    for (int c = threadIdx.x; c < g.m_z; c += blockDim.x)
        gcn[g.m_x + c] = a.logit_hz[(size_t)b * g.m_z + c] * ((a.synd_z[(size_t)b * g.m_z + c] & 1) ? -1.0f : 1.0f);
    __syncthreads();
    const float* in = a.llr + (size_t)b * 3 * n;
    float* out = a.out + (size_t)b * 3 * n;
    scalar_fp mrx = as_scalar((const float*)__builtin_assume_aligned(w.msg_rows[0], 128));
    scalar_fp mrz = as_scalar((const float*)__builtin_assume_aligned(w.msg_rows[1], 128));
    scalar_fp er = as_scalar((const float*)__builtin_assume_aligned(w.emb_rows, 64));
    scalar_fp b2x = as_scalar(w.b2[0]), b2z = as_scalar(w.b2[1]), bo = as_scalar(w.bout);
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
        const float X = in[v], Y = in[n + v], Z = in[2 * n + v];
        f2 fx[MSG + 2];  // [m_x | m_z | X, Y | Z, -] as pairs of consecutive elements (the packed embed MLP reads them as such)
        float* feat = reinterpret_cast<float*>(fx);
        float gv[DV];
#pragma unroll
        for (int k = 0; k < DV; ++k) gv[k] = gcn[g.vchk[v * DV + k]];
        // LDS reads and scalar loads share one counter (lgkmcnt): a use of gv here has the compiler wait for the LDS reads before the
        // unit loop instead of inside it, where the wait would also cover the loop's weight loads
#pragma unroll
        for (int k = 0; k < DV; ++k) asm volatile("" : "+v"(gv[k]));
        if constexpr (LITERAL) side_stream_literal<DV>(mrx, b2x, gv, X, Y, Z, feat);
        else side_stream<DV>(mrx, b2x, gv, X, Y, Z, feat);
#pragma unroll
        for (int k = 0; k < DV; ++k) gv[k] = gcn[g.m_x + g.vchk[g.E_x + v * DV + k]];
#pragma unroll
        for (int k = 0; k < DV; ++k) asm volatile("" : "+v"(gv[k]));
        if constexpr (LITERAL) side_stream_literal<DV>(mrz, b2z, gv, X, Y, Z, feat + MSG);
        else side_stream<DV>(mrz, b2z, gv, X, Y, Z, feat + MSG);
        // vn_embed_mlp Dense(40,tanh) on [m_x | m_z | X,Y,Z], then _llr_inv_embed Dense(3)  (:186)
        float o[3] = {0.0f, 0.0f, 0.0f};
        if constexpr (EMBPK) {
            fx[MSG] = f2{X, Y};
            fx[MSG + 1] = f2{Z, 0.0f};
#pragma unroll 1
            for (int q = 0; q < HID / EMB_U; ++q) emb_quad(w.emb_quads + q * EQUAD, fx, o);
        } else {
#pragma unroll FGNN_GNNS_UNROLL
            for (int j = 0; j < HID; ++j) emb_unit(load_emb_row(er + j * EROW), feat, X, Y, Z, o);
        }
        out[v] = o[0] + bo[0];
        out[n + v] = o[1] + bo[1];
        out[2 * n + v] = o[2] + bo[2];
    }
}

// ---------------------------------------------------------------------------------------------
// Runtime-shaped Feedback_GNN (any num_msg_dims / num_hidden_units / num_mlp_layers / reduce_op / activation / use_bias the
// reference's constructor accepts within the limits of fgnn.h): one thread per qubit, Dense = fmaf chain in ascending k from 0
// then (+ bias) then activation, exactly as the oracle (og_feedback_gnn_general).  Activations of one layer ping-pong between
// two per-thread buffers (scratch memory: this is the compatibility path, not the benchmark path).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gen_act(float a, int act)
{
    switch (act) {
    case FGNN_ACT_TANH: return fg_tanh(a);
    case FGNN_ACT_RELU: return FG_MAX(a, 0.0f);
    case FGNN_ACT_SIGMOID: return fg_sigmoid(a);
    default: return a;
    }
}

__device__ __forceinline__ void gen_dense(const GnnGeneralDev& w, int li, const float* in, float* out)
{
    const int K = w.K[li], J = w.J[li], act = w.act_l[li];
    const float* W = w.W[li];
    const float* b = w.b[li];
    for (int j = 0; j < J; ++j) {
        float a = 0.0f;
        for (int k = 0; k < K; ++k) a = FG_FMA(in[k], W[k * J + j], a);
        if (b) a = a + b[j];
        out[j] = gen_act(a, act);
    }
}

__global__ void __launch_bounds__(1024) gnn_general_kernel(GraphDev g, GnnGeneralDev w, GnnArgs a)
{
    extern __shared__ float lds[];
    const int cwl = threadIdx.x / a.tpc;
    const int lane = threadIdx.x - cwl * a.tpc;
    const int slot_b = blockIdx.x * a.cpb + cwl;
    const bool active = slot_b < a.B;
    const int b = (active && a.index) ? a.index[slot_b] : slot_b;
    float* gcn = lds + (size_t)cwl * a.lds_per_cw;
    const int n = g.n;
    if (active) {
        for (int c = lane; c < g.m_x; c += a.tpc)
            gcn[c] = a.logit_hx[(size_t)b * g.m_x + c] * ((a.synd_x[(size_t)b * g.m_x + c] & 1) ? -1.0f : 1.0f);
        for (int c = lane; c < g.m_z; c += a.tpc)
            gcn[g.m_x + c] = a.logit_hz[(size_t)b * g.m_z + c] * ((a.synd_z[(size_t)b * g.m_z + c] & 1) ? -1.0f : 1.0f);
    }
    __syncthreads();
    if (!active) return;
    const float* in = a.llr + (size_t)b * 3 * n;
    float* out = a.out + (size_t)b * 3 * n;
    const int D = w.D, L = w.L;
    for (int v = lane; v < n; v += a.tpc) {
        const float X = in[v], Y = in[n + v], Z = in[2 * n + v];
        float bufA[FGNN_GEN_MAX_W], bufB[FGNN_GEN_MAX_W], z[2 * FGNN_GEN_MAX_D + 3];
        for (int s = 0; s < 2; ++s) {
            const int* vptr = s == 0 ? g.vptr_x : g.vptr_z;
            const float* gc = s == 0 ? gcn : gcn + g.m_x;
            const int e0 = vptr[v], e1 = vptr[v + 1];
            float* acc = z + s * D;
            for (int i = 0; i < D; ++i) acc[i] = 0.0f;
            for (int e = e0; e < e1; ++e) {
                float* cur = bufA;
                float* nxt = bufB;
                cur[0] = gc[g.vchk[e]];  // (:175-178)
                cur[1] = X;
                cur[2] = Y;
                cur[3] = Z;
                for (int l = 0; l < L; ++l) {
                    gen_dense(w, s * L + l, cur, nxt);
                    float* t = cur;
                    cur = nxt;
                    nxt = t;
                }
                for (int i = 0; i < D; ++i) {  // reduce_msg (:130-150), edges in ascending check order
                    const float m = cur[i];
                    float r;
                    if (e == e0) r = m;
                    else if (w.reduce_op == FGNN_REDUCE_MAX) r = FG_MAX(acc[i], m);
                    else if (w.reduce_op == FGNN_REDUCE_MIN) r = FG_MIN(acc[i], m);
                    else r = acc[i] + m;
                    acc[i] = r;
                }
            }
            if (w.reduce_op == FGNN_REDUCE_MEAN && e1 > e0) {
                const float fd = (float)(e1 - e0);
                for (int i = 0; i < D; ++i) acc[i] = acc[i] / fd;
            }
        }
        z[2 * D] = X;
        z[2 * D + 1] = Y;
        z[2 * D + 2] = Z;
        const float* cur = z;
        float* nxt = bufA;
        for (int l = 0; l < L - 1; ++l) {  // vn_embed_mlp (:186)
            gen_dense(w, 2 * L + l, cur, nxt);
            cur = nxt;
            nxt = (nxt == bufA) ? bufB : bufA;
        }
        float o[3];
        gen_dense(w, 3 * L - 1, cur, o);  // _llr_inv_embed
        out[v] = o[0];
        out[n + v] = o[1];
        out[2 * n + v] = o[2];
    }
}

}  // namespace

extern "C" int fgnn_weights_create(const float* const host_arrays[12], int device, fgnn_weights** out)
{
    if (!host_arrays || !out) return fgnn_fail(FGNN_ERR_ARG, "NULL argument");
    for (int i = 0; i < 12; ++i)
        if (!host_arrays[i]) return fgnn_fail(FGNN_ERR_ARG, "weight array is NULL");
    FGNN_DEVICE_GUARD(device);
    // blob layout (floats): w1t_x 160 | b1_x 40 | w2_x 800 | b2_x 20 | same for z | wet 1760 | be 40 | wout 160 | bout 4
    std::vector<float> h;
    size_t off[12];
    auto push = [&](size_t count) {
        size_t o = h.size();
        h.resize(o + ((count + 3) & ~size_t(3)), 0.0f);
        return o;
    };
    for (int s = 0; s < 2; ++s) {
        const float* W1 = host_arrays[2 + 4 * s];  // [4,40]
        const float* B1 = host_arrays[3 + 4 * s];
        const float* W2 = host_arrays[4 + 4 * s];  // [40,20]
        const float* B2 = host_arrays[5 + 4 * s];
        off[4 * s + 0] = push(HID * 4);
        for (int j = 0; j < HID; ++j)
            for (int k = 0; k < 4; ++k) h[off[4 * s + 0] + j * 4 + k] = W1[k * HID + j];
        off[4 * s + 1] = push(HID);
        std::memcpy(&h[off[4 * s + 1]], B1, HID * sizeof(float));
        off[4 * s + 2] = push(HID * MSG);
        std::memcpy(&h[off[4 * s + 2]], W2, HID * MSG * sizeof(float));
        off[4 * s + 3] = push(MSG);
        std::memcpy(&h[off[4 * s + 3]], B2, MSG * sizeof(float));
    }
    off[8] = push(HID * 44);
    for (int j = 0; j < HID; ++j)
        for (int k = 0; k < 43; ++k) h[off[8] + j * 44 + k] = host_arrays[10][k * HID + j];
    off[9] = push(HID);
    std::memcpy(&h[off[9]], host_arrays[11], HID * sizeof(float));
    off[10] = push(HID * 4);
    for (int j = 0; j < HID; ++j)
        for (int i = 0; i < 3; ++i) h[off[10] + j * 4 + i] = host_arrays[0][j * 3 + i];
    off[11] = push(4);
    std::memcpy(&h[off[11]], host_arrays[1], 3 * sizeof(float));

    // per-unit rows of the streaming kernel (gnn_stream_kernel)
    size_t off_mr[2];
    for (int s = 0; s < 2; ++s) {
        while (h.size() % 32) h.push_back(0.0f);  // 128-byte aligned rows (hipMalloc aligns the blob to 256)
        off_mr[s] = push((size_t)HID * SROW);
        const float* W1 = host_arrays[2 + 4 * s];
        const float* B1 = host_arrays[3 + 4 * s];
        const float* W2 = host_arrays[4 + 4 * s];
        for (int j = 0; j < HID; ++j) {
            float* r = &h[off_mr[s] + (size_t)j * SROW];
            for (int k = 0; k < 4; ++k) r[k] = W1[k * HID + j];
            r[4] = B1[j];
            for (int i = 0; i < MSG; ++i) r[8 + i] = W2[j * MSG + i];
        }
    }
    while (h.size() % 32) h.push_back(0.0f);
    const size_t off_er = push((size_t)HID * EROW);
    for (int j = 0; j < HID; ++j) {
        float* r = &h[off_er + (size_t)j * EROW];
        for (int k = 0; k < 43; ++k) r[k] = host_arrays[10][k * HID + j];
        r[43] = host_arrays[11][j];
        for (int i = 0; i < 3; ++i) r[44 + i] = host_arrays[0][j * 3 + i];
    }
    while (h.size() % 32) h.push_back(0.0f);
    const size_t off_eq = push((size_t)(HID / EMB_U) * EQUAD);
    for (int q = 0; q < HID / EMB_U; ++q) {
        float* r = &h[off_eq + (size_t)q * EQUAD];
        for (int k = 0; k < 43; ++k)
            for (int u = 0; u < EMB_U; ++u) r[EMB_U * k + u] = host_arrays[10][k * HID + EMB_U * q + u];
        for (int u = 0; u < EMB_U; ++u) {
            r[43 * EMB_U + u] = host_arrays[11][EMB_U * q + u];
            for (int i = 0; i < 3; ++i) r[44 * EMB_U + 3 * u + i] = host_arrays[0][(EMB_U * q + u) * 3 + i];
        }
    }
    // per-lane MFMA operand tables (see the T_* enum above)
    const size_t off_tab = push((size_t)T_COUNT * 64);
    {
        float* T = &h[off_tab];
        auto put = [&](int entry, int lane, float val) { T[(size_t)entry * 64 + lane] = val; };
        for (int lane = 0; lane < 64; ++lane) {
            const int rho = lane & 15, kk = lane >> 4, qp = rho >> 2, rp = rho & 3, qq = lane >> 4;
            for (int s2 = 0; s2 < 2; ++s2) {
                const float* W1 = host_arrays[2 + 4 * s2];
                const float* B1 = host_arrays[3 + 4 * s2];
                const float* W2 = host_arrays[4 + 4 * s2];
                const float* B2 = host_arrays[5 + 4 * s2];
                for (int t = 0; t < 3; ++t) {
                    const int unit = 16 * t + 4 * rp + qp;
                    put(T_W1 + s2 * 3 + t, lane, unit < HID ? W1[kk * HID + unit] : 0.0f);
                }
                for (int s = 0; s < 10; ++s) put(T_B1 + s2 * 10 + s, lane, B1[4 * s + qq]);
                for (int u = 0; u < 2; ++u)
                    for (int s = 0; s < 10; ++s) {
                        const int mu = (u == 0) ? 4 * rp + qp : (rp == 0 ? 16 + qp : -1);
                        put(T_W2 + (s2 * 2 + u) * 10 + s, lane, mu >= 0 ? W2[(4 * s + kk) * MSG + mu] : 0.0f);
                    }
                for (int i = 0; i < 5; ++i) put(T_B2 + s2 * 5 + i, lane, B2[i < 4 ? 4 * i + qq : 16 + qq]);
                for (int s = 0; s < 10; ++s) put(T_W10 + s2 * 10 + s, lane, W1[0 * HID + 4 * s + qq]);
            }
            for (int t = 0; t < 3; ++t)
                for (int s = 0; s < 11; ++s) {
                    const int unit = 16 * t + 4 * rp + qp, k = 4 * s + kk;
                    put(T_WE + t * 11 + s, lane, (unit < HID && k < 43) ? host_arrays[10][k * HID + unit] : 0.0f);
                }
            for (int s = 0; s < 10; ++s) {
                put(T_BE + s, lane, host_arrays[11][4 * s + qq]);
                put(T_WO + s, lane, rho < 3 ? host_arrays[0][(4 * s + kk) * 3 + rho] : 0.0f);
            }
            for (int r = 0; r < 3; ++r) put(T_BO + r, lane, host_arrays[1][r]);
        }
    }

    fgnn_weights* w = new fgnn_weights();
    w->device = device;
    w->blob = nullptr;
    hipError_t e = hipMalloc(&w->blob, h.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(w->blob, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (w->blob) (void)hipFree(w->blob);
        delete w;
        return fgnn_fail(FGNN_ERR_HIP, std::string("weights upload: ") + hipGetErrorString(e));
    }
    const float* base = static_cast<const float*>(w->blob);
    for (int s = 0; s < 2; ++s) {
        w->d.w1t[s] = base + off[4 * s + 0];
        w->d.b1[s] = base + off[4 * s + 1];
        w->d.w2[s] = base + off[4 * s + 2];
        w->d.b2[s] = base + off[4 * s + 3];
    }
    w->d.wet = base + off[8];
    w->d.be = base + off[9];
    w->d.wout = base + off[10];
    w->d.bout = base + off[11];
    w->d.lane_tab = base + off_tab;
    w->d.msg_rows[0] = base + off_mr[0];
    w->d.msg_rows[1] = base + off_mr[1];
    w->d.emb_rows = base + off_er;
    w->d.emb_quads = base + off_eq;
    *out = w;
    return FGNN_OK;
}

extern "C" int fgnn_weights_create_general(const fgnn_gnn_config* cfg, const float* const* host_arrays, int num_arrays,
                                           int device, fgnn_weights** out)
{
    if (!cfg || !host_arrays || !out) return fgnn_fail(FGNN_ERR_ARG, "NULL argument");
    const int D = cfg->num_msg_dims, H = cfg->num_hidden_units, L = cfg->num_mlp_layers, bias = cfg->use_bias ? 1 : 0;
    if (D < 1 || D > FGNN_GEN_MAX_D || L < 1 || L > 4 || (L > 1 && (H < 1 || H > FGNN_GEN_MAX_W)))
        return fgnn_fail(FGNN_ERR_ARG, "general feedback GNN: need 1 <= num_msg_dims <= 32, 1 <= num_hidden_units <= 96, "
                                       "1 <= num_mlp_layers <= 4");
    if (cfg->reduce_op < 0 || cfg->reduce_op > 3) return fgnn_fail(FGNN_ERR_ARG, "unknown reduce operation");  // :148
    if (cfg->activation < 0 || cfg->activation > 3) return fgnn_fail(FGNN_ERR_ARG, "unsupported activation");
    if (num_arrays != 3 * L * (1 + bias)) return fgnn_fail(FGNN_ERR_ARG, "wrong number of weight arrays for this configuration");
    for (int i = 0; i < num_arrays; ++i)
        if (!host_arrays[i]) return fgnn_fail(FGNN_ERR_ARG, "weight array is NULL");
    FGNN_DEVICE_GUARD(device);
    fgnn_weights* w = new fgnn_weights();
    w->device = device;
    w->blob = nullptr;
    w->general = true;
    std::memset(&w->d, 0, sizeof(w->d));
    GnnGeneralDev& G = w->gen;
    std::memset(&G, 0, sizeof(G));
    G.D = D; G.H = H; G.L = L; G.reduce_op = cfg->reduce_op; G.act = cfg->activation; G.bias = bias; G.nl = 3 * L;
    // execution order -> (K, J, activation); file order: _llr_inv_embed first, then msg_x, msg_z, embed
    const int emb_out = L > 1 ? H : 2 * D + 3;
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < L; ++l) {
            const int li = s * L + l;
            G.K[li] = l == 0 ? 4 : H;
            G.J[li] = l == L - 1 ? D : H;
            G.act_l[li] = l == L - 1 ? FGNN_ACT_LINEAR : cfg->activation;
        }
    for (int l = 0; l < L - 1; ++l) {
        const int li = 2 * L + l;
        G.K[li] = l == 0 ? 2 * D + 3 : H;
        G.J[li] = H;
        G.act_l[li] = cfg->activation;
    }
    G.K[3 * L - 1] = emb_out;
    G.J[3 * L - 1] = 3;
    G.act_l[3 * L - 1] = FGNN_ACT_LINEAR;
    std::vector<float> h;
    std::vector<size_t> offW(G.nl), offB(G.nl);
    auto file_index = [&](int li) { return li == 3 * L - 1 ? 0 : li + 1; };  // position among the Dense layers of the file
    for (int li = 0; li < G.nl; ++li) {
        const int f = file_index(li) * (1 + bias);
        const size_t cnt = (size_t)G.K[li] * G.J[li];
        offW[li] = h.size();
        h.insert(h.end(), host_arrays[f], host_arrays[f] + cnt);
        h.resize((h.size() + 3) & ~size_t(3), 0.0f);
        offB[li] = h.size();
        if (bias) {
            h.insert(h.end(), host_arrays[f + 1], host_arrays[f + 1] + G.J[li]);
            h.resize((h.size() + 3) & ~size_t(3), 0.0f);
        }
    }
    hipError_t e = hipMalloc(&w->blob, h.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(w->blob, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (w->blob) (void)hipFree(w->blob);
        delete w;
        return fgnn_fail(FGNN_ERR_HIP, std::string("weights upload: ") + hipGetErrorString(e));
    }
    const float* base = static_cast<const float*>(w->blob);
    for (int li = 0; li < G.nl; ++li) {
        G.W[li] = base + offW[li];
        G.b[li] = bias ? base + offB[li] : nullptr;
    }
    *out = w;
    return FGNN_OK;
}

extern "C" void fgnn_weights_destroy(fgnn_weights* w)
{
    if (!w) return;
    fgnn_device_guard _dg(w->device);
    if (w->blob) (void)hipFree(w->blob);
    delete w;
}

int fgnn_feedback_gnn_impl(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                           const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B, float* out,
                           const int* index, void* stream)
{
    if (!g || !w) return fgnn_fail(FGNN_ERR_ARG, "graph or weights is NULL");
    if (B < 0) return fgnn_fail(FGNN_ERR_ARG, "B must be >= 0");
    if (w->device != g->device) return fgnn_fail(FGNN_ERR_ARG, "weights and graph live on different devices");
    if (B == 0) return FGNN_OK;  // an empty batch needs no buffers
    if (!llr || !logit_hx || !logit_hz || !synd_x || !synd_z || !out) return fgnn_fail(FGNN_ERR_ARG, "buffer is NULL");
    FGNN_DEVICE_GUARD(g->device);
    LaunchGeom L = fgnn_geom(g, B);
    GnnArgs a;
    a.B = B;
    a.tpc = L.tpc;
    a.cpb = L.cpb;
    a.lds_per_cw = (g->d.m + 3) & ~3;
    a.llr = llr;
    a.logit_hx = logit_hx;
    a.logit_hz = logit_hz;
    a.synd_x = synd_x;
    a.synd_z = synd_z;
    a.out = out;
    a.index = index;
    a.nsplit = 1;
    fgnn_prof_scope prof(g, static_cast<hipStream_t>(stream));
    if (w->general) {
        size_t lds_gen = (size_t)a.lds_per_cw * sizeof(float) * (size_t)L.cpb;
        hipLaunchKernelGGL(gnn_general_kernel, dim3(L.blocks), dim3(L.threads), lds_gen, static_cast<hipStream_t>(stream), g->d,
                           w->gen, a);
        FGNN_HIP_CHECK(hipGetLastError());
        prof.done(FGNN_PROF_TAG_GNN, B);
        return FGNN_OK;
    }
    // Below ~4 000 codewords the launch is latency-bound, and there the MFMA-tile kernel, which deals one codeword's tiles to many waves,
    // is up to 3x quicker (18 vs 52 us for <= 64 codewords of [[882,24]]; equal from 256 to 2 048; the streaming kernel wins from 4 096 on:
    // profiles/r3_gnn_stream_ab.txt; the literal association's streaming kernel likewise: profiles/r4_gnn_literal_stream_ab.txt) -
    // same bits either way.  Degrees 4 and 5 have no MFMA-tile kernel and always stream.
    const bool stream_pays = g->gnn_stream == 2 || g->d.dvx != 3 || B >= 4096;
    if (g->d.dvx == g->d.dvz && g->d.dvx >= 3 && g->d.dvx <= 5 && !g->force_generic && g->gnn_stream && stream_pays) {
        // degree-regular graph (3, 4 or 5 checks per qubit and side: the GHP, GB and bivariate-bicycle families), either association:
        // streaming VALU kernel, one codeword per workgroup, one lane per qubit.  The
        // workgroup size minimises idle lanes (882 qubits: 7 passes of 128 threads, 1270: 5 passes of 256; 882 / 896 and 1270 / 1280
        // lanes busy) and, among equals, is the largest up to 256 threads (measured: 128 .. 256 best); few codewords take the widest
        // one (latency).
        const int n = g->d.n;
        int best_tpc = 0, best_cost = 1 << 30;
        for (int k = 1; k <= 32; ++k) {
            const int per = (n + k - 1) / k, tpc = (per + 63) & ~63;
            if (tpc > 1024 || tpc < 128) continue;
            const int cost = (tpc / 64) * k;  // wave-passes per codeword
            const bool better_tie = B <= 2048 ? tpc > best_tpc : (tpc <= 256 ? (best_tpc > 256 || tpc > best_tpc) : tpc < best_tpc);
            if (cost < best_cost || (cost == best_cost && better_tie)) { best_cost = cost; best_tpc = tpc; }
        }
        if (!best_tpc) best_tpc = 128;
#ifdef FGNN_GNNS_TPC
        best_tpc = FGNN_GNNS_TPC;
#endif
        const size_t lds_s = (size_t)a.lds_per_cw * sizeof(float);
        constexpr bool PF = FGNN_GNNS_EMBPK_FACT != 0, PL = FGNN_GNNS_EMBPK_LIT != 0;
        auto skern = g->gnn_factored ? (g->d.dvx == 3 ? gnn_stream_kernel<3, false, PF> : g->d.dvx == 4 ? gnn_stream_kernel<4, false, PF> : gnn_stream_kernel<5, false, PF>)
                                     : (g->d.dvx == 3 ? gnn_stream_kernel<3, true, PL> : g->d.dvx == 4 ? gnn_stream_kernel<4, true, PL> : gnn_stream_kernel<5, true, PL>);
        hipLaunchKernelGGL(skern, dim3((unsigned)B), dim3(best_tpc), lds_s, static_cast<hipStream_t>(stream), g->d, w->d, a);
        FGNN_HIP_CHECK(hipGetLastError());
        prof.done(FGNN_PROF_TAG_GNN, B);
        return FGNN_OK;
    }
    if (g->d.dvx == 3 && g->d.dvz == 3 && !g->force_generic) {
        // degree-regular graph: MFMA kernel, four waves per codeword, GNN_CPB codewords per workgroup
        constexpr int GNN_CPB = FGNN_GNN_CPB;
        // few codewords (a compacted feedback round at low p): one codeword's 56 tiles on four waves is 0.33 ms of latency on an
        // otherwise idle chip; dealt to up to 14 wave-quads it is one tile per wave
        const int ntiles = (g->d.n + 15) / 16;
        a.nsplit = 1;
        while (a.nsplit < 16 && (long long)B * a.nsplit * 2 <= 2048 && a.nsplit * 4 < ntiles) a.nsplit *= 2;
        const size_t lds_mfma = (size_t)(T_COUNT * 64 + GNN_CPB * a.lds_per_cw) * sizeof(float);
        auto kern = g->gnn_factored ? gnn_mfma_kernel<3, GNN_CPB, true> : gnn_mfma_kernel<3, GNN_CPB, false>;
        if (lds_mfma > 48 * 1024)
            FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds_mfma));
        hipLaunchKernelGGL(kern, dim3((unsigned)(((long long)B * a.nsplit + GNN_CPB - 1) / GNN_CPB)), dim3(256 * GNN_CPB), lds_mfma,
                           static_cast<hipStream_t>(stream), g->d, w->d, a);
        FGNN_HIP_CHECK(hipGetLastError());
        prof.done(FGNN_PROF_TAG_GNN, B);
        return FGNN_OK;
    }
    size_t lds_bytes = (size_t)a.lds_per_cw * sizeof(float) * (size_t)L.cpb;
    hipLaunchKernelGGL(g->gnn_factored ? gnn_kernel<true> : gnn_kernel<false>, dim3(L.blocks), dim3(L.threads), lds_bytes,
                       static_cast<hipStream_t>(stream), g->d, w->d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    prof.done(FGNN_PROF_TAG_GNN, B);
    return FGNN_OK;
}

extern "C" int fgnn_feedback_gnn(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                                 const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B, float* out,
                                 void* stream)
{
    return fgnn_feedback_gnn_impl(g, w, llr, logit_hx, logit_hz, synd_x, synd_z, B, out, nullptr, stream);
}
