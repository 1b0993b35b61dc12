// fgnn_gnnbp4.hip — the syndrome-only "full GNN" decoder GNN_BP4 (BASELINE.json configs[4]).
//
// Replaces GNN_BP4.call of /root/reference sionna/fec/ldpc/gnn.py:383-423 with UpdateCNEmbeddings.call
// (:573-610), UpdateVNEmbeddings.call (:714-751), cal_logit (:291-314) and make_hard_decision (:359-367)
// for num_mlp_layers = 2, activation tanh, reduce_op mean, use_bias True, D = num_embed_dims = 20,
// H = num_hidden_units = 40.  The reference raises as shipped (call unpacks 5 values from cal_logit's 4,
// :408 vs :314) and has no trained weights; the behaviour implemented is the oracle's (og_gnn_bp4), which
// repairs that line and restates the list-aliasing quirk (message width = num_embed_dims).
//
// One workgroup per codeword; node embeddings live in a caller-provided global workspace (203 KB per
// [[1270,28]] codeword — too large for LDS, but L2/MALL-resident while the workgroup runs) and are updated in
// place: a node's new embedding depends on its own old one and on the OTHER side's embeddings only.  One thread
// per receiving node runs the edge MLP (2D -> H tanh -> D) for each incoming edge, the mean, and the embed MLP,
// with wave-uniform weights arriving through scalar loads (rows made contiguous at upload).  Arithmetic order is
// the oracle's: fmaf chains in ascending k from 0, + bias, ascending-edge sums / count.
#include <algorithm>
#include <cstring>

#include "fgnn_internal.h"
#include "fgnn_math.h"
#include "fgnn_pk.h"

namespace {

constexpr int D = 20;
constexpr int H = 40;

struct MlpDev {
    const float* w1;   // [nin][H]      (W1 as Keras holds it, 32-byte aligned: row k = the weights input k feeds; gnn_bp4_stream_kernel)
    const float* w1t;  // [H][nin_pad]  (W1 transposed; row j = the weights of hidden unit j)
    const float* b1;   // [H]
    const float* w2;   // [H][D]
    const float* b2;   // [D]
};

struct GnnBp4Dev {
    MlpDev cn_msg[2], cn_embed[2], vn_msg[2], vn_embed;
    const float* winv;  // [D][4]
    const float* binv;  // [4]
    // MFMA path: per-lane operand tables, entry e at lane_tab[e*64 + lane]; tab_* = first entry of each MLP
    const float* lane_tab;
    int tab_cn_msg[2], tab_cn_embed[2], tab_vn_msg[2], tab_vn_embed, tab_inv;
};

template <int NIN, int NPAD>
__device__ __forceinline__ void mlp2(const float (&in)[NIN], const MlpDev& m, float (&out)[D])
{
#pragma unroll
    for (int i = 0; i < D; ++i) out[i] = 0.0f;
#pragma unroll 1
    for (int j = 0; j < H; ++j) {
        scalar_fp r = as_scalar(m.w1t) + j * NPAD;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) a = FG_FMA(in[k], r[k], a);
        const float h = fg_tanh(a + as_scalar(m.b1)[j]);
        scalar_fp r2 = as_scalar(m.w2) + j * D;
#pragma unroll
        for (int i = 0; i < D; ++i) out[i] = FG_FMA(h, r2[i], out[i]);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) out[i] = out[i] + as_scalar(m.b2)[i];
}

// The message MLP + (signed) mean of one receiving node and side in the FACTORED association (FGNN_OPT_GNN_FACTORED; oracle:
// msg_mean_factored): the own-embedding half of the first Dense and its bias once per node and side, every edge continues that value
// with its own fmaf chain over the neighbour's D elements; the hidden activations are summed (with the edge's syndrome sign on the
// qubit side) and ONE last Dense is applied to the sum: ((sum_e sg_e h_e) W2 + b2 sum_e sg_e) / deg.
// row(e) = the neighbour's embedding row, sgn(e) = +-1.
template <typename RowFn, typename SgnFn>
__device__ __forceinline__ void msg_mean_factored(const float (&own)[D], int deg, RowFn row, SgnFn sgn, const MlpDev& m, float (&mean)[D])
{
#pragma unroll
    for (int i = 0; i < D; ++i) mean[i] = 0.0f;
    float S = 0.0f;
    for (int e = 0; e < deg; ++e) { const float se = sgn(e); S = (e == 0) ? se : S + se; }
#pragma unroll 1
    for (int j = 0; j < H; ++j) {
        scalar_fp r = as_scalar(m.w1t) + j * (2 * D);
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < D; ++k) a = FG_FMA(own[k], r[D + k], a);
        const float pb = a + as_scalar(m.b1)[j];
        float hs = 0.0f;
        for (int e = 0; e < deg; ++e) {
            const float* src = row(e);
            float t = pb;
#pragma unroll
            for (int k = 0; k < D; ++k) t = FG_FMA(src[k], r[k], t);
            const float h = fg_tanh(t) * sgn(e);
            hs = (e == 0) ? h : hs + h;
        }
        scalar_fp r2 = as_scalar(m.w2) + j * D;
#pragma unroll
        for (int i = 0; i < D; ++i) mean[i] = FG_FMA(hs, r2[i], mean[i]);
    }
    if (deg > 0) {
        const float fd = (float)deg;
#pragma unroll
        for (int i = 0; i < D; ++i) mean[i] = FG_FMA(as_scalar(m.b2)[i], S, mean[i]) / fd;
    }
}

__device__ __forceinline__ float logit_row_gnn(const float* llr, const int* __restrict__ col, int deg)
{
    unsigned neg = 0;
    float T = 0.0f;
    for (int j = 0; j < deg; ++j) {
        float v = llr[col[j]];
        neg ^= (v < 0.0f);
        T = T + fg_phi_gnn(FG_ABS(v));
    }
    const float o = fg_phi_gnn(T);
    return neg ? -o : o;
}

// ---------------------------------------------------------------------------------------------
// MFMA path (degree-regular graphs).  Same construction as fgnn_gnn.hip: every Dense layer runs transposed on
// v_mfma_f32_16x16x4_f32 so that a layer's accumulators (column = node on the lane, 4 rows per register) are
// the next layer's B operand; weight rows are permuted at upload so that k-step s finds units 4s..4s+3 on lane
// groups 0..3.  A wave owns a tile of 16 receiving nodes.  Per MLP the table holds, in this order:
//   W1 [3 row tiles][S1 k-steps] | B1 [10] | W2 [2 row tiles][10 k-steps] | B2 [5]
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int S1>
__device__ __forceinline__ void mlp_tile(const float* __restrict__ lane_tab, int first, const float (&Bin)[S1], float (&out)[5])
{
    // Launder the table offset: the operand loads are loop-invariant, and hipcc would otherwise hoist a few hundred
    // of them out of the tile loops into registers and spill.  (Laundering the integer keeps the pointer's address
    // space; each operand is then one coalesced 256-byte L1/L2 load right before its MFMA.)
    int off = first * 64;
    asm volatile("" : "+v"(off));
    const float* tab = lane_tab + off;  // lane_tab points into LDS (the phase's tables are staged there)
    const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f4 d[3] = {zero, zero, zero};
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) d[t] = mfma4(tab[(t * S1 + s) * 64], Bin[s], d[t]);
    const float* b1 = tab + 3 * S1 * 64;
    float Hh[10];
#pragma unroll
    for (int s = 0; s < 10; ++s) Hh[s] = fg_tanh(d[s >> 2][s & 3] + b1[s * 64]);
    const float* w2 = b1 + 10 * 64;
    f4 m0 = zero, m1 = zero;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
        m0 = mfma4(w2[s * 64], Hh[s], m0);
        m1 = mfma4(w2[(10 + s) * 64], Hh[s], m1);
    }
    const float* b2 = w2 + 20 * 64;
    out[0] = m0[0] + b2[0 * 64];
    out[1] = m0[1] + b2[1 * 64];
    out[2] = m0[2] + b2[2 * 64];
    out[3] = m0[3] + b2[3 * 64];
    out[4] = m1[0] + b2[4 * 64];
}

// The message MLP of a tile of 16 receiving nodes in the FACTORED association, on the same operand tables (W1's k-steps 0..4 are
// the neighbour half, 5..9 the own half).  begin: P = own-half product through 15 MFMAs, + b1 -> the accumulators every edge starts
// from.  edge: 15 MFMAs continue those accumulators over the neighbour row, tanh, signed sum of the hidden activations.
// finish: ONE 40 -> 20 layer (20 MFMAs) on the sum, then ((.) + b2 S) / deg.
struct MsgTileFact {
    f4 pb[3];
    float hs[10];
    float S;
};
__device__ __forceinline__ void msg_fact_begin(const float* __restrict__ lane_tab, int first, const float (&own)[5], MsgTileFact& st)
{
    int off = first * 64;
    asm volatile("" : "+v"(off));
    const float* tab = lane_tab + off;
    const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f4 d[3] = {zero, zero, zero};
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) d[t] = mfma4(tab[(t * 10 + 5 + s) * 64], own[s], d[t]);
    const float* b1 = tab + 30 * 64;
#pragma unroll
    for (int s = 0; s < 10; ++s) st.pb[s >> 2][s & 3] = d[s >> 2][s & 3] + b1[s * 64];
    st.pb[2][2] = 0.0f;  // rows 40..47 of the third row tile are padding
    st.pb[2][3] = 0.0f;
    st.S = 0.0f;
}
template <bool FIRST>
__device__ __forceinline__ void msg_fact_edge(const float* __restrict__ lane_tab, int first, const float (&nbr)[5], float sg, MsgTileFact& st)
{
    int off = first * 64;
    asm volatile("" : "+v"(off));
    const float* tab = lane_tab + off;
    f4 d[3] = {st.pb[0], st.pb[1], st.pb[2]};
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) d[t] = mfma4(tab[(t * 10 + s) * 64], nbr[s], d[t]);
#pragma unroll
    for (int s = 0; s < 10; ++s) {
        const float h = fg_tanh(d[s >> 2][s & 3]) * sg;
        st.hs[s] = FIRST ? h : st.hs[s] + h;
    }
    st.S = FIRST ? sg : st.S + sg;
}
__device__ __forceinline__ void msg_fact_finish(const float* __restrict__ lane_tab, int first, int deg, const MsgTileFact& st, float (&mean)[5])
{
    int off = first * 64;
    asm volatile("" : "+v"(off));
    const float* w2 = lane_tab + off + 40 * 64;
    const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f4 m0 = zero, m1 = zero;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
        m0 = mfma4(w2[s * 64], st.hs[s], m0);
        m1 = mfma4(w2[(10 + s) * 64], st.hs[s], m1);
    }
    const float* b2 = w2 + 20 * 64;
    const float fd = (float)deg;
    const float mm[5] = {m0[0], m0[1], m0[2], m0[3], m1[0]};
#pragma unroll
    for (int i = 0; i < 5; ++i) mean[i] = FG_FMA(b2[i * 64], st.S, mm[i]) / fd;
}

// A node's D = 20 embedding floats as the 5 B-operand registers of lane group q: element 4s+q for s = 0..4.  The MFMA kernel keeps
// its workspace rows in that order — [q][s], element 4s+q at q*5+s — so a lane's five values are 20 contiguous bytes (one 16-byte and
// one 4-byte access instead of five strided ones).  The layout is private to the kernel: rows are only ever read and written here.
__device__ __forceinline__ void load_row5(const float* row, int q, float (&r)[5])
{
    const float* p = row + q * 5;
#pragma unroll
    for (int s = 0; s < 5; ++s) r[s] = p[s];
}
__device__ __forceinline__ void store_row5(float* row, int q, const float (&r)[5])
{
    float* p = row + q * 5;
#pragma unroll
    for (int s = 0; s < 5; ++s) p[s] = r[s];
}

struct Args {
    int B, num_iter;
    const uint8_t* synd_x;
    const uint8_t* synd_z;
    uint8_t* x_hat;
    uint8_t* z_hat;
    float* llr_out;   // [B,3,n]
    float* xlog_all;  // [num_iter,B,m_z+rows(lz)] or null
    float* zlog_all;  // [num_iter,B,m_x+rows(lx)] or null
    float* work;      // [B,(n+m)*D]
};

template <bool FACT>
__global__ void __launch_bounds__(256) gnn_bp4_kernel(GraphDev g, GnnBp4Dev w, Args a)
{
    FG_LOG_TAB_SETUP();
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const int n = g.n, mx = g.m_x, m = g.m;
    float* lx = lds;          // [n]   llr_x of cal_logit
    float* lz = lx + n;       // [n]   llr_z
    float* hlog = lz + n;     // [m]   hx_logit then hz_logit
    float* hv = a.work + (size_t)b * (size_t)(n + m) * D;
    float* hc = hv + (size_t)n * D;
    const uint8_t* sx = a.synd_x + (size_t)b * mx;
    const uint8_t* sz = a.synd_z + (size_t)b * g.m_z;
    const int rxp = g.m_z + g.rows[5], rzp = g.m_x + g.rows[4];
    for (int i = tid; i < n * D; i += T) hv[i] = 1.0f;  // (:396)
    for (int i = tid; i < m * D; i += T) hc[i] = 0.0f;  // (:392-393)
    for (int c = tid; c < m; c += T) hlog[c] = 0.0f;    // zero logits for the first CN update (:400-401)
    __syncthreads();
    float* llr = a.llr_out + (size_t)b * 3 * n;
    for (int it = -1; it < a.num_iter; ++it) {
        if (it >= 0) {
            // ---- UpdateVNEmbeddings (:714-751) + llr / binary LLRs of cal_logit (:291-304) ----
            for (int v = tid; v < n; v += T) {
                float own[D], feat3[3 * D];
#pragma unroll
                for (int i = 0; i < D; ++i) own[i] = hv[(size_t)v * D + i];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int* vptr = s ? g.vptr_z : g.vptr_x;
                    const int e0 = vptr[v], e1 = vptr[v + 1];
                    float acc[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) acc[i] = 0.0f;
                    if constexpr (FACT) {
                        msg_mean_factored(own, e1 - e0, [&](int e) { return hc + (size_t)((s ? mx : 0) + g.vchk[e0 + e]) * D; },
                                          [&](int e) { const int c = g.vchk[e0 + e]; return ((s ? sz[c] : sx[c]) & 1) ? -1.0f : 1.0f; },
                                          w.vn_msg[s], acc);
#pragma unroll
                        for (int i = 0; i < D; ++i) feat3[s * D + i] = acc[i];
                        continue;
                    }
                    for (int e = e0; e < e1; ++e) {
                        const int c = g.vchk[e];  // side-local check id
                        const float* src = hc + (size_t)((s ? mx : 0) + c) * D;
                        float feat[2 * D], msg[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) { feat[i] = src[i]; feat[D + i] = own[i]; }
                        mlp2<2 * D, 2 * D>(feat, w.vn_msg[s], msg);
                        const float sg = ((s ? sz[c] : sx[c]) & 1) ? -1.0f : 1.0f;
#pragma unroll
                        for (int i = 0; i < D; ++i) { const float mv = msg[i] * sg; acc[i] = (e == e0) ? mv : acc[i] + mv; }
                    }
                    if (e1 > e0) {
                        const float fd = (float)(e1 - e0);
#pragma unroll
                        for (int i = 0; i < D; ++i) acc[i] = acc[i] / fd;
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) feat3[s * D + i] = acc[i];
                }
#pragma unroll
                for (int i = 0; i < D; ++i) feat3[2 * D + i] = own[i];
                float nh[D];
                mlp2<3 * D, 3 * D>(feat3, w.vn_embed, nh);
                float L[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    hv[(size_t)v * D + k] = nh[k];
                    L[0] = FG_FMA(nh[k], as_scalar(w.winv)[k * 4 + 0], L[0]);
                    L[1] = FG_FMA(nh[k], as_scalar(w.winv)[k * 4 + 1], L[1]);
                    L[2] = FG_FMA(nh[k], as_scalar(w.winv)[k * 4 + 2], L[2]);
                }
                L[0] = L[0] + w.binv[0];
                L[1] = L[1] + w.binv[1];
                L[2] = L[2] + w.binv[2];
                llr[v] = L[0];
                llr[n + v] = L[1];
                llr[2 * n + v] = L[2];
                lz[v] = fg_softplus(-L[0]) - fg_lse2(-L[2], -L[1]);
                lx[v] = fg_softplus(-L[2]) - fg_lse2(-L[0], -L[1]);
            }
            __syncthreads();
            // ---- soft syndromes (:306-314): hx rows use llr_z, hz rows use llr_x; logical rows appended ----
            float* xl = a.xlog_all ? a.xlog_all + ((size_t)it * a.B + b) * rxp : nullptr;
            float* zl = a.zlog_all ? a.zlog_all + ((size_t)it * a.B + b) * rzp : nullptr;
            for (int c = tid; c < m; c += T) {
                const int p0 = g.cptr[c];
                const float vq = logit_row_gnn(c < mx ? lz : lx, g.cvn + p0, g.cptr[c + 1] - p0);
                hlog[c] = vq;
                if (c < mx) { if (zl) zl[c] = vq; }
                else if (xl) xl[c - mx] = vq;
            }
            if (xl)
                for (int r = tid; r < g.rows[5]; r += T)
                    xl[g.m_z + r] = logit_row_gnn(lx, g.rcol[5] + g.rptr[5][r], g.rptr[5][r + 1] - g.rptr[5][r]);
            if (zl)
                for (int r = tid; r < g.rows[4]; r += T)
                    zl[g.m_x + r] = logit_row_gnn(lz, g.rcol[4] + g.rptr[4][r], g.rptr[4][r + 1] - g.rptr[4][r]);
            __syncthreads();
            if (it == a.num_iter - 1) break;  // (:414-415)
        }
        // ---- UpdateCNEmbeddings (:573-610) ----
        for (int c = tid; c < m; c += T) {
            const int s = c >= mx;
            const int p0 = g.cptr[c], p1 = g.cptr[c + 1];
            float* hto = hc + (size_t)c * D;
            float own[D], acc[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { own[i] = hto[i]; acc[i] = 0.0f; }
            if constexpr (FACT)
                msg_mean_factored(own, p1 - p0, [&](int e) { return hv + (size_t)g.cvn[p0 + e] * D; }, [](int) { return 1.0f; },
                                  w.cn_msg[s], acc);
            else
            for (int jx = p0; jx < p1; ++jx) {
                const float* src = hv + (size_t)g.cvn[jx] * D;
                float feat[2 * D], msg[D];
#pragma unroll
                for (int i = 0; i < D; ++i) { feat[i] = src[i]; feat[D + i] = own[i]; }
                mlp2<2 * D, 2 * D>(feat, w.cn_msg[s], msg);
#pragma unroll
                for (int i = 0; i < D; ++i) acc[i] = (jx == p0) ? msg[i] : acc[i] + msg[i];
            }
            if (!FACT && p1 > p0) {
                const float fd = (float)(p1 - p0);
#pragma unroll
                for (int i = 0; i < D; ++i) acc[i] = acc[i] / fd;
            }
            float feat[2 * D + 1], nh[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { feat[i] = acc[i]; feat[D + i] = own[i]; }
            const unsigned sb = (s ? sz[c - mx] : sx[c]) & 1;
            feat[2 * D] = (it >= 0) ? hlog[c] * (sb ? -1.0f : 1.0f) : 0.0f;  // (:417-418)
            mlp2<2 * D + 1, 2 * D + 4>(feat, w.cn_embed[s], nh);
#pragma unroll
            for (int i = 0; i < D; ++i) hto[i] = nh[i];
        }
        __syncthreads();
    }
    for (int v = tid; v < n; v += T) {  // make_hard_decision (:359-367)
        const float X = llr[v], Y = llr[n + v], Z = llr[2 * n + v];
        int d = 0;
        float best = 0.0f;
        if (X < best) { best = X; d = 1; }
        if (Z < best) { best = Z; d = 2; }
        if (Y < best) { best = Y; d = 3; }
        a.x_hat[(size_t)b * n + v] = (uint8_t)(d & 1);
        a.z_hat[(size_t)b * n + v] = (uint8_t)(d >> 1);
    }
}

#ifndef FGNN_GNNBP4_THREADS
// threads per workgroup (= codeword) of the MFMA kernel and waves per SIMD the registers are allocated for.  1 024 x 4: one workgroup
// per CU, 115 VGPRs, no spills, and the 80 tiles of a [[1270,28]] phase divide evenly over 16 waves (195 ms per 16 384 x 10; 768 threads
// x 2 workgroups = 6 waves per SIMD spill 49 VGPRs and leave 4 of 84 tile slots empty: 201 ms; 512 x 2: 210 ms)
#define FGNN_GNNBP4_THREADS 1024
#endif
#ifndef FGNN_GNNBP4_MINW
#define FGNN_GNNBP4_MINW 4
#endif
template <int DV, int DC, bool FACT>
__global__ void __launch_bounds__(FGNN_GNNBP4_THREADS, FGNN_GNNBP4_MINW)
gnn_bp4_mfma_kernel(GraphDev g, GnnBp4Dev w, Args a, int tab_floats, int resident)
{
    FG_LOG_TAB_SETUP();
    // LDS: [tab_floats] the per-lane operand tables (an operand read then costs an LDS access instead of an L2 round trip, which
    // is what the waves were waiting on), then lx | lz | hlog | ssg.  resident: the tables of BOTH phases are staged once at kernel
    // start (106 KB + 20 KB for [[1270,28]]: one workgroup per CU); otherwise the running phase's tables are staged at every phase
    // start into one shared region.
    extern __shared__ float lds[];
    constexpr int T = FGNN_GNNBP4_THREADS, NW = T / 64;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = g.n, mx = g.m_x, mz = g.m_z, m = g.m;
    float* tabs = lds;
    float* lx = lds + tab_floats;
    float* lz = lx + n;
    float* hlog = lz + n;
    float* ssg = hlog + m;  // [m] syndrome signs 1 - 2 s as floats (hx checks, then hz): read once from HBM, not once per edge
    float* hv = a.work + (size_t)b * (size_t)(n + m) * D;
    float* hc = hv + (size_t)n * D;
    const uint8_t* sx = a.synd_x + (size_t)b * mx;
    const uint8_t* sz = a.synd_z + (size_t)b * mz;
    const int rxp = mz + g.rows[5], rzp = mx + g.rows[4];
    for (int i = tid; i < n * D; i += T) hv[i] = 1.0f;
    for (int i = tid; i < m * D; i += T) hc[i] = 0.0f;
    for (int c = tid; c < m; c += T) {
        hlog[c] = 0.0f;
        ssg[c] = ((c < mx ? sx[c] : sz[c - mx]) & 1) ? -1.0f : 1.0f;
    }
    float* llr = a.llr_out + (size_t)b * 3 * n;
    const int l = tid & 63, wave = tid >> 6, j = l & 15, q = l >> 4;
    const float* tab_cn = tabs + l;                                                    // check-phase tables
    const float* tab = tab_cn + (resident ? (w.tab_vn_msg[0] - w.tab_cn_msg[0]) * 64 : 0);  // qubit-phase tables
    const int vtiles = (n + 15) >> 4, xtiles = (mx + 15) >> 4, ztiles = (mz + 15) >> 4;
    // phase tables are contiguous in the global table: CN phase = [tab_cn_msg[0], tab_vn_msg[0]), VN phase = the rest
    const int cn_first = w.tab_cn_msg[0], vn_first = w.tab_vn_msg[0], vn_end = w.tab_inv + 8;
    if (resident)
        for (int i = tid; i < (vn_end - cn_first) * 64; i += T) tabs[i] = w.lane_tab[(size_t)cn_first * 64 + i];
    for (int it = -1; it < a.num_iter; ++it) {
        if (it >= 0) {
            __syncthreads();
            if (!resident) {
                for (int i = tid; i < (vn_end - vn_first) * 64; i += T) tabs[i] = w.lane_tab[(size_t)vn_first * 64 + i];
                __syncthreads();
            }
            // ---- UpdateVNEmbeddings on tiles of 16 qubits ----
            for (int tile = wave; tile < vtiles; tile += NW) {
                const int vraw = tile * 16 + j;
                const bool valid = vraw < n;
                const int v = valid ? vraw : n - 1;
                float own[5], Bemb[15];
                load_row5(hv + (size_t)v * D, q, own);
                // all 2 x DV neighbour rows of the tile are requested before the first MLP starts: the gathers (L2-resident rows of
                // other workgroups' making) then complete under ~1 600 cycles of MFMA work each instead of in front of it
                // neighbour rows one edge ahead: while the MLP of edge e runs (~1 600 cycles of MFMA work) the gather of edge e + 1
                // (an L2-resident row of another workgroup's making) is in flight; all 2 x DV check ids are read up front
                int cn_of[2][DV];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int k = 0; k < DV; ++k) cn_of[s2][k] = g.vchk[(s2 ? g.E_x : 0) + v * DV + k];
                float fcur[5], fnxt[5];
                load_row5(hc + (size_t)cn_of[0][0] * D, q, fcur);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float acc[5];
                    if constexpr (FACT) {
                        MsgTileFact st;
                        msg_fact_begin(tab, w.tab_vn_msg[s2] - vn_first, own, st);
#pragma unroll
                        for (int k = 0; k < DV; ++k) {
                            const int c = cn_of[s2][k];
                            const int e1 = s2 * DV + k + 1;  // next edge, flattened over both sides
                            if (e1 < 2 * DV) load_row5(hc + (size_t)((e1 / DV ? mx : 0) + cn_of[e1 / DV][e1 % DV]) * D, q, fnxt);
                            const float sg = ssg[(s2 ? mx : 0) + c];
                            if (k == 0) msg_fact_edge<true>(tab, w.tab_vn_msg[s2] - vn_first, fcur, sg, st);
                            else msg_fact_edge<false>(tab, w.tab_vn_msg[s2] - vn_first, fcur, sg, st);
#pragma unroll
                            for (int s = 0; s < 5; ++s) fcur[s] = fnxt[s];
                        }
                        msg_fact_finish(tab, w.tab_vn_msg[s2] - vn_first, DV, st, acc);
#pragma unroll
                        for (int i = 0; i < 5; ++i) Bemb[s2 * 5 + i] = acc[i];
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < DV; ++k) {
                        const int c = cn_of[s2][k];
                        float Bin[10], msg[5];
                        {
                            const int e1 = s2 * DV + k + 1;  // next edge, flattened over both sides
                            if (e1 < 2 * DV) load_row5(hc + (size_t)((e1 / DV ? mx : 0) + cn_of[e1 / DV][e1 % DV]) * D, q, fnxt);
                        }
#pragma unroll
                        for (int s = 0; s < 5; ++s) { Bin[s] = fcur[s]; Bin[5 + s] = own[s]; }
#pragma unroll
                        for (int s = 0; s < 5; ++s) fcur[s] = fnxt[s];
                        mlp_tile<10>(tab, w.tab_vn_msg[s2] - vn_first, Bin, msg);
                        const float sg = ssg[(s2 ? mx : 0) + c];
#pragma unroll
                        for (int i = 0; i < 5; ++i) { const float mv = msg[i] * sg; acc[i] = (k == 0) ? mv : acc[i] + mv; }
                    }
#pragma unroll
                    for (int i = 0; i < 5; ++i) Bemb[s2 * 5 + i] = acc[i] / (float)DV;
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) Bemb[10 + i] = own[i];
                float nh[5];
                mlp_tile<15>(tab, w.tab_vn_embed - vn_first, Bemb, nh);
                // embed_to_llr: L^T = Winv^T nh^T (rows 0..2 land on lane group 0, registers 0..2)
                const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                f4 o = zero;
                int offi = (w.tab_inv - vn_first) * 64;
                asm volatile("" : "+v"(offi));
                const float* ti = tab + offi;
#pragma unroll
                for (int s = 0; s < 5; ++s) o = mfma4(ti[s * 64], nh[s], o);
                if (valid) {
                    store_row5(hv + (size_t)v * D, q, nh);
                    if (q == 0) {
                        const float L0 = o[0] + ti[5 * 64], L1 = o[1] + ti[6 * 64], L2 = o[2] + ti[7 * 64];
                        llr[v] = L0;
                        llr[n + v] = L1;
                        llr[2 * n + v] = L2;
                        lz[v] = fg_softplus(-L0) - fg_lse2(-L2, -L1);
                        lx[v] = fg_softplus(-L2) - fg_lse2(-L0, -L1);
                    }
                }
            }
            __syncthreads();
            float* xl = a.xlog_all ? a.xlog_all + ((size_t)it * a.B + b) * rxp : nullptr;
            float* zl = a.zlog_all ? a.zlog_all + ((size_t)it * a.B + b) * rzp : nullptr;
            for (int c = tid; c < m; c += T) {
                const int p0 = g.cptr[c];
                const float vq = logit_row_gnn(c < mx ? lz : lx, g.cvn + p0, g.cptr[c + 1] - p0);
                hlog[c] = vq;
                if (c < mx) { if (zl) zl[c] = vq; }
                else if (xl) xl[c - mx] = vq;
            }
            if (xl)
                for (int r = tid; r < g.rows[5]; r += T)
                    xl[mz + r] = logit_row_gnn(lx, g.rcol[5] + g.rptr[5][r], g.rptr[5][r + 1] - g.rptr[5][r]);
            if (zl)
                for (int r = tid; r < g.rows[4]; r += T)
                    zl[mx + r] = logit_row_gnn(lz, g.rcol[4] + g.rptr[4][r], g.rptr[4][r + 1] - g.rptr[4][r]);
            if (it == a.num_iter - 1) break;
        }
        __syncthreads();
        if (!resident) {
            for (int i = tid; i < (vn_first - cn_first) * 64; i += T) tabs[i] = w.lane_tab[(size_t)cn_first * 64 + i];
            __syncthreads();
        }
        // ---- UpdateCNEmbeddings: tiles never mix hx and hz checks (the two sides use different weights) ----
        for (int tile = wave; tile < xtiles + ztiles; tile += NW) {
            const int s2 = tile >= xtiles;
            const int local = (s2 ? tile - xtiles : tile) * 16 + j;
            const int cnt = s2 ? mz : mx;
            const bool valid = local < cnt;
            const int c = (s2 ? mx : 0) + (valid ? local : cnt - 1);  // combined check id
            float own[5], acc[5], Bemb[11];
            load_row5(hc + (size_t)c * D, q, own);
            int vn_of[DC];
#pragma unroll
            for (int k = 0; k < DC; ++k) vn_of[k] = g.cvn[c * DC + k];
            float fcur[5], fnxt[5];  // neighbour rows one edge ahead (see the qubit phase)
            load_row5(hv + (size_t)vn_of[0] * D, q, fcur);
            if constexpr (FACT) {
                MsgTileFact st;
                msg_fact_begin(tab_cn, w.tab_cn_msg[s2] - cn_first, own, st);
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    if (k + 1 < DC) load_row5(hv + (size_t)vn_of[k + 1] * D, q, fnxt);
                    if (k == 0) msg_fact_edge<true>(tab_cn, w.tab_cn_msg[s2] - cn_first, fcur, 1.0f, st);
                    else msg_fact_edge<false>(tab_cn, w.tab_cn_msg[s2] - cn_first, fcur, 1.0f, st);
#pragma unroll
                    for (int s = 0; s < 5; ++s) fcur[s] = fnxt[s];
                }
                msg_fact_finish(tab_cn, w.tab_cn_msg[s2] - cn_first, DC, st, acc);
            } else {
#pragma unroll
            for (int k = 0; k < DC; ++k) {
                float Bin[10], msg[5];
                if (k + 1 < DC) load_row5(hv + (size_t)vn_of[k + 1] * D, q, fnxt);
#pragma unroll
                for (int s = 0; s < 5; ++s) { Bin[s] = fcur[s]; Bin[5 + s] = own[s]; }
#pragma unroll
                for (int s = 0; s < 5; ++s) fcur[s] = fnxt[s];
                mlp_tile<10>(tab_cn, w.tab_cn_msg[s2] - cn_first, Bin, msg);
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = (k == 0) ? msg[i] : acc[i] + msg[i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = acc[i] / (float)DC;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) { Bemb[i] = acc[i]; Bemb[5 + i] = own[i]; }
            const float lg = (it >= 0) ? hlog[c] * ssg[c] : 0.0f;
            Bemb[10] = (q == 0) ? lg : 0.0f;
            float nh[5];
            mlp_tile<11>(tab_cn, w.tab_cn_embed[s2] - cn_first, Bemb, nh);
            if (valid) store_row5(hc + (size_t)c * D, q, nh);
        }
    }
    __syncthreads();
    for (int v = tid; v < n; v += T) {
        const float X = llr[v], Y = llr[n + v], Z = llr[2 * n + v];
        int d = 0;
        float best = 0.0f;
        if (X < best) { best = X; d = 1; }
        if (Z < best) { best = Z; d = 2; }
        if (Y < best) { best = Y; d = 3; }
        a.x_hat[(size_t)b * n + v] = (uint8_t)(d & 1);
        a.z_hat[(size_t)b * n + v] = (uint8_t)(d >> 1);
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming path (degree-regular graphs, FGNN_OPT_GNN_STREAM): no MFMA, no operand tables.  One lane per receiving node, the node's
// vectors in registers, every weight a wave-uniform scalar — and every multiply-add a v_pk_fma_f32: TWO hidden units (or two output
// elements) per instruction, the weights of the pair as one SGPR pair, the per-lane input broadcast to both halves by op_sel.  On
// gfx950 that instruction issues in the time of ONE SGPR-operand v_fmac (tools/microbench: 1.85-1.93 ns per wave-instruction and SIMD
// against 1.79), i.e. the uniform-weight multiply-adds run at the full f32 rate of 64 FLOP per clock and SIMD — the rate of the
// f32 MFMA — without the MFMA tiles' padding of 40-, 20- and 3-row outputs to multiples of 16 (76 % useful rows in this layer stack).
// A Dense layer runs k-outer / j-inner: for input k = 0, 1, ... one packed fma per PAIR of output units continues that pair's chain,
// so every output sees fmaf(in[k], W[k][j], acc) in ascending k from the oracle's start value — the oracle's bits — while consecutive
// instructions are independent (no packed-math dependency stalls).  Hidden units are walked in five blocks of eight (four pairs): the
// block's first-layer accumulators, its tanh values and the 20 second-layer accumulators they stream into are all that is live.
constexpr int HB = 8, HBP = HB / 2, NHB = H / HB;  // hidden units per block, pairs per block, blocks
constexpr int DP = D / 2;                          // an embedding / message vector as pairs of consecutive elements
static_assert(H % HB == 0 && D % 4 == 0, "block / pair structure of the streaming GNN_BP4 kernel");

constexpr int KG1 = 4;  // first-layer rows (8 floats of a block) per group: 2 x 32 SGPRs in flight
constexpr int KG2 = 1;  // last-layer rows (20 floats) per group: 2 x 20 SGPRs

__device__ __forceinline__ void load_row20(const float* row, f2 (&r)[DP])
{
    const float4* p = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int q = 0; q < D / 4; ++q) {
        const float4 v = p[q];
        r[2 * q] = f2{v.x, v.y};
        r[2 * q + 1] = f2{v.z, v.w};
    }
}
__device__ __forceinline__ void store_row20(float* row, const f2 (&r)[DP])
{
    float4* p = reinterpret_cast<float4*>(row);
#pragma unroll
    for (int q = 0; q < D / 4; ++q) p[q] = make_float4(r[2 * q].x, r[2 * q].y, r[2 * q + 1].x, r[2 * q + 1].y);
}

// A whole two-layer MLP (oracle: mlp2) on a register vector of NIN elements: out = tanh(in W1 + b1) W2 + b2, hidden units in blocks of eight.
template <int NIN>
__device__ __forceinline__ void mlp_stream(const f2 (&in2)[(NIN + 1) / 2], const MlpDev& m, f2 (&out)[DP])
{
#pragma unroll
    for (int i = 0; i < DP; ++i) out[i] = bc2(0.0f);
#pragma unroll 1
    for (int blk = 0; blk < NHB; ++blk) {
        f2 t[HBP];
#pragma unroll
        for (int jp = 0; jp < HBP; ++jp) t[jp] = bc2(0.0f);
        dense_pk<NIN, HBP, KG1>(in2, m.w1 + blk * HB, H, t);
        scalar_f2p b1 = as_scalar2(m.b1 + blk * HB);
        f2 h[HBP];
#pragma unroll
        for (int jp = 0; jp < HBP; ++jp) h[jp] = tanh2(t[jp] + b1[jp]);
        dense_pk<HB, DP, KG2>(h, m.w2 + blk * HB * D, D, out);
    }
    scalar_f2p b2 = as_scalar2(m.b2);
#pragma unroll
    for (int i = 0; i < DP; ++i) out[i] = out[i] + b2[i];
}

// The message MLP + (signed) mean of one receiving node and side in the FACTORED association (oracle: msg_mean_factored) for a node
// of DEG edges: nbr(e, sg) = the embedding row of edge e's other end and its sign +-1.  Per block of eight hidden units: pb = own
// W1[D:2D] + b1 once, every edge continues pb over its neighbour row, tanh, signed sum; the block's eight sums stream into the ONE last
// Dense.  The edge loop is a real loop with the next edge's row requested one edge ahead (an L2-resident row of another lane's making).
template <int DEG, bool SIGNED, typename NbrFn>
__device__ __forceinline__ void msg_side_stream(const f2 (&own)[DP], NbrFn nbr, const MlpDev& m, f2 (&mean)[DP])
{
#pragma unroll
    for (int i = 0; i < DP; ++i) mean[i] = bc2(0.0f);
    float S = 0.0f;
#pragma unroll 1
    for (int blk = 0; blk < NHB; ++blk) {
        f2 pb[HBP];
#pragma unroll
        for (int jp = 0; jp < HBP; ++jp) pb[jp] = bc2(0.0f);
        dense_pk<D, HBP, KG1>(own, m.w1 + D * H + blk * HB, H, pb);
        scalar_f2p b1 = as_scalar2(m.b1 + blk * HB);
#pragma unroll
        for (int jp = 0; jp < HBP; ++jp) pb[jp] = pb[jp] + b1[jp];
        f2 hs[HBP];
#pragma unroll
        for (int jp = 0; jp < HBP; ++jp) hs[jp] = bc2(0.0f);
        f2 src[DP];
        float sg;
        load_row20(nbr(0, sg), src);
#pragma unroll 1
        for (int e = 0; e < DEG; ++e) {
            f2 nxt[DP];
            float sgn = 1.0f;
            if (e + 1 < DEG) load_row20(nbr(e + 1, sgn), nxt);
            f2 t[HBP];
#pragma unroll
            for (int jp = 0; jp < HBP; ++jp) t[jp] = pb[jp];
            dense_pk<D, HBP, KG1>(src, m.w1 + blk * HB, H, t);
#pragma unroll
            for (int jp = 0; jp < HBP; ++jp) {
                const f2 h = tanh2(t[jp]);
                // oracle: h sg, then first ? that : acc + that.  h sg is exact (sg = +-1), so the fma rounds once like the add
                const f2 first = SIGNED ? h * bc2(sg) : h;
                const f2 later = SIGNED ? pk_fma(h, bc2(sg), hs[jp]) : hs[jp] + h;
                hs[jp] = (e == 0) ? first : later;
            }
            S = (e == 0) ? sg : S + sg;
#pragma unroll
            for (int i = 0; i < DP; ++i) src[i] = nxt[i];
            sg = sgn;
        }
        dense_pk<HB, DP, KG2>(hs, m.w2 + blk * HB * D, D, mean);
    }
    scalar_f2p b2 = as_scalar2(m.b2);
    const float fd = (float)DEG;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        const f2 o = pk_fma(b2[i], bc2(S), mean[i]);
        mean[i] = f2{o.x / fd, o.y / fd};
    }
}

// The same in the LITERAL association: one whole message MLP per edge, (signed) sum, / DEG.
template <int DEG, bool SIGNED, typename NbrFn>
__device__ __forceinline__ void msg_side_literal(const f2 (&own)[DP], NbrFn nbr, const MlpDev& m, f2 (&mean)[DP])
{
#pragma unroll
    for (int i = 0; i < DP; ++i) mean[i] = bc2(0.0f);
#pragma unroll 1
    for (int e = 0; e < DEG; ++e) {
        f2 feat[D], msg[DP];
        float sg;
        {
            f2 src[DP];
            load_row20(nbr(e, sg), src);
#pragma unroll
            for (int i = 0; i < DP; ++i) { feat[i] = src[i]; feat[DP + i] = own[i]; }
        }
        mlp_stream<2 * D>(feat, m, msg);
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            const f2 mv = SIGNED ? msg[i] * bc2(sg) : msg[i];
            mean[i] = (e == 0) ? mv : mean[i] + mv;
        }
    }
    const float fd = (float)DEG;
#pragma unroll
    for (int i = 0; i < DP; ++i) mean[i] = f2{mean[i].x / fd, mean[i].y / fd};
}

// one of two weight sets by a wave-uniform index, field by field (scalar selects: indexing the by-value kernel argument with a loop
// counter would send the struct through private memory and the pointers through VGPRs)
__device__ __forceinline__ MlpDev pick_mlp(const MlpDev& a, const MlpDev& b, int which)
{
    MlpDev m;
    m.w1 = which ? b.w1 : a.w1;
    m.w1t = which ? b.w1t : a.w1t;
    m.b1 = which ? b.b1 : a.b1;
    m.w2 = which ? b.w2 : a.w2;
    m.b2 = which ? b.b2 : a.b2;
    return m;
}

#ifndef FGNN_GNNBP4_STREAM_WAVES
#define FGNN_GNNBP4_STREAM_WAVES 3  // waves per SIMD the registers are budgeted for (168 VGPRs: own, two means, a block's state, two rows)
#endif
template <int DV, int DC, bool FACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FGNN_GNNBP4_STREAM_WAVES, FGNN_GNNBP4_STREAM_WAVES)))
gnn_bp4_stream_kernel(GraphDev g, GnnBp4Dev w, Args a)
{
    FG_LOG_TAB_SETUP();
    extern __shared__ float lds[];
    constexpr int T = 256;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = g.n, mx = g.m_x, mz = g.m_z, m = g.m;
    float* lx = lds;       // [n] llr_x of cal_logit
    float* lz = lx + n;    // [n] llr_z
    float* hlog = lz + n;  // [m] hx_logit then hz_logit
    float* ssg = hlog + m; // [m] syndrome signs 1 - 2 s as floats
    float* hv = a.work + (size_t)b * (size_t)(n + m) * D;  // rows in natural element order (private to this kernel)
    float* hc = hv + (size_t)n * D;
    const uint8_t* sx = a.synd_x + (size_t)b * mx;
    const uint8_t* sz = a.synd_z + (size_t)b * mz;
    const int rxp = mz + g.rows[5], rzp = mx + g.rows[4];
    for (int i = tid; i < n * D; i += T) hv[i] = 1.0f;  // (:396)
    for (int i = tid; i < m * D; i += T) hc[i] = 0.0f;  // (:392-393)
    for (int c = tid; c < m; c += T) {
        hlog[c] = 0.0f;  // zero logits for the first CN update (:400-401)
        ssg[c] = ((c < mx ? sx[c] : sz[c - mx]) & 1) ? -1.0f : 1.0f;
    }
    __syncthreads();
    float* llr = a.llr_out + (size_t)b * 3 * n;
    for (int it = -1; it < a.num_iter; ++it) {
        if (it >= 0) {
            // ---- UpdateVNEmbeddings (:714-751) + llr / binary LLRs of cal_logit (:291-304) ----
            for (int v = tid; v < n; v += T) {
                f2 feat3[3 * DP];
                {
                    f2 own[DP];
                    load_row20(hv + (size_t)v * D, own);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f2 mean[DP];
                        const int* chk = g.vchk + (s ? g.E_x : 0) + v * DV;
                        const int cbase = s ? mx : 0;
                        auto nbr = [&](int e, float& sg) {
                            const int c = cbase + chk[e];
                            sg = ssg[c];
                            return hc + (size_t)c * D;
                        };
                        if constexpr (FACT) msg_side_stream<DV, true>(own, nbr, w.vn_msg[s], mean);
                        else msg_side_literal<DV, true>(own, nbr, w.vn_msg[s], mean);
#pragma unroll
                        for (int i = 0; i < DP; ++i) feat3[s * DP + i] = mean[i];
                    }
#pragma unroll
                    for (int i = 0; i < DP; ++i) feat3[2 * DP + i] = own[i];
                }
                f2 nh[DP];
                mlp_stream<3 * D>(feat3, w.vn_embed, nh);
                store_row20(hv + (size_t)v * D, nh);
                f2 Lp[2] = {bc2(0.0f), bc2(0.0f)};
                dense_pk<D, 2, 5>(nh, w.winv, 4, Lp);
                const float L0 = Lp[0].x + as_scalar(w.binv)[0], L1 = Lp[0].y + as_scalar(w.binv)[1], L2 = Lp[1].x + as_scalar(w.binv)[2];
                llr[v] = L0;
                llr[n + v] = L1;
                llr[2 * n + v] = L2;
                lz[v] = fg_softplus(-L0) - fg_lse2(-L2, -L1);
                lx[v] = fg_softplus(-L2) - fg_lse2(-L0, -L1);
            }
            __syncthreads();
            // ---- soft syndromes (:306-314): hx rows use llr_z, hz rows use llr_x; logical rows appended ----
            float* xl = a.xlog_all ? a.xlog_all + ((size_t)it * a.B + b) * rxp : nullptr;
            float* zl = a.zlog_all ? a.zlog_all + ((size_t)it * a.B + b) * rzp : nullptr;
            for (int c = tid; c < m; c += T) {
                const int p0 = g.cptr[c];
                const float vq = logit_row_gnn(c < mx ? lz : lx, g.cvn + p0, g.cptr[c + 1] - p0);
                hlog[c] = vq;
                if (c < mx) { if (zl) zl[c] = vq; }
                else if (xl) xl[c - mx] = vq;
            }
            if (xl)
                for (int r = tid; r < g.rows[5]; r += T)
                    xl[mz + r] = logit_row_gnn(lx, g.rcol[5] + g.rptr[5][r], g.rptr[5][r + 1] - g.rptr[5][r]);
            if (zl)
                for (int r = tid; r < g.rows[4]; r += T)
                    zl[mx + r] = logit_row_gnn(lz, g.rcol[4] + g.rptr[4][r], g.rptr[4][r + 1] - g.rptr[4][r]);
            __syncthreads();
            if (it == a.num_iter - 1) break;  // (:414-415)
        }
        // ---- UpdateCNEmbeddings (:573-610) ----
        for (int c = tid; c < m; c += T) {
            const int s = c >= mx;
            f2 feat[DP + DP + 1];
            {
                f2 own[DP], mean[DP];
                load_row20(hc + (size_t)c * D, own);
                const int* vns = g.cvn + c * DC;
                auto nbr = [&](int e, float& sg) {
                    sg = 1.0f;
                    return hv + (size_t)vns[e] * D;
                };
                // a wave of 64 checks may straddle the hx / hz boundary, and the two sides have their own weights: each side's update
                // runs under its own lane mask with wave-uniform (scalar) weights; a wave that holds one side skips the other pass
                // (readfirstlane: inside `s == ss` the compiler substitutes the per-lane s for the loop counter, and the weight
                // pointers selected by it would travel through VGPRs and vector loads)
#pragma unroll 1
                for (int ss = 0; ss < 2; ++ss)
                    if (s == ss) {
                        const MlpDev mm = pick_mlp(w.cn_msg[0], w.cn_msg[1], __builtin_amdgcn_readfirstlane(ss));
                        if constexpr (FACT) msg_side_stream<DC, false>(own, nbr, mm, mean);
                        else msg_side_literal<DC, false>(own, nbr, mm, mean);
                    }
#pragma unroll
                for (int i = 0; i < DP; ++i) { feat[i] = mean[i]; feat[DP + i] = own[i]; }
            }
            feat[2 * DP] = f2{(it >= 0) ? hlog[c] * ssg[c] : 0.0f, 0.0f};  // (:417-418)
            f2 nh[DP];
#pragma unroll 1
            for (int ss = 0; ss < 2; ++ss)
                if (s == ss) mlp_stream<2 * D + 1>(feat, pick_mlp(w.cn_embed[0], w.cn_embed[1], __builtin_amdgcn_readfirstlane(ss)), nh);
            store_row20(hc + (size_t)c * D, nh);
        }
        __syncthreads();
    }
    for (int v = tid; v < n; v += T) {  // make_hard_decision (:359-367)
        const float X = llr[v], Y = llr[n + v], Z = llr[2 * n + v];
        int d = 0;
        float best = 0.0f;
        if (X < best) { best = X; d = 1; }
        if (Z < best) { best = Z; d = 2; }
        if (Y < best) { best = Y; d = 3; }
        a.x_hat[(size_t)b * n + v] = (uint8_t)(d & 1);
        a.z_hat[(size_t)b * n + v] = (uint8_t)(d >> 1);
    }
}

// ---------------------------------------------------------------------------------------------
// Runtime-shaped GNN_BP4 (fgnn_gnnbp4_weights_create_general): any num_embed_dims / num_hidden_units / num_mlp_layers / reduce_op /
// activation / use_bias / use_attributes setting the reference's classes accept (gnn.py:131-207, :494-751) within the limits of
// fgnn.h.  One thread per receiving node, Dense = fmaf chain in ascending k from 0, (+ bias), activation — the oracle's
// og_gnn_bp4_general, always the literal association.  Activations ping-pong between per-thread buffers (scratch memory: this is
// the compatibility path, not the benchmark path).
// ---------------------------------------------------------------------------------------------
constexpr int GG_MAXW = 128, GG_MAXD = 32;
struct GnnBp4GenDev {
    int D, H, L, rop, act, bias, An, Am;
    const float* W[7][4];
    const float* b[7][4];
    int K[7][4], J[7][4], actl[7][4];
    const float* winv;  // [D][3]
    const float* binv;  // [3] or null
    const float* cn_node[2];  // [m_s][An]
    const float* cn_msga[2];  // [E_s][Am], check-major (= the reference's np.where(pcm) edge order)
    const float* vn_node;     // [n][An]
    const float* vn_msga[2];  // [E_s][Am], permuted to the VN-major slot order at upload
};

__device__ __forceinline__ float gg_act(float a, int act)
{
    switch (act) {
    case FGNN_ACT_TANH: return fg_tanh(a);
    case FGNN_ACT_RELU: return FG_MAX(a, 0.0f);
    case FGNN_ACT_SIGMOID: return fg_sigmoid(a);
    default: return a;
    }
}

__device__ __forceinline__ void gg_run(const GnnBp4GenDev& w, int q, const float* in, float* out, float* bufA, float* bufB)
{
    const float* cur = in;
    for (int k = 0; k < w.L; ++k) {
        float* nxt = (k == w.L - 1) ? out : ((k & 1) ? bufB : bufA);
        const int K = w.K[q][k], J = w.J[q][k], act = w.actl[q][k];
        const float* W = w.W[q][k];
        const float* bb = w.b[q][k];
        for (int j = 0; j < J; ++j) {
            float a = 0.0f;
            for (int kk = 0; kk < K; ++kk) a = FG_FMA(cur[kk], W[kk * J + j], a);
            if (bb) a = a + bb[j];
            nxt[j] = gg_act(a, act);
        }
        cur = nxt;
    }
}

__device__ __forceinline__ void gg_reduce(float* acc, const float* msg, int D, bool first, int op)
{
    for (int i = 0; i < D; ++i) {
        const float m = msg[i];
        float r;
        if (first) r = m;
        else if (op == FGNN_REDUCE_MAX) r = FG_MAX(acc[i], m);
        else if (op == FGNN_REDUCE_MIN) r = FG_MIN(acc[i], m);
        else r = acc[i] + m;
        acc[i] = r;
    }
}

__global__ void __launch_bounds__(256) gnn_bp4_general_kernel(GraphDev g, GnnBp4GenDev w, Args a)
{
    FG_LOG_TAB_SETUP();
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const int n = g.n, mx = g.m_x, m = g.m, D = w.D, An = w.An, Am = w.Am;
    float* lx = lds;
    float* lz = lx + n;
    float* hlog = lz + n;
    float* hv = a.work + (size_t)b * (size_t)(n + m) * D;
    float* hc = hv + (size_t)n * D;
    const uint8_t* sx = a.synd_x + (size_t)b * mx;
    const uint8_t* sz = a.synd_z + (size_t)b * g.m_z;
    const int rxp = g.m_z + g.rows[5], rzp = g.m_x + g.rows[4];
    for (int i = tid; i < n * D; i += T) hv[i] = 1.0f;  // (:396)
    for (int i = tid; i < m * D; i += T) hc[i] = 0.0f;  // (:392-393)
    for (int c = tid; c < m; c += T) hlog[c] = 0.0f;
    __syncthreads();
    float* llr = a.llr_out + (size_t)b * 3 * n;
    float feat[GG_MAXW], msg[GG_MAXD], bufA[GG_MAXW], bufB[GG_MAXW], acc[2][GG_MAXD];
    for (int it = -1; it < a.num_iter; ++it) {
        if (it >= 0) {
            // ---- UpdateVNEmbeddings.call (:714-751) ----
            for (int v = tid; v < n; v += T) {
                for (int s = 0; s < 2; ++s) {
                    const int* vptr = s ? g.vptr_z : g.vptr_x;
                    const int e0 = vptr[v], e1 = vptr[v + 1];
                    for (int i = 0; i < D; ++i) acc[s][i] = 0.0f;
                    for (int e = e0; e < e1; ++e) {
                        const int c = g.vchk[e];
                        const float* src = hc + (size_t)((s ? mx : 0) + c) * D;
                        for (int i = 0; i < D; ++i) { feat[i] = src[i]; feat[D + i] = hv[(size_t)v * D + i]; }
                        const float* at = w.vn_msga[s] + (size_t)(e - (s ? g.E_x : 0)) * Am;
                        for (int i = 0; i < Am; ++i) feat[2 * D + i] = at[i];
                        gg_run(w, 4 + s, feat, msg, bufA, bufB);
                        const float sg = ((s ? sz[c] : sx[c]) & 1) ? -1.0f : 1.0f;
                        for (int i = 0; i < D; ++i) msg[i] = msg[i] * sg;
                        gg_reduce(acc[s], msg, D, e == e0, w.rop);
                    }
                    if (w.rop == FGNN_REDUCE_MEAN && e1 > e0) {
                        const float fd = (float)(e1 - e0);
                        for (int i = 0; i < D; ++i) acc[s][i] = acc[s][i] / fd;
                    }
                }
                for (int i = 0; i < D; ++i) { feat[i] = acc[0][i]; feat[D + i] = acc[1][i]; }
                for (int i = 0; i < An; ++i) feat[2 * D + i] = w.vn_node[(size_t)v * An + i];
                for (int i = 0; i < D; ++i) feat[2 * D + An + i] = hv[(size_t)v * D + i];
                gg_run(w, 6, feat, msg, bufA, bufB);
                float Lv[3] = {0.0f, 0.0f, 0.0f};
                for (int k = 0; k < D; ++k) {
                    hv[(size_t)v * D + k] = msg[k];
                    Lv[0] = FG_FMA(msg[k], w.winv[k * 3 + 0], Lv[0]);
                    Lv[1] = FG_FMA(msg[k], w.winv[k * 3 + 1], Lv[1]);
                    Lv[2] = FG_FMA(msg[k], w.winv[k * 3 + 2], Lv[2]);
                }
                if (w.binv) {
                    Lv[0] = Lv[0] + w.binv[0];
                    Lv[1] = Lv[1] + w.binv[1];
                    Lv[2] = Lv[2] + w.binv[2];
                }
                llr[v] = Lv[0];
                llr[n + v] = Lv[1];
                llr[2 * n + v] = Lv[2];
                lz[v] = fg_softplus(-Lv[0]) - fg_lse2(-Lv[2], -Lv[1]);
                lx[v] = fg_softplus(-Lv[2]) - fg_lse2(-Lv[0], -Lv[1]);
            }
            __syncthreads();
            float* xl = a.xlog_all ? a.xlog_all + ((size_t)it * a.B + b) * rxp : nullptr;
            float* zl = a.zlog_all ? a.zlog_all + ((size_t)it * a.B + b) * rzp : nullptr;
            for (int c = tid; c < m; c += T) {
                const int p0 = g.cptr[c];
                const float vq = logit_row_gnn(c < mx ? lz : lx, g.cvn + p0, g.cptr[c + 1] - p0);
                hlog[c] = vq;
                if (c < mx) { if (zl) zl[c] = vq; }
                else if (xl) xl[c - mx] = vq;
            }
            if (xl)
                for (int r = tid; r < g.rows[5]; r += T)
                    xl[g.m_z + r] = logit_row_gnn(lx, g.rcol[5] + g.rptr[5][r], g.rptr[5][r + 1] - g.rptr[5][r]);
            if (zl)
                for (int r = tid; r < g.rows[4]; r += T)
                    zl[g.m_x + r] = logit_row_gnn(lz, g.rcol[4] + g.rptr[4][r], g.rptr[4][r + 1] - g.rptr[4][r]);
            __syncthreads();
            if (it == a.num_iter - 1) break;  // (:414-415)
        }
        // ---- UpdateCNEmbeddings.call (:573-610) ----
        for (int c = tid; c < m; c += T) {
            const int s = c >= mx;
            const int p0 = g.cptr[c], p1 = g.cptr[c + 1];
            float* hto = hc + (size_t)c * D;
            for (int i = 0; i < D; ++i) acc[0][i] = 0.0f;
            for (int jx = p0; jx < p1; ++jx) {
                const float* src = hv + (size_t)g.cvn[jx] * D;
                for (int i = 0; i < D; ++i) { feat[i] = src[i]; feat[D + i] = hto[i]; }
                const float* at = w.cn_msga[s] + (size_t)(jx - (s ? g.E_x : 0)) * Am;
                for (int i = 0; i < Am; ++i) feat[2 * D + i] = at[i];
                gg_run(w, s, feat, msg, bufA, bufB);
                gg_reduce(acc[0], msg, D, jx == p0, w.rop);
            }
            if (w.rop == FGNN_REDUCE_MEAN && p1 > p0) {
                const float fd = (float)(p1 - p0);
                for (int i = 0; i < D; ++i) acc[0][i] = acc[0][i] / fd;
            }
            const unsigned sb = (s ? sz[c - mx] : sx[c]) & 1;
            const float lg = (it >= 0) ? hlog[c] * (sb ? -1.0f : 1.0f) : 0.0f;  // (:417-418)
            for (int i = 0; i < D; ++i) feat[i] = acc[0][i];
            for (int i = 0; i < An; ++i) feat[D + i] = w.cn_node[s][(size_t)(c - (s ? mx : 0)) * An + i];
            for (int i = 0; i < D; ++i) feat[D + An + i] = hto[i];
            feat[2 * D + An] = lg;
            gg_run(w, 2 + s, feat, msg, bufA, bufB);
            for (int i = 0; i < D; ++i) hto[i] = msg[i];
        }
        __syncthreads();
    }
    for (int v = tid; v < n; v += T) {  // make_hard_decision (:359-367)
        const float X = llr[v], Y = llr[n + v], Z = llr[2 * n + v];
        int d = 0;
        float best = 0.0f;
        if (X < best) { best = X; d = 1; }
        if (Z < best) { best = Z; d = 2; }
        if (Y < best) { best = Y; d = 3; }
        a.x_hat[(size_t)b * n + v] = (uint8_t)(d & 1);
        a.z_hat[(size_t)b * n + v] = (uint8_t)(d >> 1);
    }
}

}  // namespace

struct fgnn_gnnbp4_weights {
    GnnBp4Dev d;
    int device;
    void* blob;
    bool general = false;
    GnnBp4GenDev gen;
};

extern "C" int fgnn_gnnbp4_weights_create(const float* const host_arrays[30], int num_embed_dims, int num_hidden_units,
                                          int device, fgnn_gnnbp4_weights** out)
{
    if (!host_arrays || !out) return fgnn_fail(FGNN_ERR_ARG, "NULL argument");
    if (num_embed_dims != D || num_hidden_units != H)
        return fgnn_fail(FGNN_ERR_ARG, "the GNN_BP4 kernel is built for num_embed_dims=20, num_hidden_units=40");
    for (int i = 0; i < 30; ++i)
        if (!host_arrays[i]) return fgnn_fail(FGNN_ERR_ARG, "weight array is NULL");
    FGNN_DEVICE_GUARD(device);
    std::vector<float> h;
    auto push = [&](size_t count) {  // offsets and sizes in multiples of 8 floats: every array starts 32-byte aligned
        size_t o = h.size();
        h.resize(o + ((count + 7) & ~size_t(7)), 0.0f);
        return o;
    };
    const int nin[7] = {2 * D, 2 * D, 2 * D + 1, 2 * D + 1, 2 * D, 2 * D, 3 * D};
    const int npad[7] = {2 * D, 2 * D, 2 * D + 4, 2 * D + 4, 2 * D, 2 * D, 3 * D};
    size_t off[7][4], off_w1[7];
    for (int q = 0; q < 7; ++q) {
        const float* const* a = host_arrays + 4 * q;
        off_w1[q] = push((size_t)nin[q] * H);  // [nin][H] as given: the streaming kernel's scalar pair loads walk a row in j
        std::memcpy(&h[off_w1[q]], a[0], (size_t)nin[q] * H * sizeof(float));
        off[q][0] = push((size_t)H * npad[q]);
        for (int j = 0; j < H; ++j)
            for (int k = 0; k < nin[q]; ++k) h[off[q][0] + (size_t)j * npad[q] + k] = a[0][(size_t)k * H + j];
        off[q][1] = push(H);
        std::memcpy(&h[off[q][1]], a[1], H * sizeof(float));
        off[q][2] = push((size_t)H * D);
        std::memcpy(&h[off[q][2]], a[2], (size_t)H * D * sizeof(float));
        off[q][3] = push(D);
        std::memcpy(&h[off[q][3]], a[3], D * sizeof(float));
    }
    const size_t owi = push((size_t)D * 4), obi = push(4);
    for (int k = 0; k < D; ++k)
        for (int i = 0; i < 3; ++i) h[owi + k * 4 + i] = host_arrays[28][k * 3 + i];
    std::memcpy(&h[obi], host_arrays[29], 3 * sizeof(float));
    // per-lane MFMA operand tables
    std::vector<float> T;
    int tab_start[7], tab_inv;
    auto entry = [&]() { size_t e = T.size() / 64; T.resize(T.size() + 64, 0.0f); return e; };
    for (int qm = 0; qm < 7; ++qm) {
        const float* const* a = host_arrays + 4 * qm;
        const int S1 = (nin[qm] + 3) / 4;
        tab_start[qm] = (int)(T.size() / 64);
        for (int t = 0; t < 3; ++t)
            for (int st = 0; st < S1; ++st) {
                size_t e = entry();
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 15, kk = lane >> 4, unit = 16 * t + 4 * (rho & 3) + (rho >> 2), k = 4 * st + kk;
                    T[e * 64 + lane] = (unit < H && k < nin[qm]) ? a[0][(size_t)k * H + unit] : 0.0f;
                }
            }
        for (int st = 0; st < 10; ++st) {
            size_t e = entry();
            for (int lane = 0; lane < 64; ++lane) T[e * 64 + lane] = a[1][4 * st + (lane >> 4)];
        }
        for (int u = 0; u < 2; ++u)
            for (int st = 0; st < 10; ++st) {
                size_t e = entry();
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 15, kk = lane >> 4, rp = rho & 3, qp = rho >> 2;
                    const int mu = (u == 0) ? 4 * rp + qp : (rp == 0 ? 16 + qp : -1);
                    T[e * 64 + lane] = mu >= 0 ? a[2][(size_t)(4 * st + kk) * D + mu] : 0.0f;
                }
            }
        for (int i = 0; i < 5; ++i) {
            size_t e = entry();
            for (int lane = 0; lane < 64; ++lane) T[e * 64 + lane] = a[3][i < 4 ? 4 * i + (lane >> 4) : 16 + (lane >> 4)];
        }
    }
    tab_inv = (int)(T.size() / 64);
    for (int st = 0; st < 5; ++st) {
        size_t e = entry();
        for (int lane = 0; lane < 64; ++lane) {
            const int rho = lane & 15, kk = lane >> 4;
            T[e * 64 + lane] = rho < 3 ? host_arrays[28][(size_t)(4 * st + kk) * 3 + rho] : 0.0f;
        }
    }
    for (int r = 0; r < 3; ++r) {
        size_t e = entry();
        for (int lane = 0; lane < 64; ++lane) T[e * 64 + lane] = host_arrays[29][r];
    }
    const size_t otab = push(T.size());
    std::memcpy(&h[otab], T.data(), T.size() * sizeof(float));

    fgnn_gnnbp4_weights* w = new fgnn_gnnbp4_weights();
    w->device = device;
    w->blob = nullptr;
    hipError_t e = hipMalloc(&w->blob, h.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(w->blob, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (w->blob) (void)hipFree(w->blob);
        delete w;
        return fgnn_fail(FGNN_ERR_HIP, std::string("GNN_BP4 weights upload: ") + hipGetErrorString(e));
    }
    const float* base = static_cast<const float*>(w->blob);
    MlpDev* dst[7] = {&w->d.cn_msg[0], &w->d.cn_msg[1], &w->d.cn_embed[0], &w->d.cn_embed[1], &w->d.vn_msg[0], &w->d.vn_msg[1],
                      &w->d.vn_embed};
    for (int q = 0; q < 7; ++q) {
        dst[q]->w1 = base + off_w1[q];
        dst[q]->w1t = base + off[q][0];
        dst[q]->b1 = base + off[q][1];
        dst[q]->w2 = base + off[q][2];
        dst[q]->b2 = base + off[q][3];
    }
    w->d.winv = base + owi;
    w->d.binv = base + obi;
    w->d.lane_tab = base + otab;
    w->d.tab_cn_msg[0] = tab_start[0];
    w->d.tab_cn_msg[1] = tab_start[1];
    w->d.tab_cn_embed[0] = tab_start[2];
    w->d.tab_cn_embed[1] = tab_start[3];
    w->d.tab_vn_msg[0] = tab_start[4];
    w->d.tab_vn_msg[1] = tab_start[5];
    w->d.tab_vn_embed = tab_start[6];
    w->d.tab_inv = tab_inv;
    *out = w;
    return FGNN_OK;
}

extern "C" int fgnn_gnnbp4_weights_create_general(const fgnn_graph* g, const fgnn_gnnbp4_config* cfg, const float* const* host_arrays,
                                                  int num_arrays, fgnn_gnnbp4_weights** out)
{
    if (!g || !cfg || !host_arrays || !out) return fgnn_fail(FGNN_ERR_ARG, "NULL argument");
    const int Dg = cfg->num_embed_dims, Hg = cfg->num_hidden_units, L = cfg->num_mlp_layers, bias = cfg->use_bias ? 1 : 0;
    const int attr = cfg->use_attributes ? 1 : 0, An = attr ? cfg->node_attribute_dims : 0, Am = attr ? cfg->msg_attribute_dims : 0;
    if (Dg < 1 || Dg > GG_MAXD || L < 1 || L > 4 || (L > 1 && (Hg < 1 || Hg > 96)) || An < 0 || An > 16 || Am < 0 || Am > 16)
        return fgnn_fail(FGNN_ERR_ARG, "general GNN_BP4: need 1 <= num_embed_dims <= 32, 1 <= num_hidden_units <= 96, 1 <= num_mlp_layers "
                                       "<= 4, attribute dims <= 16");
    if (cfg->reduce_op < 0 || cfg->reduce_op > 3) return fgnn_fail(FGNN_ERR_ARG, "unknown reduce operation");  // gnn.py:568
    if (cfg->activation < 0 || cfg->activation > 3) return fgnn_fail(FGNN_ERR_ARG, "unsupported activation");
    const int st = 1 + bias;
    if (num_arrays != (7 * L + 1) * st + (attr ? 7 : 0)) return fgnn_fail(FGNN_ERR_ARG, "wrong number of weight arrays for this configuration");
    for (int i = 0; i < num_arrays; ++i)
        if (!host_arrays[i]) return fgnn_fail(FGNN_ERR_ARG, "weight array is NULL");
    FGNN_DEVICE_GUARD(g->device);
    const int n = g->d.n, E[2] = {g->d.E_x, g->d.E_z}, mm[2] = {g->d.m_x, g->d.m_z};
    const int nin[7] = {2 * Dg + Am, 2 * Dg + Am, 2 * Dg + An + 1, 2 * Dg + An + 1, 2 * Dg + Am, 2 * Dg + Am, 3 * Dg + An};
    fgnn_gnnbp4_weights* w = new fgnn_gnnbp4_weights();
    w->device = g->device;
    w->blob = nullptr;
    w->general = true;
    std::memset(&w->d, 0, sizeof(w->d));
    GnnBp4GenDev& G = w->gen;
    std::memset(&G, 0, sizeof(G));
    G.D = Dg; G.H = Hg; G.L = L; G.rop = cfg->reduce_op; G.act = cfg->activation; G.bias = bias; G.An = An; G.Am = Am;
    std::vector<float> h;
    auto add = [&](const float* src, size_t cnt) {
        size_t o = h.size();
        h.insert(h.end(), src, src + cnt);
        h.resize((h.size() + 3) & ~size_t(3), 0.0f);
        return o;
    };
    size_t offW[7][4], offB[7][4];
    int pos = 0;
    for (int q = 0; q < 7; ++q)
        for (int k = 0; k < L; ++k) {
            G.K[q][k] = k == 0 ? nin[q] : Hg;
            G.J[q][k] = k == L - 1 ? Dg : Hg;
            G.actl[q][k] = k == L - 1 ? FGNN_ACT_LINEAR : cfg->activation;
            offW[q][k] = add(host_arrays[pos], (size_t)G.K[q][k] * G.J[q][k]);
            offB[q][k] = bias ? add(host_arrays[pos + 1], (size_t)G.J[q][k]) : 0;
            pos += st;
        }
    const size_t owi = add(host_arrays[pos], (size_t)Dg * 3);
    const size_t obi = bias ? add(host_arrays[pos + 1], 3) : 0;
    pos += st;
    size_t o_cn_node[2] = {0, 0}, o_cn_msg[2] = {0, 0}, o_vn_node = 0, o_vn_msg[2] = {0, 0};
    if (attr) {
        for (int s = 0; s < 2; ++s) o_cn_node[s] = add(host_arrays[pos + s], (size_t)mm[s] * An);
        for (int s = 0; s < 2; ++s) o_cn_msg[s] = add(host_arrays[pos + 2 + s], (size_t)E[s] * Am);
        o_vn_node = add(host_arrays[pos + 4], (size_t)n * An);
        for (int s = 0; s < 2; ++s) {
            // the reference indexes edge attributes by np.where(pcm) (check-major); the qubit update walks VN-major slots: permute here.
            // h_chk / h_var are the canonical (qubit, check)-sorted edge lists; the check-major rank of slot e = the number of edges
            // (c', v') with (c', v') < (c, v) in (check, qubit) order
            std::vector<int> order(E[s]);
            for (int e = 0; e < E[s]; ++e) order[e] = e;
            const std::vector<int32_t>& chk = g->h_chk[s];
            const std::vector<int32_t>& var = g->h_var[s];
            std::sort(order.begin(), order.end(), [&](int x, int y) { return chk[x] != chk[y] ? chk[x] < chk[y] : var[x] < var[y]; });
            std::vector<float> perm((size_t)E[s] * Am + 1, 0.0f);
            for (int r = 0; r < E[s]; ++r)
                for (int i = 0; i < Am; ++i) perm[(size_t)order[r] * Am + i] = host_arrays[pos + 5 + s][(size_t)r * Am + i];
            o_vn_msg[s] = add(perm.data(), (size_t)E[s] * Am);
        }
    }
    if (h.empty()) h.resize(4, 0.0f);
    hipError_t e = hipMalloc(&w->blob, h.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(w->blob, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (w->blob) (void)hipFree(w->blob);
        delete w;
        return fgnn_fail(FGNN_ERR_HIP, std::string("GNN_BP4 weights upload: ") + hipGetErrorString(e));
    }
    const float* base = static_cast<const float*>(w->blob);
    for (int q = 0; q < 7; ++q)
        for (int k = 0; k < L; ++k) {
            G.W[q][k] = base + offW[q][k];
            G.b[q][k] = bias ? base + offB[q][k] : nullptr;
        }
    G.winv = base + owi;
    G.binv = bias ? base + obi : nullptr;
    for (int s = 0; s < 2; ++s) {
        G.cn_node[s] = base + o_cn_node[s];
        G.cn_msga[s] = base + o_cn_msg[s];
        G.vn_msga[s] = base + o_vn_msg[s];
    }
    G.vn_node = base + o_vn_node;
    *out = w;
    return FGNN_OK;
}

extern "C" size_t fgnn_gnnbp4_weights_workspace_bytes(const fgnn_graph* g, const fgnn_gnnbp4_weights* w, int B)
{
    if (!g || !w || B <= 0) return 0;
    return (size_t)B * (size_t)(g->d.n + g->d.m) * (size_t)(w->general ? w->gen.D : D) * sizeof(float);
}

extern "C" void fgnn_gnnbp4_weights_destroy(fgnn_gnnbp4_weights* w)
{
    if (!w) return;
    fgnn_device_guard _dg(w->device);
    if (w->blob) (void)hipFree(w->blob);
    delete w;
}

extern "C" size_t fgnn_gnnbp4_workspace_bytes(const fgnn_graph* g, int B)
{
    if (!g || B <= 0) return 0;
    return (size_t)B * (size_t)(g->d.n + g->d.m) * D * sizeof(float);
}

// FGNN_OPT_GNN_STREAM for GNN_BP4 on a (3,3,6)-regular graph: only the explicit value 2 ("always") runs the streaming packed-FMA kernel.
// Measured in round 4 (profiles/r4_gnnbp4_stream_ab.txt, r4_gnnbp4_stream_pmc_summary.txt): bit-equal to the MFMA-tile kernel in both
// associations, and slower — 180 ms against 134 ms per 16 384 x 10 in the factored order, 212 against 193 in the literal one.  The
// packed fma holds a SIMD for ~4.4 cycles per two multiply-adds, so the uniform-weight multiply-adds run at the f32 MFMA's rate minus
// nothing: what the streaming form saves on the tiles' padded rows (24 % of the MFMA cycles) it pays back in the packed instruction's
// 10 % issue overhead, a 70 KB weight stream through a 16 KB scalar cache (12 % misses; the MFMA kernel keeps its operands in LDS) and
// three waves per SIMD of 168 VGPRs.  Both forms price at ~107-110 ms if issued perfectly; the MFMA kernel is at 83 % of that, the
// streaming kernel at 60 %.  It stays in the library as the tested second implementation of the same float operations.
static bool gnnbp4_takes_stream(const fgnn_graph* g, int B)
{
    (void)B;
    return g->gnn_stream == 2;
}

extern "C" int fgnn_gnnbp4_decode(const fgnn_graph* g, const fgnn_gnnbp4_weights* w, int num_iter, const uint8_t* synd_x,
                                  const uint8_t* synd_z, int B, uint8_t* x_hat, uint8_t* z_hat, float* llr_out,
                                  float* x_logit_all, float* z_logit_all, void* workspace, size_t ws_bytes, void* stream)
{
    if (!g || !w) return fgnn_fail(FGNN_ERR_ARG, "graph or weights is NULL");
    if (num_iter < 1 || B < 0) return fgnn_fail(FGNN_ERR_ARG, "num_iter must be >= 1 and B >= 0");
    if (!g->d.rptr[4] || !g->d.rptr[5]) return fgnn_fail(FGNN_ERR_STATE, "lx / lz row sets not installed (fgnn_graph_set_rows 4, 5)");
    if (w->device != g->device) return fgnn_fail(FGNN_ERR_ARG, "weights and graph live on different devices");
    if (B == 0) return FGNN_OK;  // an empty batch needs no buffers
    if (!synd_x || !synd_z || !x_hat || !z_hat || !llr_out) return fgnn_fail(FGNN_ERR_ARG, "required buffer is NULL");
    if (!workspace || ws_bytes < fgnn_gnnbp4_weights_workspace_bytes(g, w, B)) return fgnn_fail(FGNN_ERR_ARG, "workspace too small");
    FGNN_DEVICE_GUARD(g->device);
    Args a;
    a.B = B;
    a.num_iter = num_iter;
    a.synd_x = synd_x;
    a.synd_z = synd_z;
    a.x_hat = x_hat;
    a.z_hat = z_hat;
    a.llr_out = llr_out;
    a.xlog_all = x_logit_all;
    a.zlog_all = z_logit_all;
    a.work = static_cast<float*>(workspace);
    const size_t lds_bytes = (size_t)(2 * g->d.n + g->d.m) * sizeof(float);
    fgnn_prof_scope prof(g, static_cast<hipStream_t>(stream));
    if (w->general) {
        hipLaunchKernelGGL(gnn_bp4_general_kernel, dim3(B), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), g->d, w->gen, a);
        FGNN_HIP_CHECK(hipGetLastError());
        prof.done(FGNN_PROF_TAG_GNNBP4, B);
        return FGNN_OK;
    }
    if (g->d.dvx == 3 && g->d.dvz == 3 && g->d.dc == 6 && !g->force_generic && gnnbp4_takes_stream(g, B)) {
        const size_t lds_s = (size_t)(2 * g->d.n + 2 * g->d.m) * sizeof(float);
        if (lds_s > FGNN_LDS_BUDGET) return fgnn_fail(FGNN_ERR_ARG, "code too large for the GNN_BP4 streaming kernel");
        auto kern = g->gnn_factored ? gnn_bp4_stream_kernel<3, 6, true> : gnn_bp4_stream_kernel<3, 6, false>;
        GnnBp4Dev wd = w->d;
#ifdef FGNN_PROBES  // timing probe, never in the shipped library (make EXTRA=-DFGNN_PROBES; results wrong by construction: every MLP reads ONE 10 KB weight set)
        static const bool alias_probe = getenv("FGNN_PROBE_ALIAS_WEIGHTS") != nullptr;
        if (alias_probe) {
            for (MlpDev* q : {&wd.cn_msg[0], &wd.cn_msg[1], &wd.cn_embed[0], &wd.cn_embed[1], &wd.vn_msg[1], &wd.vn_embed}) *q = wd.vn_msg[0];
        }
#endif
        if (lds_s > 48 * 1024)
            FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
        hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_s, static_cast<hipStream_t>(stream), g->d, wd, a);
        FGNN_HIP_CHECK(hipGetLastError());
        prof.done(FGNN_PROF_TAG_GNNBP4, B);
        return FGNN_OK;
    }
    if (g->d.dvx == 3 && g->d.dvz == 3 && g->d.dc == 6 && !g->force_generic) {
        const int cn_entries = w->d.tab_vn_msg[0] - w->d.tab_cn_msg[0], vn_entries = w->d.tab_inv + 8 - w->d.tab_vn_msg[0];
        const size_t fixed = lds_bytes + (size_t)g->d.m * sizeof(float);  // + the syndrome signs
        const int resident = fixed + (size_t)(cn_entries + vn_entries) * 256 <= FGNN_LDS_BUDGET && !getenv("FGNN_GNNBP4_NO_RESIDENT");
        const int tab_floats = (resident ? cn_entries + vn_entries : (cn_entries > vn_entries ? cn_entries : vn_entries)) * 64;
        const size_t lds2 = fixed + (size_t)tab_floats * sizeof(float);
        auto kern = g->gnn_factored ? gnn_bp4_mfma_kernel<3, 6, true> : gnn_bp4_mfma_kernel<3, 6, false>;
        if (lds2 > FGNN_LDS_BUDGET) return fgnn_fail(FGNN_ERR_ARG, "code too large for the GNN_BP4 MFMA kernel");
        FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(kern, dim3(B), dim3(FGNN_GNNBP4_THREADS), lds2, static_cast<hipStream_t>(stream), g->d, w->d, a, tab_floats, resident);
        FGNN_HIP_CHECK(hipGetLastError());
        prof.done(FGNN_PROF_TAG_GNNBP4, B);
        return FGNN_OK;
    }
    hipLaunchKernelGGL(g->gnn_factored ? gnn_bp4_kernel<true> : gnn_bp4_kernel<false>, dim3(B), dim3(256), lds_bytes,
                       static_cast<hipStream_t>(stream), g->d, w->d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    prof.done(FGNN_PROF_TAG_GNNBP4, B);
    return FGNN_OK;
}
