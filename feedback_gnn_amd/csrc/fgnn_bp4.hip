// fgnn_bp4.hip — quaternary belief propagation (BP4) over the two Tanner graphs of a CSS code,
// LDS-resident: one workgroup keeps ALL messages of its codeword(s) in LDS for ALL iterations.
//
// Replaces QLDPCBPDecoder.call of /root/reference sionna/fec/ldpc/decoding_q.py:661-797, i.e. per
// iteration _vn_update (:227-275), the CN-order gather (:752-753), _cn_update_{phi,minsum,tanh}
// (:376-431 / :539-644 / :313-363), the normalisation (:759-760) and the gather back (:766-767);
// then the marginals (:777), cal_logit (:455-471) and the hard decision (:783-790).
//
// The reference streams every [E,bs] message tensor through HBM ~10 times per half-iteration.
// Here a codeword's state (E floats of messages, 21 KB for [[882,24]], in LDS; per-qubit channel LLRs in registers for the
// benchmark codes, else 3n more floats of LDS) never leaves the CU: HBM sees the syndromes once and the results once.  The kernel is therefore bound
// by VALU issue (about 830 fma-class ops per qubit-iteration for the exact exp/log of
// fgnn_math.h), not by HBM; DESIGN.md §4 gives the accounting next to the streaming-model figure.
//
// Message layout in LDS: slot e in [0,E_x) = hx edges, [E_x,E) = hz edges, both sorted by
// (qubit, check).  The VN phase reads/writes a contiguous run per qubit; the CN phase gathers its
// slots through g.cslot.  Updates are in place; two workgroup barriers per iteration.
//
// Summation order is the canonical one of the oracle (ascending neighbour index) and every
// float op is the one the oracle executes, so results are bit-identical to oracle/fgnn_oracle.c.
#include <cstdlib>

#include "fgnn_internal.h"
#include "fgnn_math.h"

#ifndef FGNN_BP4_WAVES
#define FGNN_BP4_WAVES 7  // waves per SIMD the register allocator must leave room for (LDS admits 5-7 workgroups of 4 waves per CU)
#endif
#ifndef FGNN_HWT_PIN_CLIP_POINTS
#define FGNN_HWT_PIN_CLIP_POINTS 1  // opt-in hardware-transcendental phi: pin phi(clip max) = 0, phi(clip min) = 16.635532
#endif
#ifndef FGNN_PHI_STAGE
#define FGNN_PHI_STAGE 1  // phi evaluations staged together in the regular check-node update (phi_n), must divide DC
#endif

namespace {

struct BpArgs {
    int B, num_iter, tpc, cpb, lds_per_cw, lch_off;  // per-codeword LDS floats; offset of the channel LLRs
    float factor, llr_const;
    const float* llr_ch;      // [B,3,n] or null
    const uint8_t* synd_x;    // [B,m_x]
    const uint8_t* synd_z;    // [B,m_z]
    const float* msg_init_x;  // [B,E_x] or null
    const float* msg_init_z;
    float* llr_out;           // [B,3,n]
    uint8_t* x_hat;
    uint8_t* z_hat;
    float* x_logit;           // [B,rows0] or null
    float* z_logit;           // [B,rows1] or null
    float* msg_out_x;
    float* msg_out_z;
    const int* index;         // optional: workgroup slot -> sample (compacted rounds of the sandwich driver)
    int shortcut;             // 1: wave-uniform exact shortcuts for saturated nodes (regular kernel)
    int early_exit;           // 1: leave the iteration loop at a proven fixed point (needs shortcut, cpb == 1, phi rule)
    int sig_off;              // float offset of the fixed-point detector's LDS words (n sign words + 4 flags)
    int lreg;                 // 4 / 5: channel LLRs in registers (kernel variant NQ = lreg), 0: in LDS
    int hwt;                  // 1: v_exp_f32 / v_log_f32 instead of fgnn_math.h (FGNN_OPT_HW_TRANSCENDENTALS; phi rule, fixed dataflow)
    uint8_t* flagged;         // optional [B]: 1 iff the decision's syndrome differs from the measured one (feedback_gnn.py:324-328)
    int flag_off;             // float offset of n decision bytes + one word in LDS (only when flagged != null)
    float* gmem;              // GMEM variant: workspace of (blocks * cpb) rows of lds_per_cw floats
    float* trace_x;           // TRACE variant: [num_iter+1, B, rows0] soft syndromes after 0, 1, ..., num_iter iterations (:743-746)
    float* trace_z;           //                [num_iter+1, B, rows1]
    int trace_off;            // float offset of the 2n binary LLRs the per-iteration soft syndromes are formed from
    float* tape_x;            // TRACE variant, optional: [num_iter+1, B, E_x] c->v messages before iteration k (k = num_iter: after the last),
    float* tape_z;            //                          [num_iter+1, B, E_z] — the tape fgnn_bp4_backward reads
    int shared_lse;           // 1: the (a - b)-dependent part of the qubit update's log-sum-exp once per qubit and side (FGNN_OPT_BP4_SHARED_LSE)
};

__device__ __forceinline__ unsigned sign_bit(float x) { return fg_f2u(x) >> 31; }
__device__ __forceinline__ float with_sign(float mag, unsigned neg) { return fg_u2f(fg_f2u(mag) ^ (neg << 31)); }

// Elementary-function policy of the kernel.  Mx<false> = fgnn_math.h: the float operations the CPU oracle executes, bit for bit
// (the default, the headline, every parity test).  Mx<true> = the same TensorFlow op structure on the hardware's own
// v_exp_f32 / v_log_f32 (1-ulp table units whose bits no CPU reproduces): FGNN_OPT_HW_TRANSCENDENTALS, explicitly opt-in, fixed
// dataflow only, reported by bench.py under `extras` next to its measured agreement with the exact kernel — the price of
// bit-exactness as a number.  What a TensorFlow-ROCm build of the reference would execute is of this kind.
template <bool HWT>
struct Mx {
    static __device__ __forceinline__ float softplus(float t) { return fg_softplus(t); }
    static __device__ __forceinline__ float lse2(float a, float b) { return fg_lse2(a, b); }
    static __device__ __forceinline__ float lse2_corr(float a, float b) { return fg_lse2_corr(a, b); }
    static __device__ __forceinline__ float phi(float x) { return fg_phi(x); }
};
template <>
struct Mx<true> {
    static __device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(x * FG_LOG2E); }
    static __device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * 0.693147182f; }
    static __device__ __forceinline__ float softplus(float t)  // tf2xla Softplus, as fg_softplus
    {
        const float y = exp(FG_CLAMP(t, -87.0f, FG_SOFTPLUS_THRESH));
        const float r = (t < -FG_SOFTPLUS_THRESH) ? ((t < -87.0f) ? 0.0f : y) : log(1.0f + y);
        return (t > FG_SOFTPLUS_THRESH) ? t : r;
    }
    static __device__ __forceinline__ float lse2_corr(float a, float b)  // as fg_lse2_corr
    {
        const float y = exp(-FG_MIN(FG_ABS(a - b), 20.0f));
        return log(1.0f + y);
    }
    static __device__ __forceinline__ float lse2(float a, float b)  // max-shifted reduce_logsumexp of a pair, as fg_lse2
    {
        return lse2_corr(a, b) + FG_MAX(a, b);
    }
    // decoding_q.py:365-373, as fg_phi.  The two clip points are pinned to the values the reference's own known answer fixes
    // (examples/n1270.ipynb cell 12: saturated marginals log 57 +- deg * 16.635532, i.e. phi(clip max) = 0 and phi(clip min) =
    // 16.635532 exactly, which TensorFlow's kernels and fgnn_math.h deliver): the units' composition log(exp(x) - 1) misses both
    // (exp(8.5e-8) may round to 1: log 0), and a decoder whose saturated messages are off sits at a different fixed point.
    static __device__ __forceinline__ float phi(float x)
    {
        const float xc = FG_CLAMP(x, FG_PHI_MIN, FG_PHI_MAX);
        const float y = exp(xc);
        const float sp = (xc > FG_SOFTPLUS_THRESH) ? xc : log(1.0f + y);
        float r = sp - log(y - 1.0f);
#if FGNN_HWT_PIN_CLIP_POINTS
        r = (x >= FG_PHI_MAX) ? 0.0f : r;
        r = (x <= FG_PHI_MIN) ? 16.6355324f : r;
#endif
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// Check-node rules on runtime-degree rows.  `msg` = this codeword's LDS message array, `slot` =
// the check's slot list.  Pass 1 parks |.|-type intermediates in the slots themselves (sign kept in
// the sign bit), pass 2 writes the c->v messages.  decoding_q.py line numbers as in the oracle.
// ---------------------------------------------------------------------------------------------
template <int CN_TYPE, bool HWT = false>
__device__ __forceinline__ void cn_update(float* msg, const int* __restrict__ slot, int deg, unsigned synd, float factor)
{
    if constexpr (CN_TYPE == FGNN_CN_BOXPLUS_PHI) {  // _cn_update_phi (:376-431)
        unsigned neg = synd;
        float T = 0.0f;
        for (int j = 0; j < deg; ++j) {
            int s = slot[j];
            float v = msg[s];
            unsigned ng = v < 0.0f;
            neg ^= ng;
            float a = Mx<HWT>::phi(FG_ABS(v));
            T = T + a;
            msg[s] = with_sign(a, ng);
        }
        for (int j = 0; j < deg; ++j) {
            int s = slot[j];
            float w = msg[s];
            float out = Mx<HWT>::phi(T - FG_ABS(w));
            msg[s] = with_sign(out, neg ^ sign_bit(w)) * factor;
        }
    } else if constexpr (CN_TYPE == FGNN_CN_MINSUM) {  // _cn_update_minsum (:539-644)
        const float LARGE = 10000.0f;
        unsigned neg = synd;
        float minv = 0.0f;
        for (int j = 0; j < deg; ++j) {
            int s = slot[j];
            float v = FG_MIN(FG_MAX(msg[s], -20.0f), 20.0f);
            unsigned ng = v < 0.0f;
            neg ^= ng;
            float a = FG_ABS(v);
            minv = (j == 0) ? a : FG_MIN(minv, a);
            msg[s] = with_sign(a, ng);
        }
        float min2 = 0.0f, nsum = 0.0f;
        for (int j = 0; j < deg; ++j) {
            float d = FG_ABS(msg[slot[j]]) - minv;
            d = (d == 0.0f) ? LARGE : d;
            min2 = (j == 0) ? d : FG_MIN(min2, d);
            nsum = nsum + d;
        }
        min2 = min2 + minv;
        nsum = nsum - (2.0f * LARGE - 1.0f);
        float sg = (nsum > 0.0f) ? 1.0f : ((nsum < 0.0f) ? -1.0f : 0.0f);
        float dm = 0.5f * (1.0f - sg);
        float min_e = (1.0f - dm) * minv + dm * min2;
        for (int j = 0; j < deg; ++j) {
            int s = slot[j];
            float w = msg[s];
            float d = FG_ABS(w) - minv;
            float out = (d == 0.0f) ? min_e : minv;
            msg[s] = with_sign(out, neg ^ sign_bit(w)) * factor;
        }
    } else {  // _cn_update_tanh (:313-363)
        float P = 1.0f;
        for (int j = 0; j < deg; ++j) {
            int s = slot[j];
            float t = fg_tanh(msg[s] / 2.0f);
            t = (t == 0.0f) ? 1e-12f : t;
            P = (j == 0) ? t : P * t;
            msg[s] = t;
        }
        P = P * (synd ? -1.0f : 1.0f);
        const float clipv = 0.99999988f;
        for (int j = 0; j < deg; ++j) {
            int s = slot[j];
            float q = fg_rcp_unit(msg[s]) * P;
            q = (FG_ABS(q) < 1e-7f) ? 0.0f : q;
            q = FG_MIN(FG_MAX(q, -clipv), clipv);
            msg[s] = (2.0f * fg_atanh(q)) * factor;
        }
    }
}

// soft syndrome of one row, _cn_update_phi_loss (:433-453)
template <bool HWT = false>
__device__ __forceinline__ float logit_row(const float* llr, const int* __restrict__ col, int deg)
{
    unsigned neg = 0;
    float T = 0.0f;
    for (int j = 0; j < deg; ++j) {
        float v = llr[col[j]];
        neg ^= (v < 0.0f);
        T = T + Mx<HWT>::phi(FG_ABS(v));
    }
    return with_sign(Mx<HWT>::phi(T), neg);
}

// The same with the exact saturation shortcut: a wave whose rows all see |llr| >= 16.635532 everywhere has phi(|.|) = 0 for every
// term, T = 0 and phi(T) = phi(clip min) = phi0 — the soft syndrome of a converged codeword.
template <bool HWT = false>
__device__ __forceinline__ float logit_row_opt(const float* llr, const int* __restrict__ col, int deg, float phi0, bool shortcut)
{
    if (shortcut) {
        unsigned neg = 0;
        bool sat = true;
        for (int j = 0; j < deg; ++j) {
            const float v = llr[col[j]];
            neg ^= (v < 0.0f);
            sat = sat && (FG_ABS(v) >= FG_PHI_MAX);
        }
        if (__all(sat)) return with_sign(phi0, neg);
    }
    return logit_row<HWT>(llr, col, deg);
}

// N evaluations of fg_phi (fgnn_math.h) with the work of the N lanes-worth of values laid out in three stages, so that the 2N table
// reads of the logs are in flight together instead of one read - wait - use per log: (A) clamp, exp, the two log arguments and
// their table addresses; (B) the 2N two-dword LDS reads; (C) remainders, polynomials, assembly.  Every value goes through exactly
// the float operations of fg_phi in the same order — only the instruction schedule differs, the bits do not.
template <int N, bool HWT = false>
__device__ __forceinline__ void phi_n(const float (&x)[N], float (&out)[N])
{
    if constexpr (HWT) {
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = Mx<true>::phi(x[k]);
        return;
    }
    const float* tab = fg_log_tab();
    float xc[N], y[N];
    uint32_t eb1[N], eb2[N], j1[N], j2[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        xc[k] = FG_CLAMP(x[k], FG_PHI_MIN, FG_PHI_MAX);
        y[k] = fg_exp(xc[k]);
        const uint32_t w1 = fg_f2u(1.0f + y[k]) - FG_LOG_OFFS;  // fg_log1p(y)
        const uint32_t w2 = fg_f2u(y[k] - 1.0f) - FG_LOG_OFFS;  // fg_log(y - 1)
        eb1[k] = w1 & 0xff800000u;
        eb2[k] = w2 & 0xff800000u;
        j1[k] = (w1 >> 18) & 31u;
        j2[k] = (w2 >> 18) & 31u;
    }
    __builtin_amdgcn_sched_barrier(0);
    float rc1[N], lc1[N], rc2[N], lc2[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        rc1[k] = tab[j1[k]];
        lc1[k] = tab[32 + j1[k]];
        rc2[k] = tab[j2[k]];
        lc2[k] = tab[32 + j2[k]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float a = fg_u2f(fg_f2u(rc1[k]) - eb1[k]);
        const float r1 = FG_FMA(y[k], a, a - 1.0f);
        float sp = FG_FMA((float)(int32_t)eb1[k], FG_LN2_S23, lc1[k] + fg_log1p_small(r1));
        sp = (xc[k] > FG_SOFTPLUS_THRESH) ? xc[k] : sp;
        const float mm = fg_u2f(fg_f2u(y[k] - 1.0f) - eb2[k]);
        const float r2 = FG_FMA(mm, rc2[k], -1.0f);
        const float lg = FG_FMA((float)(int32_t)eb2[k], FG_LN2_S23, lc2[k] + fg_log1p_small(r2));
        out[k] = sp - lg;
    }
}

// c->v update of one (DC-regular) check with every message in registers: the phi rule of the benchmark
// configurations without the LDS round trip of the runtime-degree version.  Same float ops, same order.
// Signs are carried as integer sign words: neg = (synd << 31) ^ bits(v_0) ^ ... (bit 31 = the parity of :398-399), and the
// outgoing sign of edge j is bit 31 of neg ^ bits(v_j) — one xor per edge in, one xor + one bit-field insert per edge out.
// `sl` = BYTE offsets of the check's slots from `msg` (the packed rows of g.cslot16 hold 4 * slot); F1 = the normalisation factor is
// exactly 1 (feedback_gnn.py / n882.py: every paper run), so the product with it — the identity on every float — is not issued.
__device__ __forceinline__ float& slot_ref(float* msg, int byte_off)
{
    return *reinterpret_cast<float*>(reinterpret_cast<char*>(msg) + byte_off);
}
template <int DC, bool HWT = false, bool F1 = false>
__device__ __forceinline__ bool cn_phi_regular(float* msg, const int (&sl)[DC], unsigned synd, float factor, float phi0,
                                               bool shortcut)
{
    constexpr int H = (DC % FGNN_PHI_STAGE == 0) ? FGNN_PHI_STAGE : DC / 2;
    static_assert(DC % H == 0, "a phi stage must take a whole fraction of a check");
    float v[DC], aa[DC];
    uint32_t neg = synd << 31;
    bool sat = true;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        v[j] = slot_ref(msg, sl[j]);
        neg ^= fg_f2u(v[j]);
        sat = sat && (FG_ABS(v[j]) >= FG_PHI_MAX);
    }
    // Saturation shortcut (exact): if every incoming |nu| >= 16.635532 then every phi(|nu|) is phi(clip max) = 0
    // bit for bit, T = 0, and every outgoing magnitude is phi(0 - 0) = phi(clip min) =: phi0.  Taken only when ALL
    // lanes of the wave are saturated (one v_cmp + s_cbranch), which is the steady state of a converged codeword.
    if (shortcut && __all(sat)) {
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            const float o = fg_u2f(fg_f2u(phi0) | ((neg ^ fg_f2u(v[j])) & 0x80000000u));
            slot_ref(msg, sl[j]) = F1 ? o : o * factor;
        }
        return true;
    }
#pragma unroll
    for (int g0 = 0; g0 < DC; g0 += H) {
        float xa[H], oa[H];
#pragma unroll
        for (int j = 0; j < H; ++j) xa[j] = FG_ABS(v[g0 + j]);
        phi_n<H, HWT>(xa, oa);
#pragma unroll
        for (int j = 0; j < H; ++j) aa[g0 + j] = oa[j];
    }
    float T = 0.0f;
#pragma unroll
    for (int j = 0; j < DC; ++j) T = T + aa[j];
#pragma unroll
    for (int g0 = 0; g0 < DC; g0 += H) {
        float xa[H], oa[H];
#pragma unroll
        for (int j = 0; j < H; ++j) xa[j] = T - aa[g0 + j];
        phi_n<H, HWT>(xa, oa);
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const float o = fg_u2f(fg_f2u(oa[j]) | ((neg ^ fg_f2u(v[g0 + j])) & 0x80000000u));
            slot_ref(msg, sl[g0 + j]) = F1 ? o : o * factor;
        }
    }
    return false;
}

// The min-sum and tanh rules on a check of compile-time degree DC: the DC messages are read once, every intermediate that cn_update
// parks in the slots (|v| with the sign bit, tanh(v/2)) stays in registers, each slot is written once.  Same float operations in the
// same order as cn_update<FGNN_CN_MINSUM> / <FGNN_CN_BOXPLUS>.
template <int DC>
__device__ __forceinline__ void cn_minsum_regular(float* msg, const int (&sl)[DC], unsigned synd, float factor)
{
    const float LARGE = 10000.0f;
    float a[DC];
    unsigned ng[DC];
    unsigned neg = synd;
    float minv = 0.0f;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        const float v = FG_MIN(FG_MAX(slot_ref(msg, sl[j]), -20.0f), 20.0f);
        ng[j] = v < 0.0f;
        neg ^= ng[j];
        a[j] = FG_ABS(v);
        minv = (j == 0) ? a[j] : FG_MIN(minv, a[j]);
    }
    float min2 = 0.0f, nsum = 0.0f;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        float d = a[j] - minv;
        d = (d == 0.0f) ? LARGE : d;
        min2 = (j == 0) ? d : FG_MIN(min2, d);
        nsum = nsum + d;
    }
    min2 = min2 + minv;
    nsum = nsum - (2.0f * LARGE - 1.0f);
    const float sg = (nsum > 0.0f) ? 1.0f : ((nsum < 0.0f) ? -1.0f : 0.0f);
    const float dm = 0.5f * (1.0f - sg);
    const float min_e = (1.0f - dm) * minv + dm * min2;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        const float out = ((a[j] - minv) == 0.0f) ? min_e : minv;
        slot_ref(msg, sl[j]) = with_sign(out, neg ^ ng[j]) * factor;
    }
}

template <int DC>
__device__ __forceinline__ void cn_tanh_regular(float* msg, const int (&sl)[DC], unsigned synd, float factor)
{
    float t[DC];
    float P = 1.0f;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        float tj = fg_tanh(slot_ref(msg, sl[j]) / 2.0f);
        tj = (tj == 0.0f) ? 1e-12f : tj;
        P = (j == 0) ? tj : P * tj;
        t[j] = tj;
    }
    P = P * (synd ? -1.0f : 1.0f);
    const float clipv = 0.99999988f;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        float q = fg_rcp_unit(t[j]) * P;
        q = (FG_ABS(q) < 1e-7f) ? 0.0f : q;
        q = FG_MIN(FG_MAX(q, -clipv), clipv);
        slot_ref(msg, sl[j]) = (2.0f * fg_atanh(q)) * factor;
    }
}

// Runtime-degree phi rule with the exact saturation shortcut of cn_phi_regular: returns true when the whole wave took it.
template <bool HWT = false>
__device__ __forceinline__ bool cn_phi_generic(float* msg, const int* __restrict__ slot, int deg, unsigned synd, float factor,
                                               float phi0, bool shortcut)
{
    if (shortcut) {
        unsigned neg = synd;
        bool sat = true;
        for (int j = 0; j < deg; ++j) {
            const float v = msg[slot[j]];
            neg ^= (v < 0.0f);
            sat = sat && (FG_ABS(v) >= FG_PHI_MAX);
        }
        if (__all(sat)) {
            for (int j = 0; j < deg; ++j) {
                const int s = slot[j];
                msg[s] = with_sign(phi0, neg ^ (unsigned)(msg[s] < 0.0f)) * factor;
            }
            return true;
        }
    }
    cn_update<FGNN_CN_BOXPLUS_PHI, HWT>(msg, slot, deg, synd, factor);
    return false;
}

// tf2xla softplus for |t| > 13.94 only: identity above, exp(t) below (flushed under -87); same bits as fg_softplus there.
__device__ __forceinline__ float softplus_saturated(float t)
{
    const float y = fg_exp(FG_MIN(FG_MAX(t, -87.0f), 0.0f));
    return (t > 0.0f) ? t : ((t < -87.0f) ? 0.0f : y);
}

// DVX/DVZ/DC > 0: every qubit has exactly DVX hx-edges and DVZ hz-edges and every check DC edges, so a
// qubit's slots are v*DVX+k / E_x+v*DVZ+k (no index loads) and a check's DC slots come as one packed
// 16-byte row of g.cslot16.  DVX = 0: runtime degrees through the CSR tables.
// OPT = false compiles the exact optimisations (saturation shortcut, fixed-point detector) out: the fixed-dataflow
// variant bench.py's headline times carries none of their tests.
// NQ > 0: per-qubit channel LLRs (llr_ch != null: every decoder of a sandwich but the first) live in REGISTERS, 3 x NQ per thread
// for the at most NQ qubits lane, lane + tpc, ... a thread owns, instead of 3n floats of LDS: the workgroup then needs the message
// area only and 7 instead of 4 ([[882,24]]) / 5 instead of 3 ([[1270,28]]) workgroups share a CU; the register budget is that of
// 6 waves per SIMD.  The launch picks NQ = ceil(n / tpc) when that is 4 or 5.
// TRACE: the `trainable` / `stage_two` return mode of the reference (decoding_q.py:743-746, 779-781): the soft syndromes of the
// marginals are recorded after EVERY iteration (and before the first) by the kernel itself, from 2n binary LLRs in their own LDS
// area — one launch instead of num_iter + 1 chained one-iteration launches with the messages going through HBM in between.  Same
// float operations in the same order as the epilogue below, so the trace equals that chain's bit for bit.
// GMEM: the per-codeword state (messages, channel LLRs, epilogue scratch) lives in a global-memory workspace row instead of LDS — the
// fallback for codes whose E + 3n floats exceed the CU's LDS (runtime degrees, fixed dataflow only).  Same float operations in the same
// order; within a workgroup __syncthreads() orders the global stores of one phase before the loads of the next.
// LSE: the form of the qubit update's log-sum-exp term — 0 = per edge (the reference's formulas term by term, decoding_q.py:254-273: the
// library default), 1 = once per qubit and side (FGNN_OPT_BP4_SHARED_LSE), both compiled in for the (3,3,6)-regular phi kernels; 2 = either,
// chosen by a.shared_lse at run time (every other instantiation).  The compile-time forms run at the speed of the runtime flag (44.8 / 39.5
// ms either way); what they buy is one kernel SYMBOL per form, so that a rocprofv3 trace or PMC pass prices each form by itself.
template <int CN_TYPE, int DVX, int DVZ, int DC, bool OPT, bool HWT = false, int NQ = 0, bool TRACE = false, bool GMEM = false, int LSE = 2>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(NQ > 0 ? FGNN_BP4_WAVES - 1 : FGNN_BP4_WAVES)))
bp4_kernel(GraphDev g, BpArgs a)
{
    static_assert(!(HWT && OPT), "the hardware-transcendental variant is the fixed dataflow only: the exact shortcuts are proofs about fgnn_math.h");
    static_assert(!GMEM || (DVX == 0 && !OPT && !HWT && NQ == 0 && !TRACE), "the global-memory variant is the plain runtime-degree kernel");
    using MX = Mx<HWT>;
    FG_LOG_TAB_SETUP();
    constexpr bool REGULAR = DVX > 0;
    constexpr bool LREG = NQ > 0;
    constexpr int NQA = LREG ? NQ : 1;
    const bool opt_shortcut = OPT && a.shortcut != 0;
    const bool opt_exit = OPT && a.early_exit != 0;
    const bool shl = LSE == 2 ? a.shared_lse != 0 : LSE == 1;  // workgroup-uniform; a compile-time constant for LSE = 0 / 1
    extern __shared__ float lds[];
    const int cwl = threadIdx.x / a.tpc;
    const int lane = threadIdx.x - cwl * a.tpc;
    const int slot_b = blockIdx.x * a.cpb + cwl;
    const bool active = slot_b < a.B;  // padding codewords of the last block only keep the barriers company
    const int b = (active && a.index) ? a.index[slot_b] : slot_b;
    float* msg;
    if constexpr (GMEM) msg = a.gmem + (size_t)slot_b * a.lds_per_cw;
    else msg = lds + (size_t)cwl * a.lds_per_cw;
    float* Lch = msg + a.lch_off;  // [3n], only when llr_ch != null
    const int n = g.n;

    // Closed-form first iteration (exact; product default only): with zero initial messages and one constant channel LLR every
    // qubit sends the same v->c value in iteration 0, so a check's six outputs differ by the syndrome sign alone.
    const bool first_closed = OPT && REGULAR && CN_TYPE == FGNN_CN_BOXPLUS_PHI && opt_shortcut && !a.llr_ch && !a.msg_init_x &&
                              !a.msg_init_z && a.num_iter > 0;
    if (active && !first_closed) {
        for (int e = lane; e < g.E_x; e += a.tpc) msg[e] = a.msg_init_x ? a.msg_init_x[(size_t)b * g.E_x + e] : 0.0f;
        for (int e = lane; e < g.E_z; e += a.tpc)
            msg[g.E_x + e] = a.msg_init_z ? a.msg_init_z[(size_t)b * g.E_z + e] : 0.0f;
        if (a.llr_ch && !LREG)
            for (int i = lane; i < 3 * n; i += a.tpc) Lch[i] = a.llr_ch[(size_t)b * 3 * n + i];
    }
    float lreg[3][NQA];  // LREG: channel LLRs of qubits lane + i * tpc
    if constexpr (LREG) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int v = lane + i * a.tpc;
            const bool in = active && v < n;
#pragma unroll
            for (int c = 0; c < 3; ++c) lreg[c][i] = in ? a.llr_ch[(size_t)b * 3 * n + c * n + v] : 0.0f;
        }
    }
    __syncthreads();

    const uint8_t* sx = a.synd_x + (size_t)b * g.m_x;
    const uint8_t* sz = a.synd_z + (size_t)b * g.m_z;
    const int lane_c = lane + (a.tpc >> 1) < a.tpc ? lane + (a.tpc >> 1) : lane - a.tpc + (a.tpc >> 1);

    const float phi0 = MX::phi(0.0f);  // = phi(clip min) = 16.6355324, the saturated message magnitude
    (void)phi0;
    // Fixed-point detector (exact early exit, regular phi kernel, one codeword per block).  If the check-node phases of
    // iterations t-1 and t-2 were all-shortcut, every c->v message entering iterations t-1 and t has the same magnitude
    // phi0*factor, so "the sign words of iteration t equal those of iteration t-1" means mu^t == mu^(t-1) bit for bit; the
    // iteration map is deterministic, hence mu^(t+1) == mu^t and every later iteration is the identity.
    unsigned* sigw = reinterpret_cast<unsigned*>(msg + a.sig_off);  // [n] sign words, then 2 x {changed, cn_slow}
    int* flags = reinterpret_cast<int*>(sigw + n);
    bool a1 = false, a2 = false;
    if (opt_exit) {
        // (closed-form start: the sign words iteration 0 would have recorded are those of the all-zero messages)
        for (int v = lane; v < n; v += a.tpc) sigw[v] = first_closed ? 0u : 0xffffffffu;
        if (lane < 4) flags[lane] = 0;
        __syncthreads();
    }
    int it_begin = 0;
    if constexpr (OPT && REGULAR && CN_TYPE == FGNN_CN_BOXPLUS_PHI) {
        if (first_closed) {
            // the float operations of iteration 0, evaluated once per thread instead of once per edge: totals of zero messages
            // (:244-248), v->c (:254-273), then the phi rule on six equal inputs (:376-431) in the summation order of cn_phi_regular
            const float L = a.llr_const;
            const float Y = (0.0f + 0.0f) + L, X = 0.0f + L, Z = 0.0f + L;
            const float nu_x = MX::softplus(-X) - MX::lse2(-(Z - 0.0f), -(Y - 0.0f));
            const float nu_z = MX::softplus(-Z) - MX::lse2(-(X - 0.0f), -(Y - 0.0f));
            if (active)
                for (int c = lane_c; c < g.m; c += a.tpc) {
                    const unsigned synd = (c < g.m_x ? sx[c] : sz[c - g.m_x]) & 1u;
                    const uint4 pk = reinterpret_cast<const uint4*>(g.cslot16)[c];
                    const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
                    const float nu = c < g.m_x ? nu_x : nu_z;
                    const unsigned ng = nu < 0.0f;
                    unsigned neg = synd;
                    const float aa = MX::phi(FG_ABS(nu));
                    float T = 0.0f;
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        neg ^= ng;
                        T = T + aa;
                    }
                    const float val = with_sign(MX::phi(T - aa), neg ^ ng) * a.factor;
#pragma unroll
                    for (int j = 0; j < DC; ++j) slot_ref(msg, (int)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu)) = val;
                }
            a1 = FG_ABS(nu_x) >= FG_PHI_MAX && FG_ABS(nu_z) >= FG_PHI_MAX;  // "iteration 0's check phase was all-saturated"
            it_begin = 1;
            __syncthreads();
        }
    }
    const bool cn_one = a.cpb == 1, cn_f1 = a.factor == 1.0f;  // workgroup-uniform: which copy of the regular phi update runs
    // the syndrome bits of this thread's checks (lane_c, lane_c + tpc, ...) as one register: read once, not once per iteration
    unsigned synd_bits = 0;
    if (active) {
        int i = 0;
        for (int c = lane_c; c < g.m && i < 32; c += a.tpc, ++i) synd_bits |= ((c < g.m_x ? sx[c] : sz[c - g.m_x]) & 1u) << i;
    }
    const bool synd_in_reg = (g.m + a.tpc - 1) / a.tpc <= 32;
    // TRACE: soft syndromes of the current messages into slot k of the trace (cal_logit :455-471 on the totals of :244-248)
    auto trace_step = [&](const int k) __attribute__((always_inline)) {
        float* tlx = msg + a.trace_off;  // [n] llr_x of cal_logit, then [n] llr_z
        float* tlz = tlx + n;
        if (active && a.tape_x) {
            float* px = a.tape_x + ((size_t)k * a.B + b) * g.E_x;
            float* pz = a.tape_z + ((size_t)k * a.B + b) * g.E_z;
            for (int e = lane; e < g.E_x; e += a.tpc) px[e] = msg[e];
            for (int e = lane; e < g.E_z; e += a.tpc) pz[e] = msg[g.E_x + e];
        }
        if (active)
            for (int v = lane; v < n; v += a.tpc) {
                const int x0 = g.vptr_x[v], x1 = g.vptr_x[v + 1], z0 = g.vptr_z[v], z1 = g.vptr_z[v + 1];
                float Sz = 0.0f, Sx = 0.0f;
                for (int e = z0; e < z1; ++e) Sz = Sz + msg[e];
                for (int e = x0; e < x1; ++e) Sx = Sx + msg[e];
                const float lx = a.llr_ch ? Lch[v] : a.llr_const, ly = a.llr_ch ? Lch[n + v] : a.llr_const,
                            lz = a.llr_ch ? Lch[2 * n + v] : a.llr_const;
                const float Y = (Sz + Sx) + ly;
                const float X = Sz + lx;
                const float Z = Sx + lz;
                tlz[v] = MX::softplus(-X) - MX::lse2(-Z, -Y);
                tlx[v] = MX::softplus(-Z) - MX::lse2(-X, -Y);
            }
        __syncthreads();
        if (active) {
            float* tx = a.trace_x + ((size_t)k * a.B + b) * g.rows[0];
            for (int r = lane; r < g.rows[0]; r += a.tpc) {
                const int p0 = g.rptr[0][r];
                tx[r] = logit_row<HWT>(tlx, g.rcol[0] + p0, g.rptr[0][r + 1] - p0);
            }
            float* tz = a.trace_z + ((size_t)k * a.B + b) * g.rows[1];
            for (int r = lane; r < g.rows[1]; r += a.tpc) {
                const int p0 = g.rptr[1][r];
                tz[r] = logit_row<HWT>(tlz, g.rcol[1] + p0, g.rptr[1][r + 1] - p0);
            }
        }
        // the binary LLRs are next written one iteration (two barriers) later: no barrier needed here
    };
    if constexpr (TRACE) trace_step(0);
    for (int it = it_begin; it < a.num_iter; ++it) {
        bool changed = false, cn_slow = false;
        // ---- variable nodes: _vn_update (:227-275) ----
        auto vn_body = [&](const int v, const float lx, const float ly, const float lz) __attribute__((always_inline)) {
                if constexpr (REGULAR) {
                    float* px = msg + v * DVX;
                    float* pz = msg + g.E_x + v * DVZ;
                    float mx[DVX], mz[DVZ];
                    float Sz = 0.0f, Sx = 0.0f;
#pragma unroll
                    for (int k = 0; k < DVZ; ++k) { mz[k] = pz[k]; Sz = Sz + mz[k]; }
#pragma unroll
                    for (int k = 0; k < DVX; ++k) { mx[k] = px[k]; Sx = Sx + mx[k]; }
                    if (opt_exit) {
                        unsigned sig = 0;
#pragma unroll
                        for (int k = 0; k < DVX; ++k) sig |= sign_bit(mx[k]) << k;
#pragma unroll
                        for (int k = 0; k < DVZ; ++k) sig |= sign_bit(mz[k]) << (DVX + k);
                        if (sigw[v] != sig) {
                            changed = true;
                            sigw[v] = sig;
                        }
                    }
                    const float Y = (Sz + Sx) + ly;
                    const float X = Sz + lx;
                    const float Z = Sx + lz;
                    // Saturation shortcut (exact): |X|,|Z| beyond the softplus thresholds and every pair (Z_e,Y_e) /
                    // (X_e,Y_e) at least 20 apart => log(1+exp(-d)) is log(1) = 0 bit for bit and lse2 returns its max.
                    bool sat = opt_shortcut && FG_ABS(X) > FG_SOFTPLUS_THRESH && FG_ABS(Z) > FG_SOFTPLUS_THRESH;
                    if (shl) {  // shared form: ONE pair per side, the arguments fg_lse2_corr sees
                        sat = sat && FG_ABS((-Z) - (-Y)) >= 20.0f && FG_ABS((-X) - (-Y)) >= 20.0f;
                    } else {
#pragma unroll
                        for (int k = 0; k < DVX; ++k) sat = sat && (FG_ABS((Z - mx[k]) - (Y - mx[k])) >= 20.0f);
#pragma unroll
                        for (int k = 0; k < DVZ; ++k) sat = sat && (FG_ABS((X - mz[k]) - (Y - mz[k])) >= 20.0f);
                    }
                    if (opt_shortcut && __all(sat)) {
                        const float numx = softplus_saturated(-X);
                        const float numz = softplus_saturated(-Z);
#pragma unroll
                        for (int k = 0; k < DVX; ++k) {
                            const float Ze = Z - mx[k], Ye = Y - mx[k];
                            px[k] = numx - (0.0f + FG_MAX(-Ze, -Ye));
                        }
#pragma unroll
                        for (int k = 0; k < DVZ; ++k) {
                            const float Xe = X - mz[k], Ye = Y - mz[k];
                            pz[k] = numz - (0.0f + FG_MAX(-Xe, -Ye));
                        }
                        return;
                    }
                    const float numx = MX::softplus(-X);
                    const float numz = MX::softplus(-Z);
                    if (shl) {  // (Z - mu_e) - (Y - mu_e) = Z - Y for every edge: that part of the log-sum-exp once per side
                        const float cx = MX::lse2_corr(-Z, -Y), cz = MX::lse2_corr(-X, -Y);
#pragma unroll
                        for (int k = 0; k < DVX; ++k) {
                            const float Ze = Z - mx[k], Ye = Y - mx[k];
                            px[k] = numx - (cx + FG_MAX(-Ze, -Ye));
                        }
#pragma unroll
                        for (int k = 0; k < DVZ; ++k) {
                            const float Xe = X - mz[k], Ye = Y - mz[k];
                            pz[k] = numz - (cz + FG_MAX(-Xe, -Ye));
                        }
                        return;
                    }
                    // the reference's per-edge form (the library default).  Round 6, measured and NOT merged (profiles/
                    // r6_literal_kernel_attempts.txt): the three edges of a side staged like phi_n (exps, then the table reads, then the
                    // polynomials) with the range reduction forced onto full-rate VOP2 forms — BP4-64 44.97 ms against 45.06 ([[882,24]] x
                    // 65 536, run-to-run +-0.2), as in round 2 (45.30 against 45.36): at seven waves per SIMD nothing is left to hide
#pragma unroll
                    for (int k = 0; k < DVX; ++k) {
                        const float Ze = Z - mx[k], Ye = Y - mx[k];
                        px[k] = numx - MX::lse2(-Ze, -Ye);
                    }
#pragma unroll
                    for (int k = 0; k < DVZ; ++k) {
                        const float Xe = X - mz[k], Ye = Y - mz[k];
                        pz[k] = numz - MX::lse2(-Xe, -Ye);
                    }
                } else {
                    const int x0 = g.vptr_x[v], x1 = g.vptr_x[v + 1], z0 = g.vptr_z[v], z1 = g.vptr_z[v + 1];
                    float Sz = 0.0f, Sx = 0.0f;
                    for (int e = z0; e < z1; ++e) Sz = Sz + msg[e];
                    for (int e = x0; e < x1; ++e) Sx = Sx + msg[e];
                    const float Y = (Sz + Sx) + ly;
                    const float X = Sz + lx;
                    const float Z = Sx + lz;
                    if (opt_shortcut) {  // same exact shortcut (and sign words) as the regular path, runtime degrees
                        bool sat = FG_ABS(X) > FG_SOFTPLUS_THRESH && FG_ABS(Z) > FG_SOFTPLUS_THRESH;
                        if (shl) sat = sat && FG_ABS((-Z) - (-Y)) >= 20.0f && FG_ABS((-X) - (-Y)) >= 20.0f;
                        unsigned sig = 0;
                        for (int e = x0; e < x1; ++e) {
                            const float mm = msg[e];
                            sat = sat && (shl || FG_ABS((Z - mm) - (Y - mm)) >= 20.0f);
                            sig |= sign_bit(mm) << (e - x0);
                        }
                        for (int e = z0; e < z1; ++e) {
                            const float mm = msg[e];
                            sat = sat && (shl || FG_ABS((X - mm) - (Y - mm)) >= 20.0f);
                            sig |= sign_bit(mm) << ((x1 - x0) + (e - z0));
                        }
                        if (opt_exit && sigw[v] != sig) {
                            changed = true;
                            sigw[v] = sig;
                        }
                        if (__all(sat)) {
                            const float nx = softplus_saturated(-X), nz = softplus_saturated(-Z);
                            for (int e = x0; e < x1; ++e) {
                                const float mm = msg[e];
                                const float Ze = Z - mm, Ye = Y - mm;
                                msg[e] = nx - (0.0f + FG_MAX(-Ze, -Ye));
                            }
                            for (int e = z0; e < z1; ++e) {
                                const float mm = msg[e];
                                const float Xe = X - mm, Ye = Y - mm;
                                msg[e] = nz - (0.0f + FG_MAX(-Xe, -Ye));
                            }
                            return;
                        }
                    }
                    const float numx = MX::softplus(-X);
                    const float numz = MX::softplus(-Z);
                    if (shl) {
                        const float cx = MX::lse2_corr(-Z, -Y), cz = MX::lse2_corr(-X, -Y);
                        for (int e = x0; e < x1; ++e) {
                            float m = msg[e];
                            float Ze = Z - m, Ye = Y - m;
                            msg[e] = numx - (cx + FG_MAX(-Ze, -Ye));
                        }
                        for (int e = z0; e < z1; ++e) {
                            float m = msg[e];
                            float Xe = X - m, Ye = Y - m;
                            msg[e] = numz - (cz + FG_MAX(-Xe, -Ye));
                        }
                        return;
                    }
                    for (int e = x0; e < x1; ++e) {
                        float m = msg[e];
                        float Ze = Z - m, Ye = Y - m;
                        msg[e] = numx - MX::lse2(-Ze, -Ye);
                    }
                    for (int e = z0; e < z1; ++e) {
                        float m = msg[e];
                        float Xe = X - m, Ye = Y - m;
                        msg[e] = numz - MX::lse2(-Xe, -Ye);
                    }
                }
            };
        if (active) {
            if constexpr (LREG) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int v = lane + i * a.tpc;
                    if (v < n) vn_body(v, lreg[0][i], lreg[1][i], lreg[2][i]);
                }
            } else {
                for (int v = lane; v < n; v += a.tpc) {
                    if (a.llr_ch) vn_body(v, Lch[v], Lch[n + v], Lch[2 * n + v]);
                    else vn_body(v, a.llr_const, a.llr_const, a.llr_const);
                }
            }
        }
        if (opt_exit && changed) flags[2 * (it & 1)] = 1;
        __syncthreads();
        // ---- check nodes of both graphs (:752-767) ----
        if (opt_exit && lane == 0) {  // clear the other parity's flags: nobody reads or sets them during this phase
            flags[2 * ((it + 1) & 1)] = 0;
            flags[2 * ((it + 1) & 1) + 1] = 0;
        }
        // checks are dealt to the threads half a workgroup out of phase with the qubits: with 882 nodes on 4 waves two waves get
        // 4 slices of 64 and two get 3 — the qubit phase gives the extra slice to the low waves, the check phase to the high ones
        if (active) {
            int ci = 0;
            for (int c = lane_c; c < g.m; c += a.tpc, ++ci) {
                const unsigned synd = synd_in_reg ? (synd_bits >> ci) & 1u : (c < g.m_x ? sx[c] : sz[c - g.m_x]) & 1u;
                if constexpr (REGULAR) {
                    const uint4 pk = reinterpret_cast<const uint4*>(g.cslot16)[c];
                    const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
                    int sl[DC];  // byte offsets
#pragma unroll
                    for (int j = 0; j < DC; ++j) sl[j] = (int)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
                    if constexpr (CN_TYPE == FGNN_CN_BOXPLUS_PHI) {
                        // one codeword per workgroup: the message area starts at the (compile-time) base of the dynamic LDS, so the
                        // byte offsets are the LDS addresses up to an immediate; factor 1: no product
                        bool fast;
                        if (cn_one && cn_f1) fast = cn_phi_regular<DC, HWT, true>(lds, sl, synd, 1.0f, phi0, opt_shortcut);
                        else if (cn_one) fast = cn_phi_regular<DC, HWT, false>(lds, sl, synd, a.factor, phi0, opt_shortcut);
                        else fast = cn_phi_regular<DC, HWT, false>(msg, sl, synd, a.factor, phi0, opt_shortcut);
                        cn_slow = !fast || cn_slow;
                    } else if constexpr (CN_TYPE == FGNN_CN_MINSUM) {
                        cn_minsum_regular<DC>(msg, sl, synd, a.factor);
                    } else {
                        cn_tanh_regular<DC>(msg, sl, synd, a.factor);
                    }
                } else {
                    const int c0 = g.cptr[c], deg = g.cptr[c + 1] - c0;
                    if constexpr (CN_TYPE == FGNN_CN_BOXPLUS_PHI)
                        cn_slow = !cn_phi_generic<HWT>(msg, g.cslot + c0, deg, synd, a.factor, phi0, opt_shortcut) || cn_slow;
                    else cn_update<CN_TYPE, HWT>(msg, g.cslot + c0, deg, synd, a.factor);
                }
            }
        }
        if (opt_exit && cn_slow) flags[2 * (it & 1) + 1] = 1;
        __syncthreads();
        if (opt_exit) {
            const bool stable = a1 && a2 && flags[2 * (it & 1)] == 0;
            a2 = a1;
            a1 = flags[2 * (it & 1) + 1] == 0;
            if (stable) break;  // block-uniform: every thread reads the same LDS words after the barrier
        }
        if constexpr (TRACE) trace_step(it + 1);
    }

    // ---- marginals (:777), hard decision (:783-790), binary LLRs of cal_logit (:455-464) ----
    if (active) {
        if (a.msg_out_x)
            for (int e = lane; e < g.E_x; e += a.tpc) a.msg_out_x[(size_t)b * g.E_x + e] = msg[e];
        if (a.msg_out_z)
            for (int e = lane; e < g.E_z; e += a.tpc) a.msg_out_z[(size_t)b * g.E_z + e] = msg[g.E_x + e];
    }
    // The binary LLRs go to LDS where the messages were; totals are computed first by every thread,
    // parked in global memory (llr_out), and only then may the message area be overwritten.
    auto total_body = [&](const int v, const float lx, const float ly, const float lz) __attribute__((always_inline)) {
            const int x0 = g.vptr_x[v], x1 = g.vptr_x[v + 1], z0 = g.vptr_z[v], z1 = g.vptr_z[v + 1];
            float Sz = 0.0f, Sx = 0.0f;
            for (int e = z0; e < z1; ++e) Sz = Sz + msg[e];
            for (int e = x0; e < x1; ++e) Sx = Sx + msg[e];
            const float Y = (Sz + Sx) + ly;
            const float X = Sz + lx;
            const float Z = Sx + lz;
            float* o = a.llr_out + (size_t)b * 3 * n;
            o[v] = X;
            o[n + v] = Y;
            o[2 * n + v] = Z;
            int d = 0;
            float best = 0.0f;
            if (X < best) { best = X; d = 1; }
            if (Z < best) { best = Z; d = 2; }
            if (Y < best) { best = Y; d = 3; }
            a.x_hat[(size_t)b * n + v] = (uint8_t)(d & 1);
            a.z_hat[(size_t)b * n + v] = (uint8_t)(d >> 1);
    };
    if (active) {
        if constexpr (LREG) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int v = lane + i * a.tpc;
                if (v < n) total_body(v, lreg[0][i], lreg[1][i], lreg[2][i]);
            }
        } else {
            for (int v = lane; v < n; v += a.tpc) {
                if (a.llr_ch) total_body(v, Lch[v], Lch[n + v], Lch[2 * n + v]);
                else total_body(v, a.llr_const, a.llr_const, a.llr_const);
            }
        }
    }
    if (a.flagged) {
        // Fused flag test of the sandwich (feedback_gnn.py:324-328): does the estimate reproduce the measured syndrome?  hx rows
        // check z_hat (bit 1 of the decision), hz rows x_hat (bit 0).  The decisions of this codeword go to LDS — into the message
        // area once every thread is done reading messages (flag_off points there whenever the area is large enough: the n bytes
        // then cost no LDS of their own, which is what lets five [[882,24]] workgroups with per-qubit channel LLRs share a CU).
        uint8_t* dec = reinterpret_cast<uint8_t*>(msg + a.flag_off);
        unsigned* fword = reinterpret_cast<unsigned*>(dec + ((n + 3) & ~3));
        __syncthreads();
        if (active)
            for (int v = lane; v < n; v += a.tpc) {  // the thread's own decisions again, from its own writes
                dec[v] = (uint8_t)(a.x_hat[(size_t)b * n + v] | (a.z_hat[(size_t)b * n + v] << 1));
            }
        if (lane == 0) *fword = 0u;
        __syncthreads();
        unsigned mine = 0;
        if (active)
            for (int c = lane; c < g.m; c += a.tpc) {
                const int sh = c < g.m_x ? 1 : 0;
                unsigned par = (c < g.m_x ? sx[c] : sz[c - g.m_x]) & 1u;
                for (int e = g.cptr[c]; e < g.cptr[c + 1]; ++e) par ^= (dec[g.cvn[e]] >> sh) & 1u;
                mine |= par;
            }
        if (mine) atomicOr(fword, 1u);
        __syncthreads();
        if (active && lane == 0) a.flagged[b] = (uint8_t)(*fword != 0u);
    }
    if (!a.x_logit && !a.z_logit) return;
    __syncthreads();  // every thread is done reading messages
    float* llx = msg;      // [n] llr_x of cal_logit
    float* llz = msg + n;  // [n] llr_z
    if (active)
        for (int v = lane; v < n; v += a.tpc) {
            const float* o = a.llr_out + (size_t)b * 3 * n;
            const float X = o[v], Y = o[n + v], Z = o[2 * n + v];  // own writes: visible to this thread
            // same exact shortcut as in the qubit phase: softplus beyond its threshold, log(1 + exp(-d)) = log 1 = 0 for d >= 20
            const bool sat = opt_shortcut && FG_ABS(X) > FG_SOFTPLUS_THRESH && FG_ABS(Z) > FG_SOFTPLUS_THRESH &&
                             FG_ABS((-Z) - (-Y)) >= 20.0f && FG_ABS((-X) - (-Y)) >= 20.0f;
            if (opt_shortcut && __all(sat)) {
                llz[v] = softplus_saturated(-X) - (0.0f + FG_MAX(-Z, -Y));
                llx[v] = softplus_saturated(-Z) - (0.0f + FG_MAX(-X, -Y));
                continue;
            }
            llz[v] = MX::softplus(-X) - MX::lse2(-Z, -Y);
            llx[v] = MX::softplus(-Z) - MX::lse2(-X, -Y);
        }
    __syncthreads();
    if (active) {
        if (a.x_logit)
            for (int r = lane; r < g.rows[0]; r += a.tpc) {
                const int p0 = g.rptr[0][r];
                a.x_logit[(size_t)b * g.rows[0] + r] = logit_row_opt<HWT>(llx, g.rcol[0] + p0, g.rptr[0][r + 1] - p0, phi0, opt_shortcut);
            }
        if (a.z_logit)
            for (int r = lane; r < g.rows[1]; r += a.tpc) {
                const int p0 = g.rptr[1][r];
                a.z_logit[(size_t)b * g.rows[1] + r] = logit_row_opt<HWT>(llz, g.rcol[1] + p0, g.rptr[1][r + 1] - p0, phi0, opt_shortcut);
            }
    }
}

template <int CN_TYPE, int DVX, int DVZ, int DC>
int launch_bp4_k(const fgnn_graph* g, const BpArgs& a, const LaunchGeom& L, size_t lds_bytes, hipStream_t st)
{
    auto kern = a.shortcut ? bp4_kernel<CN_TYPE, DVX, DVZ, DC, true> : bp4_kernel<CN_TYPE, DVX, DVZ, DC, false>;
    if constexpr ((DVX == 3 && DVZ == 3 && DC == 6) || DVX == 0) {
        if (a.trace_x) kern = bp4_kernel<CN_TYPE, DVX, DVZ, DC, false, false, 0, true>;  // fixed dataflow, channel LLRs in LDS
    }
    if constexpr (CN_TYPE == FGNN_CN_BOXPLUS_PHI) {
        if (a.hwt) kern = bp4_kernel<CN_TYPE, DVX, DVZ, DC, false, true>;  // opt-in, fixed dataflow (fgnn_graph_set_option 3)
        if constexpr (DVX == 3 && DVZ == 3 && DC == 6) {
            // the benchmark codes' kernels carry the form of the qubit update as a compile-time argument (LSE = 0 literal / 1 shared)
            if (!a.trace_x && !a.hwt) {
                const int v = (a.shortcut ? 1 : 0) | (a.shared_lse ? 2 : 0);
                if (a.lreg == 4) {
                    kern = v == 0 ? bp4_kernel<CN_TYPE, 3, 3, 6, false, false, 4, false, false, 0> : v == 1 ? bp4_kernel<CN_TYPE, 3, 3, 6, true, false, 4, false, false, 0>
                         : v == 2 ? bp4_kernel<CN_TYPE, 3, 3, 6, false, false, 4, false, false, 1> : bp4_kernel<CN_TYPE, 3, 3, 6, true, false, 4, false, false, 1>;
                } else if (a.lreg == 5) {
                    kern = v == 0 ? bp4_kernel<CN_TYPE, 3, 3, 6, false, false, 5, false, false, 0> : v == 1 ? bp4_kernel<CN_TYPE, 3, 3, 6, true, false, 5, false, false, 0>
                         : v == 2 ? bp4_kernel<CN_TYPE, 3, 3, 6, false, false, 5, false, false, 1> : bp4_kernel<CN_TYPE, 3, 3, 6, true, false, 5, false, false, 1>;
                } else {
                    kern = v == 0 ? bp4_kernel<CN_TYPE, 3, 3, 6, false, false, 0, false, false, 0> : v == 1 ? bp4_kernel<CN_TYPE, 3, 3, 6, true, false, 0, false, false, 0>
                         : v == 2 ? bp4_kernel<CN_TYPE, 3, 3, 6, false, false, 0, false, false, 1> : bp4_kernel<CN_TYPE, 3, 3, 6, true, false, 0, false, false, 1>;
                }
            }
        }
    }
    if (lds_bytes > 48 * 1024)
        FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(L.blocks), dim3(L.threads), lds_bytes, st, g->d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

template <int CN_TYPE>
int launch_bp4(const fgnn_graph* g, const BpArgs& a, const LaunchGeom& L, size_t lds_bytes, hipStream_t st)
{
    const GraphDev& d = g->d;
    if (d.cslot16 && !g->force_generic) {
        if (d.dvx == 3 && d.dvz == 3 && d.dc == 6) return launch_bp4_k<CN_TYPE, 3, 3, 6>(g, a, L, lds_bytes, st);
        if (d.dvx == 4 && d.dvz == 4 && d.dc == 8 && !a.trace_x) return launch_bp4_k<CN_TYPE, 4, 4, 8>(g, a, L, lds_bytes, st);
    }
    return launch_bp4_k<CN_TYPE, 0, 0, 0>(g, a, L, lds_bytes, st);
}

}  // namespace

static int bp4_decode_core(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                           float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* msg_init_x,
                           const float* msg_init_z, float* llr_out, uint8_t* x_hat, uint8_t* z_hat, float* x_logit,
                           float* z_logit, float* msg_out_x, float* msg_out_z, const int* index, uint8_t* flagged,
                           float* trace_x, float* trace_z, float* tape_x, float* tape_z, void* stream)
{
    const bool trace = trace_x != nullptr;
    if (!g) return fgnn_fail(FGNN_ERR_ARG, "graph is NULL");
    if (B < 0 || num_iter < 0) return fgnn_fail(FGNN_ERR_ARG, "B and num_iter must be >= 0");
    if (cn_type < 0 || cn_type > 2) return fgnn_fail(FGNN_ERR_ARG, "Unknown node type.");  // decoding_q.py:107
    if ((x_logit && !g->d.rptr[0]) || (z_logit && !g->d.rptr[1]))
        return fgnn_fail(FGNN_ERR_STATE, "logit row sets not installed (fgnn_graph_set_rows)");
    if (B == 0) return FGNN_OK;  // an empty batch needs no buffers (an empty torch tensor's data pointer is NULL)
    if (!synd_x || !synd_z || !llr_out || !x_hat || !z_hat) return fgnn_fail(FGNN_ERR_ARG, "required buffer is NULL");
    FGNN_DEVICE_GUARD(g->device);
    LaunchGeom L = fgnn_geom(g, B);
    BpArgs a;
    a.B = B;
    a.num_iter = num_iter;
    a.tpc = L.tpc;
    a.cpb = L.cpb;
    a.factor = normalization_factor;
    a.llr_const = llr_const;
    a.llr_ch = llr_ch;
    a.synd_x = synd_x;
    a.synd_z = synd_z;
    a.msg_init_x = msg_init_x;
    a.msg_init_z = msg_init_z;
    a.llr_out = llr_out;
    a.x_hat = x_hat;
    a.z_hat = z_hat;
    a.x_logit = x_logit;
    a.z_logit = z_logit;
    a.msg_out_x = msg_out_x;
    a.msg_out_z = msg_out_z;
    a.index = index;
    a.trace_x = trace_x;
    a.trace_z = trace_z;
    a.trace_off = 0;
    a.tape_x = tape_x;
    a.tape_z = tape_z;
    a.gmem = nullptr;
    a.shared_lse = g->bp4_shared_lse ? 1 : 0;
    // the trace variant is the fixed dataflow on the shared float32 routines, channel LLRs in LDS
    a.hwt = (g->hw_transcendentals && cn_type == FGNN_CN_BOXPLUS_PHI && !trace) ? 1 : 0;
    a.shortcut = (g->shortcut && !a.hwt && !trace) ? 1 : 0;
    // floats per codeword: messages (>= 2n so the epilogue's binary LLRs fit) + channel LLRs (unless they fit the registers of
    // the NQ variant: regular (3,3,6) graph, phi rule, exact math, one codeword per workgroup, 4 or 5 qubits per thread)
    a.lch_off = g->d.E > 2 * g->d.n ? g->d.E : 2 * g->d.n;
    a.lreg = 0;
    {
        static const bool no_lreg = getenv("FGNN_BP4_NO_LREG") != nullptr;  // A/B knob (tools/ab_bp4_lch.py)
        const int per_thread = (g->d.n + L.tpc - 1) / L.tpc;
        if (llr_ch && !no_lreg && !trace && !a.hwt && cn_type == FGNN_CN_BOXPLUS_PHI && L.cpb == 1 && g->d.cslot16 && !g->force_generic &&
            g->d.dvx == 3 && g->d.dvz == 3 && g->d.dc == 6 && per_thread <= 5)
            a.lreg = per_thread <= 4 ? 4 : 5;
    }
    int per_cw = a.lch_off + ((llr_ch && !a.lreg) ? 3 * g->d.n : 0);
    if (trace) {  // 2n binary LLRs of the per-iteration soft syndromes, behind the messages and the channel LLRs
        a.trace_off = per_cw;
        per_cw += 2 * g->d.n;
    }
    per_cw = (per_cw + 3) & ~3;
    a.lds_per_cw = per_cw;
    size_t lds_bytes = (size_t)per_cw * sizeof(float) * (size_t)L.cpb;
    const bool regular = g->d.cslot16 && !g->force_generic &&
                         ((g->d.dvx == 3 && g->d.dvz == 3 && g->d.dc == 6) || (g->d.dvx == 4 && g->d.dvz == 4 && g->d.dc == 8));
    (void)regular;
    a.early_exit = (a.shortcut && g->early_exit && g->d.max_vdeg <= 32 && cn_type == FGNN_CN_BOXPLUS_PHI && L.cpb == 1 &&
                    num_iter > 2) ? 1 : 0;
    a.sig_off = per_cw;
    if (a.early_exit) {
        const size_t with_det = lds_bytes + ((size_t)g->d.n + 4) * sizeof(float);
        if (with_det <= FGNN_LDS_BUDGET) lds_bytes = with_det;
        else a.early_exit = 0;
    }
    a.flagged = flagged;
    a.flag_off = 0;
    if (flagged) {  // n decision bytes + one word per codeword: inside the message area (behind the 2n floats the soft-syndrome
                    // epilogue reuses) when it is large enough, else behind everything else
        if (L.cpb != 1) return fgnn_fail(FGNN_ERR_STATE, "the fused flag test needs one codeword per workgroup");
        const size_t need = (size_t)((g->d.n + 3) & ~3) + sizeof(unsigned);
        if ((size_t)a.lch_off * sizeof(float) >= (size_t)2 * g->d.n * sizeof(float) + need) {
            a.flag_off = 2 * g->d.n;
        } else {
            a.flag_off = (int)(lds_bytes / sizeof(float));
            lds_bytes += need;
        }
    }
    {  // experiment knob (tools/ab_bp4_lch.py): pad the dynamic LDS to lower the number of resident workgroups
        static const long pad = getenv("FGNN_BP4_LDS_PAD") ? atol(getenv("FGNN_BP4_LDS_PAD")) : 0;
        if (pad > 0 && lds_bytes + (size_t)pad <= FGNN_LDS_BUDGET) lds_bytes += (size_t)pad;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (lds_bytes > FGNN_LDS_BUDGET) {
        // The codeword's state does not fit a CU's LDS (about E + 3n > 40 000 floats): run the same runtime-degree kernel with that
        // state in a global-memory workspace row per workgroup slot (bp4_kernel<..., GMEM>).  Fixed dataflow, float32 routines of
        // fgnn_math.h, no trace: the reference's tensors are in device memory too, so this is its dataflow with the iterations fused
        // into one launch.  The workspace is stream-ordered (allocated and freed on the caller's stream, no synchronisation).
        if (trace) return fgnn_fail(FGNN_ERR_ARG, "code too large for the LDS-resident trace kernel");
        if (L.cpb != 1) return fgnn_fail(FGNN_ERR_STATE, "the global-memory BP4 variant needs one codeword per workgroup");
        a.shortcut = a.early_exit = a.hwt = a.lreg = 0;
        int per = a.lch_off + (llr_ch ? 3 * g->d.n : 0);
        a.sig_off = per;
        if (flagged) {  // n decision bytes + one word, behind the 2n floats the soft-syndrome epilogue reuses when they fit, else behind everything
            const size_t need = (size_t)((g->d.n + 3) & ~3) + sizeof(unsigned);
            if ((size_t)a.lch_off * sizeof(float) >= (size_t)2 * g->d.n * sizeof(float) + need) a.flag_off = 2 * g->d.n;
            else {
                a.flag_off = per;
                per += (int)((need + sizeof(float) - 1) / sizeof(float));
            }
        }
        per = (per + 3) & ~3;
        a.lds_per_cw = per;
        const size_t ws_bytes = (size_t)L.blocks * (size_t)per * sizeof(float);
        {   // one workspace row per codeword of the launch (254 KB for a [[6480,1296]] code): say so before an allocation of that size
            // fails as a bare HIP error — the caller decodes in smaller batches
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && ws_bytes > free_b)
                return fgnn_fail(FGNN_ERR_ARG, "code too large for the LDS-resident kernel: the global-memory variant needs " +
                                                   std::to_string(ws_bytes >> 20) + " MiB of workspace for " + std::to_string(B) +
                                                   " codewords (" + std::to_string((size_t)per * sizeof(float) >> 10) + " KiB each) and " +
                                                   std::to_string(free_b >> 20) + " MiB are free: decode in batches of at most " +
                                                   std::to_string(free_b / ((size_t)per * sizeof(float) * 2)) + " codewords");
        }
        void* ws = nullptr;
        FGNN_HIP_CHECK(hipMallocAsync(&ws, ws_bytes, st));
        a.gmem = static_cast<float*>(ws);
        fgnn_prof_scope prof(g, st);
        switch (cn_type) {
        case FGNN_CN_BOXPLUS_PHI:
            hipLaunchKernelGGL((bp4_kernel<FGNN_CN_BOXPLUS_PHI, 0, 0, 0, false, false, 0, false, true>), dim3(L.blocks), dim3(L.threads), 0, st, g->d, a);
            break;
        case FGNN_CN_MINSUM:
            hipLaunchKernelGGL((bp4_kernel<FGNN_CN_MINSUM, 0, 0, 0, false, false, 0, false, true>), dim3(L.blocks), dim3(L.threads), 0, st, g->d, a);
            break;
        default:
            hipLaunchKernelGGL((bp4_kernel<FGNN_CN_BOXPLUS, 0, 0, 0, false, false, 0, false, true>), dim3(L.blocks), dim3(L.threads), 0, st, g->d, a);
            break;
        }
        const hipError_t launched = hipGetLastError();
        (void)hipFreeAsync(ws, st);
        FGNN_HIP_CHECK(launched);
        prof.done(num_iter, B);
        return FGNN_OK;
    }
    fgnn_prof_scope prof(g, st);
    int rc;
    switch (cn_type) {
    case FGNN_CN_BOXPLUS_PHI: rc = launch_bp4<FGNN_CN_BOXPLUS_PHI>(g, a, L, lds_bytes, st); break;
    case FGNN_CN_MINSUM: rc = launch_bp4<FGNN_CN_MINSUM>(g, a, L, lds_bytes, st); break;
    default: rc = launch_bp4<FGNN_CN_BOXPLUS>(g, a, L, lds_bytes, st); break;
    }
    if (rc == FGNN_OK) prof.done(num_iter, B);
    return rc;
}

int fgnn_bp4_decode_impl(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                         float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* msg_init_x,
                         const float* msg_init_z, float* llr_out, uint8_t* x_hat, uint8_t* z_hat, float* x_logit,
                         float* z_logit, float* msg_out_x, float* msg_out_z, const int* index, uint8_t* flagged, void* stream)
{
    return bp4_decode_core(g, cn_type, num_iter, normalization_factor, llr_ch, llr_const, synd_x, synd_z, B, msg_init_x, msg_init_z,
                           llr_out, x_hat, z_hat, x_logit, z_logit, msg_out_x, msg_out_z, index, flagged, nullptr, nullptr, nullptr,
                           nullptr, stream);
}

extern "C" int fgnn_bp4_decode_trace(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor,
                                     const float* llr_ch, float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B,
                                     const float* msg_init_x, const float* msg_init_z, float* llr_out, uint8_t* x_hat,
                                     uint8_t* z_hat, float* x_logit_trace, float* z_logit_trace, float* tape_x,
                                     float* tape_z, void* stream)
{
    if (B == 0 && g) return FGNN_OK;
    if (!x_logit_trace || !z_logit_trace) return fgnn_fail(FGNN_ERR_ARG, "trace buffer is NULL");
    if ((tape_x == nullptr) != (tape_z == nullptr)) return fgnn_fail(FGNN_ERR_ARG, "tape_x and tape_z go together");
    if (g && (!g->d.rptr[0] || !g->d.rptr[1]))
        return fgnn_fail(FGNN_ERR_STATE, "logit row sets not installed (fgnn_graph_set_rows)");
    return bp4_decode_core(g, cn_type, num_iter, normalization_factor, llr_ch, llr_const, synd_x, synd_z, B, msg_init_x, msg_init_z,
                           llr_out, x_hat, z_hat, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x_logit_trace,
                           z_logit_trace, tape_x, tape_z, stream);
}

extern "C" int fgnn_bp4_decode(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor,
                               const float* llr_ch, float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B,
                               const float* msg_init_x, const float* msg_init_z, float* llr_out, uint8_t* x_hat,
                               uint8_t* z_hat, float* x_logit, float* z_logit, float* msg_out_x, float* msg_out_z,
                               void* stream)
{
    return fgnn_bp4_decode_impl(g, cn_type, num_iter, normalization_factor, llr_ch, llr_const, synd_x, synd_z, B, msg_init_x,
                                msg_init_z, llr_out, x_hat, z_hat, x_logit, z_logit, msg_out_x, msg_out_z, nullptr, nullptr, stream);
}
