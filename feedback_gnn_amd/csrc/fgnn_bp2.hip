// fgnn_bp2.hip — binary syndrome belief propagation on one Tanner graph, LDS-resident.
//
// Replaces LDPCBPDecoder.call with is_syndrome=True of /root/reference sionna/fec/ldpc/decoding.py:874-1048
// (the fork's additions to upstream Sionna: syndrome sign :905-908 applied inside the check-node rules :595, :658,
// :767; normalization_factor :991) and the BSC draw of BP_BSC_Model.call (feedback_gnn.py:213-214).  SURVEY.md §8(f)
// rank 1: same check-node rules as BP4 with one LLR per bit and a linear variable-node update.
//
// The parity-check matrix is side 0 (hx) of an fgnn_graph.  One workgroup holds the E_x messages of its codeword(s)
// in LDS for all iterations; the variable-node phase is a sum and a subtraction per edge, the check-node phase is
// the phi / min-sum / tanh rule (phi in its log(exp(x)+1) - log(exp(x)-1) form, decoding.py:632-633).
// Bit-identical to oracle/fgnn_oracle.c: og_bp2_decode.
#include "fgnn_internal.h"
#include "fgnn_math.h"
#include "fgnn_rng.h"

namespace {

struct Bp2Args {
    int B, num_iter, tpc, cpb, lds_per_cw;
    float factor, llr_const;
    const float* llr_ch;   // [B,n] logits or null
    const uint8_t* synd;   // [B,m_x] or null (all-zero syndrome)
    float* soft_out;       // [B,n] or null
    uint8_t* hard_out;     // [B,n] or null
};

__device__ __forceinline__ unsigned sign_bit(float x) { return fg_f2u(x) >> 31; }
__device__ __forceinline__ float with_sign(float mag, unsigned neg) { return fg_u2f(fg_f2u(mag) ^ (neg << 31)); }

template <int CN_TYPE>
__device__ __forceinline__ void cn_update2(float* msg, const int* __restrict__ slot, int deg, unsigned synd, float factor)
{
    if constexpr (CN_TYPE == FGNN_CN_BOXPLUS_PHI) {  // _cn_update_phi (decoding.py:637-693)
        unsigned neg = synd;
        float T = 0.0f;
        for (int j = 0; j < deg; ++j) {
            const int s = slot[j];
            const float v = msg[s];
            const unsigned ng = v < 0.0f;
            neg ^= ng;
            const float a = fg_phi_gnn(FG_ABS(v));
            T = T + a;
            msg[s] = with_sign(a, ng);
        }
        for (int j = 0; j < deg; ++j) {
            const int s = slot[j];
            const float w = msg[s];
            msg[s] = with_sign(fg_phi_gnn(T - FG_ABS(w)), neg ^ sign_bit(w)) * factor;
        }
    } else if constexpr (CN_TYPE == FGNN_CN_MINSUM) {  // _cn_update_minsum (decoding.py:744-850)
        const float LARGE = 10000.0f;
        unsigned neg = synd;
        float minv = 0.0f;
        for (int j = 0; j < deg; ++j) {
            const int s = slot[j];
            const float v = FG_MIN(FG_MAX(msg[s], -20.0f), 20.0f);
            const unsigned ng = v < 0.0f;
            neg ^= ng;
            const float a = FG_ABS(v);
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
        const float sg = (nsum > 0.0f) ? 1.0f : ((nsum < 0.0f) ? -1.0f : 0.0f);
        const float dm = 0.5f * (1.0f - sg);
        const float min_e = (1.0f - dm) * minv + dm * min2;
        for (int j = 0; j < deg; ++j) {
            const int s = slot[j];
            const float w = msg[s];
            const float d = FG_ABS(w) - minv;
            msg[s] = with_sign((d == 0.0f) ? min_e : minv, neg ^ sign_bit(w)) * factor;
        }
    } else {  // _cn_update_tanh (decoding.py:575-623)
        float P = 1.0f;
        for (int j = 0; j < deg; ++j) {
            const int s = slot[j];
            float t = fg_tanh(msg[s] / 2.0f);
            t = (t == 0.0f) ? 1e-12f : t;
            P = (j == 0) ? t : P * t;
            msg[s] = t;
        }
        P = P * (synd ? -1.0f : 1.0f);
        const float clipv = 0.99999988f;
        for (int j = 0; j < deg; ++j) {
            const int s = slot[j];
            float q = fg_rcp_unit(msg[s]) * P;
            q = (FG_ABS(q) < 1e-7f) ? 0.0f : q;
            q = FG_MIN(FG_MAX(q, -clipv), clipv);
            msg[s] = (2.0f * fg_atanh(q)) * factor;
        }
    }
}

template <int CN_TYPE>
__global__ void __launch_bounds__(1024) bp2_kernel(GraphDev g, Bp2Args a)
{
    FG_LOG_TAB_SETUP();
    extern __shared__ float lds[];
    const int cwl = threadIdx.x / a.tpc;
    const int lane = threadIdx.x - cwl * a.tpc;
    const int b = blockIdx.x * a.cpb + cwl;
    const bool active = b < a.B;
    float* msg = lds + (size_t)cwl * a.lds_per_cw;
    const int n = g.n, m = g.m_x;
    if (active)
        for (int e = lane; e < g.E_x; e += a.tpc) msg[e] = 0.0f;
    __syncthreads();
    for (int it = 0; it <= a.num_iter; ++it) {
        if (active)
            for (int v = lane; v < n; v += a.tpc) {
                float lc = a.llr_ch ? a.llr_ch[(size_t)b * n + v] : a.llr_const;
                lc = FG_MIN(FG_MAX(lc, -20.0f), 20.0f);
                const float L = -1.0f * lc;
                const int e0 = g.vptr_x[v], e1 = g.vptr_x[v + 1];
                float S = 0.0f;
                for (int e = e0; e < e1; ++e) S = S + msg[e];
                if (it == a.num_iter) {
                    const float o = -1.0f * (L + S);
                    if (a.soft_out) a.soft_out[(size_t)b * n + v] = o;
                    if (a.hard_out) a.hard_out[(size_t)b * n + v] = (uint8_t)(0.0f < o);
                    continue;
                }
                const float x = S + L;
                for (int e = e0; e < e1; ++e) msg[e] = x - msg[e];
            }
        if (it == a.num_iter) break;
        __syncthreads();
        if (active)
            for (int c = lane; c < m; c += a.tpc) {
                const int c0 = g.cptr[c], deg = g.cptr[c + 1] - c0;
                const unsigned sy = a.synd ? (a.synd[(size_t)b * m + c] & 1u) : 0u;
                cn_update2<CN_TYPE>(msg, g.cslot + c0, deg, sy, a.factor);
            }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) bsc_kernel(uint64_t seed, float p, uint64_t first, int B, int n, int nblk,
                                                  uint8_t* __restrict__ noise)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * nblk) return;
    const int b = (int)(t / nblk), blk = (int)(t - (long long)b * nblk);
    float u[4];
    fg_uniform4(seed, first + (uint64_t)b, (uint32_t)blk, u);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (blk * 4 + k < n) noise[(size_t)b * n + blk * 4 + k] = (uint8_t)(u[k] < p);
}

template <int CN_TYPE>
int launch(const fgnn_graph* g, const Bp2Args& a, const LaunchGeom& L, size_t lds_bytes, hipStream_t st)
{
    auto kern = bp2_kernel<CN_TYPE>;
    if (lds_bytes > 48 * 1024)
        FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(L.blocks), dim3(L.threads), lds_bytes, st, g->d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

}  // namespace

extern "C" int fgnn_bp2_decode(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                               float llr_const, const uint8_t* synd, int B, float* soft_out, uint8_t* hard_out, void* stream)
{
    if (!g) return fgnn_fail(FGNN_ERR_ARG, "graph is NULL");
    if (B < 0 || num_iter < 0) return fgnn_fail(FGNN_ERR_ARG, "B and num_iter must be >= 0");
    if (cn_type < 0 || cn_type > 2) return fgnn_fail(FGNN_ERR_ARG, "Unknown node type.");
    if (!soft_out && !hard_out) return fgnn_fail(FGNN_ERR_ARG, "no output buffer");
    if (B == 0) return FGNN_OK;
    FGNN_DEVICE_GUARD(g->device);
    LaunchGeom L = fgnn_geom(g, B);
    Bp2Args a;
    a.B = B;
    a.num_iter = num_iter;
    a.tpc = L.tpc;
    a.cpb = L.cpb;
    a.factor = normalization_factor;
    a.llr_const = llr_const;
    a.llr_ch = llr_ch;
    a.synd = synd;
    a.soft_out = soft_out;
    a.hard_out = hard_out;
    a.lds_per_cw = (g->d.E_x + 3) & ~3;
    const size_t lds_bytes = (size_t)a.lds_per_cw * sizeof(float) * (size_t)L.cpb;
    if (lds_bytes > FGNN_LDS_BUDGET) return fgnn_fail(FGNN_ERR_ARG, "code too large for the LDS-resident kernel");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (cn_type) {
    case FGNN_CN_BOXPLUS_PHI: return launch<FGNN_CN_BOXPLUS_PHI>(g, a, L, lds_bytes, st);
    case FGNN_CN_MINSUM: return launch<FGNN_CN_MINSUM>(g, a, L, lds_bytes, st);
    default: return launch<FGNN_CN_BOXPLUS>(g, a, L, lds_bytes, st);
    }
}

extern "C" int fgnn_bsc_noise(uint64_t seed, float p, uint64_t first_sample, int B, int n, uint8_t* noise, void* stream)
{
    if (B < 0 || n <= 0 || !noise) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    if (B == 0) return FGNN_OK;
    const int nblk = (n + 3) / 4;
    const long long total = (long long)B * nblk;
    hipLaunchKernelGGL(bsc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), seed, p,
                       first_sample, B, n, nblk, noise);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}
