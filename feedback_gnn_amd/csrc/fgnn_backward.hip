// fgnn_backward.hip — reverse pass of the second training stage: soft-syndrome losses -> BP4 iterations -> feedback GNN.
//
// The reference trains the GNN with tf.GradientTape through Second_Stage_GNN_BP_Model.call
// (/root/reference sionna/fec/ldpc/feedback_gnn.py:423-463; driver train_n882.ipynb cell 7): GNN -> 16 boxplus-phi BP4
// iterations in stage_two mode (per-iteration soft syndromes, decoding_q.py:455-471, :768-775) -> sum of BCE losses.
// Here the same chain rule is written out by hand, one workgroup per codeword with the state in LDS:
//
//   fgnn_bp4_backward     d loss / d llr_ch   from d loss / d(soft syndromes after k iterations), k = 0..T, given the
//                         recorded c->v messages mu^k (the forward is re-run one iteration per launch and its msg_out
//                         kept: "tape").  v->c messages and totals are recomputed from the tape, not stored.
//   fgnn_feedback_gnn_backward   per-edge / per-qubit activations and deltas of the four dense layers, from which the
//                         twelve weight gradients are plain X^T * Delta GEMMs (rocBLAS through the host framework).
//
// Gradient conventions are TensorFlow's (what the reference's tape applies): softplus' = sigmoid, reduce_logsumexp' =
// softmax, clip_by_value passes the gradient inside [lo,hi] and blocks it outside, |x|' = sign(x), the sign products are
// constants (tf.stop_gradient, decoding_q.py:392-409).  With phi(x) = -log tanh(x/2): phi'(x) = -1/sinh(x).
// Checked against autograd of the float64 restatement (oracle/torch_ref.py) in tests/test_gpu_backward.py; no bit-level
// claim is made for gradients (float32 here, tolerance in the test).
#include <cstring>

#include "fgnn_internal.h"
#include "fgnn_math.h"

namespace {

// expm1(t) for t >= 0 with the reduction of fg_tanh (full relative accuracy near 0)
__device__ __forceinline__ float bw_expm1(float t)
{
    t = FG_MIN(t, 60.0f);
    float tt = FG_FMA(t, FG_LOG2E, FG_RND_MAGIC);
    float k = tt - FG_RND_MAGIC;
    float r = FG_FMA(k, -FG_LN2_HI, t);
    r = FG_FMA(k, -FG_LN2_LO, r);
    float q = 1.381461043e-03f;
    q = FG_FMA(q, r, 8.368710056e-03f);
    q = FG_FMA(q, r, 4.166838899e-02f);
    q = FG_FMA(q, r, 1.666652113e-01f);
    q = FG_FMA(q, r, 4.999999404e-01f);
    float pm1 = FG_FMA(r * r, q, r);
    float sc = fg_u2f((fg_f2u(tt) << 23) + 0x3f800000u);
    return FG_FMA(sc, pm1, sc - 1.0f);
}

// d/dx of the clipped phi: -1/sinh(x) inside the clip interval, 0 outside
__device__ __forceinline__ float dphi(float x)
{
    if (!(x >= FG_PHI_MIN && x <= FG_PHI_MAX)) return 0.0f;
    const float em1 = bw_expm1(x);
    return -(2.0f * (em1 + 1.0f)) / (em1 * (em1 + 2.0f));
}

__device__ __forceinline__ float sigmoidf(float x)
{
    const float e = fg_exp(-FG_ABS(x));
    return x >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}

__device__ __forceinline__ float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

struct BwArgs {
    int B, T, nu_sz;
    float factor;
    const float* llr_ch;    // [B,3,n]
    const uint8_t* synd_x;  // [B,m_x]
    const uint8_t* synd_z;
    const float* tape_x;    // [T+1,B,E_x]  c->v messages before iteration k (k = T: after the last one)
    const float* tape_z;    // [T+1,B,E_z]
    const float* gx;        // [T+1,B,rows0]  d loss / d x_logit after k iterations, or null
    const float* gz;        // [T+1,B,rows1]
    const uint8_t* has_g;   // [T+1] host-evaluated: any non-zero gradient enters at k
    float* dllr;            // [B,3,n]
};

// One workgroup per codeword.  LDS: mu[E] | nu[max(E,2n)] | dmu[E] | tot[3n] | dtot[3n] | coef[rows0+rows1].
__global__ void __launch_bounds__(256) bp4_backward_kernel(GraphDev g, BwArgs a)
{
    FG_LOG_TAB_SETUP();
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n = g.n, E = g.E;
    float* mu = lds;
    float* nu = mu + E;
    float* dmu = nu + a.nu_sz;
    float* tot = dmu + E;
    float* dtot = tot + 3 * n;
    float* coef0 = dtot + 3 * n;
    float* coef1 = coef0 + g.rows[0];
    float* llx = nu;  // binary LLRs of cal_logit alias the v->c area (used before it is filled)
    float* llz = nu + n;

    const float* L = a.llr_ch + (size_t)b * 3 * n;
    float* dL = a.dllr + (size_t)b * 3 * n;
    const uint8_t* sx = a.synd_x + (size_t)b * g.m_x;
    const uint8_t* sz = a.synd_z + (size_t)b * g.m_z;
    for (int i = tid; i < 3 * n; i += nt) dL[i] = 0.0f;
    for (int e = tid; e < E; e += nt) dmu[e] = 0.0f;

    for (int k = a.T; k >= 0; --k) {
        const bool has_g = a.has_g[k] != 0;
        const bool has_it = k < a.T;
        __syncthreads();
        {  // P0: c->v messages entering iteration k
            const float* tx = a.tape_x + ((size_t)k * a.B + b) * g.E_x;
            const float* tz = a.tape_z + ((size_t)k * a.B + b) * g.E_z;
            for (int e = tid; e < g.E_x; e += nt) mu[e] = tx[e];
            for (int e = tid; e < g.E_z; e += nt) mu[g.E_x + e] = tz[e];
        }
        __syncthreads();
        // P1: totals (and the binary LLRs the soft syndromes are taken from)
        for (int v = tid; v < n; v += nt) {
            float Sz = 0.0f, Sx = 0.0f;
            for (int e = g.vptr_z[v]; e < g.vptr_z[v + 1]; ++e) Sz = Sz + mu[e];
            for (int e = g.vptr_x[v]; e < g.vptr_x[v + 1]; ++e) Sx = Sx + mu[e];
            const float X = Sz + L[v], Y = (Sz + Sx) + L[n + v], Z = Sx + L[2 * n + v];
            tot[v] = X;
            tot[n + v] = Y;
            tot[2 * n + v] = Z;
            dtot[v] = 0.0f;
            dtot[n + v] = 0.0f;
            dtot[2 * n + v] = 0.0f;
            if (has_g) {
                llz[v] = fg_softplus(-X) - fg_lse2(-Z, -Y);
                llx[v] = fg_softplus(-Z) - fg_lse2(-X, -Y);
            }
        }
        if (has_g) {
            __syncthreads();
            // P2: per row r, coef = dloss/dlogit_r * sign_r * phi'(T_r)   (cal_logit / _cn_update_phi_loss)
            for (int s = 0; s < 2; ++s) {
                const float* gl = s == 0 ? a.gx : a.gz;
                const float* ll = s == 0 ? llx : llz;
                float* coef = s == 0 ? coef0 : coef1;
                const int R = g.rows[s];
                for (int r = tid; r < R; r += nt) {
                    float c = 0.0f;
                    if (gl) {
                        unsigned neg = 0;
                        float T = 0.0f;
                        for (int p = g.rptr[s][r]; p < g.rptr[s][r + 1]; ++p) {
                            const float v = ll[g.rcol[s][p]];
                            neg ^= (v < 0.0f);
                            T = T + fg_phi(FG_ABS(v));
                        }
                        const float up = gl[((size_t)k * a.B + b) * R + r];
                        c = (neg ? -up : up) * dphi(T);
                    }
                    coef[r] = c;
                }
            }
            __syncthreads();
            // P3: rows -> binary LLRs -> totals
            for (int v = tid; v < n; v += nt) {
                float gxl = 0.0f, gzl = 0.0f;
                for (int p = g.tptr[0][v]; p < g.tptr[0][v + 1]; ++p) gxl = gxl + coef0[g.trow[0][p]];
                for (int p = g.tptr[1][v]; p < g.tptr[1][v + 1]; ++p) gzl = gzl + coef1[g.trow[1][p]];
                gxl = gxl * dphi(FG_ABS(llx[v])) * signf(llx[v]);
                gzl = gzl * dphi(FG_ABS(llz[v])) * signf(llz[v]);
                const float X = tot[v], Y = tot[n + v], Z = tot[2 * n + v];
                // llx = softplus(-Z) - lse(-X,-Y);  llz = softplus(-X) - lse(-Z,-Y)
                const float wx = sigmoidf(Y - X);  // weight of -X in lse(-X,-Y)
                const float wz = sigmoidf(Y - Z);
                float dX = gxl * wx - gzl * sigmoidf(-X);
                float dZ = gzl * wz - gxl * sigmoidf(-Z);
                float dY = gxl * (1.0f - wx) + gzl * (1.0f - wz);
                dtot[v] = dX;
                dtot[n + v] = dY;
                dtot[2 * n + v] = dZ;
            }
        }
        if (has_it) {
            __syncthreads();  // llx/llz are dead, nu may be written
            // P4: v->c messages of iteration k (_vn_update)
            for (int v = tid; v < n; v += nt) {
                const float X = tot[v], Y = tot[n + v], Z = tot[2 * n + v];
                const float numx = fg_softplus(-X), numz = fg_softplus(-Z);
                for (int e = g.vptr_x[v]; e < g.vptr_x[v + 1]; ++e) nu[e] = numx - fg_lse2(-(Z - mu[e]), -(Y - mu[e]));
                for (int e = g.vptr_z[v]; e < g.vptr_z[v + 1]; ++e) nu[e] = numz - fg_lse2(-(X - mu[e]), -(Y - mu[e]));
            }
            __syncthreads();
            // P5: check nodes: d loss / d mu^{k+1}  ->  d loss / d nu   (in place in dmu)
            for (int c = tid; c < g.m; c += nt) {
                const int c0 = g.cptr[c], c1 = g.cptr[c + 1];
                unsigned neg = (c < g.m_x ? sx[c] : sz[c - g.m_x]) & 1u;
                float T = 0.0f;
                for (int p = c0; p < c1; ++p) {
                    const float v = nu[g.cslot[p]];
                    neg ^= (v < 0.0f);
                    T = T + fg_phi(FG_ABS(v));
                }
                float U = 0.0f;
                for (int p = c0; p < c1; ++p) {
                    const int s = g.cslot[p];
                    const float v = nu[s];
                    const float aj = fg_phi(FG_ABS(v));
                    const bool ng = (neg ^ (unsigned)(v < 0.0f)) != 0;
                    float u = dmu[s] * a.factor * dphi(T - aj);
                    u = ng ? -u : u;
                    dmu[s] = u;
                    U = U + u;
                }
                for (int p = c0; p < c1; ++p) {
                    const int s = g.cslot[p];
                    const float v = nu[s];
                    dmu[s] = (U - dmu[s]) * dphi(FG_ABS(v)) * signf(v);
                }
            }
        }
        __syncthreads();
        // P6: variable nodes: d loss / d nu (+ soft-syndrome part)  ->  d loss / d mu^k and d loss / d llr_ch
        for (int v = tid; v < n; v += nt) {
            const float X = tot[v], Y = tot[n + v], Z = tot[2 * n + v];
            float dX = dtot[v], dY = dtot[n + v], dZ = dtot[2 * n + v];
            const int x0 = g.vptr_x[v], x1 = g.vptr_x[v + 1], z0 = g.vptr_z[v], z1 = g.vptr_z[v + 1];
            if (has_it) {
                const float sgX = sigmoidf(-X), sgZ = sigmoidf(-Z);
                for (int e = x0; e < x1; ++e) {  // nu = softplus(-X) - lse(-(Z-mu), -(Y-mu))
                    const float d = dmu[e];
                    const float w = sigmoidf(Y - Z);  // weight of -(Z-mu) against -(Y-mu); mu cancels
                    dX = dX - d * sgX;
                    dZ = dZ + d * w;
                    dY = dY + d * (1.0f - w);
                }
                for (int e = z0; e < z1; ++e) {  // nu = softplus(-Z) - lse(-(X-mu), -(Y-mu))
                    const float d = dmu[e];
                    const float w = sigmoidf(Y - X);
                    dZ = dZ - d * sgZ;
                    dX = dX + d * w;
                    dY = dY + d * (1.0f - w);
                }
            }
            // X = Sz + Lx, Y = Sz + Sx + Ly, Z = Sx + Lz;  an hx message sits in Sx, an hz message in Sz
            for (int e = x0; e < x1; ++e) dmu[e] = (has_it ? -dmu[e] : 0.0f) + (dZ + dY);
            for (int e = z0; e < z1; ++e) dmu[e] = (has_it ? -dmu[e] : 0.0f) + (dX + dY);
            dL[v] += dX;
            dL[n + v] += dY;
            dL[2 * n + v] += dZ;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Feedback GNN (feedback_gnn.py:161-188): re-run the forward per qubit, push d loss / d out through _llr_inv_embed,
// vn_embed_mlp, the mean and the two edge MLPs, and leave (activation, delta) pairs of every dense layer in HBM.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HID = 40, MSG = 20;

struct GnnBwArgs {
    int B;
    const float* llr;       // [B,3,n]
    const float* logit_hx;  // [B,m_x]
    const float* logit_hz;
    const uint8_t* synd_x;
    const uint8_t* synd_z;
    const float* gout;      // [B,3,n]
    float* node_in;         // [B,n,44]  [mean_x | mean_z | X Y Z | 0]
    float* node_h2;         // [B,n,40]
    float* node_d2;         // [B,n,40]  delta at the pre-activation of vn_embed_mlp
    float* e_feat[2];       // [B,E_s,4]
    float* e_h1[2];         // [B,E_s,40]
    float* e_d1[2];         // [B,E_s,40]
    float* e_dm[2];         // [B,E_s,20] delta at the output of the edge MLP
};

__global__ void __launch_bounds__(256) gnn_backward_kernel(GraphDev g, WeightsDev w, GnnBwArgs a)
{
    extern __shared__ float lds[];
    float* gcn = lds;  // [m_x] then [m_z]: h_cn of :168-172
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, n = g.n;
    for (int c = tid; c < g.m_x; c += nt)
        gcn[c] = a.logit_hx[(size_t)b * g.m_x + c] * ((a.synd_x[(size_t)b * g.m_x + c] & 1) ? -1.0f : 1.0f);
    for (int c = tid; c < g.m_z; c += nt)
        gcn[g.m_x + c] = a.logit_hz[(size_t)b * g.m_z + c] * ((a.synd_z[(size_t)b * g.m_z + c] & 1) ? -1.0f : 1.0f);
    __syncthreads();
    const float* in = a.llr + (size_t)b * 3 * n;
    const float* go = a.gout + (size_t)b * 3 * n;
    for (int v = tid; v < n; v += nt) {
        const float X = in[v], Y = in[n + v], Z = in[2 * n + v];
        const float g0 = go[v], g1 = go[n + v], g2 = go[2 * n + v];
        float z[44];
        // ---- forward: mean messages of both sides ----
        for (int s = 0; s < 2; ++s) {
            const int* vptr = s == 0 ? g.vptr_x : g.vptr_z;
            const float* gc = s == 0 ? gcn : gcn + g.m_x;
            const int e0 = vptr[v], e1 = vptr[v + 1];
            float acc[MSG];
#pragma unroll
            for (int i = 0; i < MSG; ++i) acc[i] = 0.0f;
            for (int e = e0; e < e1; ++e) {
                const float hc = gc[g.vchk[e]];
                float m[MSG];
#pragma unroll
                for (int i = 0; i < MSG; ++i) m[i] = 0.0f;
                for (int j = 0; j < HID; ++j) {
                    const float* r = w.w1t[s] + j * 4;
                    float p = FG_FMA(Z, r[3], FG_FMA(Y, r[2], FG_FMA(X, r[1], hc * r[0])));
                    const float h = fg_tanh(p + w.b1[s][j]);
                    const float* r2 = w.w2[s] + j * MSG;
#pragma unroll
                    for (int i = 0; i < MSG; ++i) m[i] = FG_FMA(h, r2[i], m[i]);
                }
#pragma unroll
                for (int i = 0; i < MSG; ++i) acc[i] = acc[i] + (m[i] + w.b2[s][i]);
            }
            const float fd = (float)(e1 - e0);
#pragma unroll
            for (int i = 0; i < MSG; ++i) z[s * MSG + i] = (e1 > e0) ? acc[i] / fd : 0.0f;
        }
        z[40] = X;
        z[41] = Y;
        z[42] = Z;
        z[43] = 0.0f;
        float* zo = a.node_in + ((size_t)b * n + v) * 44;
#pragma unroll
        for (int k = 0; k < 44; ++k) zo[k] = z[k];
        // ---- node MLP forward + backward ----
        float dz[2 * MSG];
#pragma unroll
        for (int k = 0; k < 2 * MSG; ++k) dz[k] = 0.0f;
        float* h2o = a.node_h2 + ((size_t)b * n + v) * HID;
        float* d2o = a.node_d2 + ((size_t)b * n + v) * HID;
        for (int j = 0; j < HID; ++j) {
            const float* r = w.wet + j * 44;
            float p = 0.0f;
#pragma unroll
            for (int k = 0; k < 43; ++k) p = FG_FMA(z[k], r[k], p);
            const float h = fg_tanh(p + w.be[j]);
            const float* ro = w.wout + j * 4;
            const float dh = FG_FMA(g2, ro[2], FG_FMA(g1, ro[1], g0 * ro[0]));
            const float d = dh * (1.0f - h * h);
            h2o[j] = h;
            d2o[j] = d;
#pragma unroll
            for (int k = 0; k < 2 * MSG; ++k) dz[k] = FG_FMA(r[k], d, dz[k]);
        }
        // ---- edge MLPs backward (activations recomputed) ----
        for (int s = 0; s < 2; ++s) {
            const int* vptr = s == 0 ? g.vptr_x : g.vptr_z;
            const float* gc = s == 0 ? gcn : gcn + g.m_x;
            const int e0 = vptr[v], e1 = vptr[v + 1], Es = s == 0 ? g.E_x : g.E_z, ebase = s == 0 ? 0 : g.E_x;
            const float fd = (float)(e1 - e0);
            float dm[MSG];
#pragma unroll
            for (int i = 0; i < MSG; ++i) dm[i] = dz[s * MSG + i] / fd;
            for (int e = e0; e < e1; ++e) {
                const size_t row = (size_t)b * Es + (e - ebase);
                const float hc = gc[g.vchk[e]];
                float* fo = a.e_feat[s] + row * 4;
                fo[0] = hc;
                fo[1] = X;
                fo[2] = Y;
                fo[3] = Z;
                float* dmo = a.e_dm[s] + row * MSG;
#pragma unroll
                for (int i = 0; i < MSG; ++i) dmo[i] = dm[i];
                float* h1o = a.e_h1[s] + row * HID;
                float* d1o = a.e_d1[s] + row * HID;
                for (int j = 0; j < HID; ++j) {
                    const float* r = w.w1t[s] + j * 4;
                    float p = FG_FMA(Z, r[3], FG_FMA(Y, r[2], FG_FMA(X, r[1], hc * r[0])));
                    const float h = fg_tanh(p + w.b1[s][j]);
                    const float* r2 = w.w2[s] + j * MSG;
                    float dh = 0.0f;
#pragma unroll
                    for (int i = 0; i < MSG; ++i) dh = FG_FMA(r2[i], dm[i], dh);
                    h1o[j] = h;
                    d1o[j] = dh * (1.0f - h * h);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Reverse pass of the RUNTIME-SHAPED feedback GNN (fgnn_weights_create_general: any num_msg_dims / num_hidden_units / num_mlp_layers /
// reduce_op / activation / use_bias the reference's constructor accepts) — the reference trains whatever Feedback_GNN.__init__ built
// (feedback_gnn.py:423-463 under tf.GradientTape).  Same division of labour as gnn_backward_kernel: one thread per qubit recomputes the
// forward (the float operations of gnn_general_kernel) and leaves, for every Dense layer in execution order, its input activations
// and the gradient at its pre-activation in HBM; the sums over batch x edges (weight gradient = activations^T deltas, bias gradient =
// column sums of the deltas) are library GEMMs on the caller's side.  Reduce ops: sum / mean pass the gradient through (/ deg for
// mean); max / min route it to the edges that attain the extremum, shared equally among ties (TensorFlow's _MinOrMaxGrad).
// Activation derivatives are formed from the activation's output (tanh: 1 - h^2, sigmoid: h (1 - h), relu: h > 0, linear: 1).
// The compatibility path: activations of a whole MLP live in per-thread scratch.
struct GnnGenBwArgs {
    int B;
    const float* llr;
    const float* logit_hx;
    const float* logit_hz;
    const uint8_t* synd_x;
    const uint8_t* synd_z;
    const float* gout;                      // [B,3,n]
    float* acts[FGNN_GEN_MAX_LAYERS];       // layer li: [rows_li, K_li]; rows = B*E_x | B*E_z (message MLPs), B*n (embed MLP, _llr_inv_embed)
    float* deltas[FGNN_GEN_MAX_LAYERS];     // layer li: [rows_li, J_li]
};

__device__ __forceinline__ float gen_act_bw(float a, int act)
{
    switch (act) {
    case FGNN_ACT_TANH: return fg_tanh(a);
    case FGNN_ACT_RELU: return FG_MAX(a, 0.0f);
    case FGNN_ACT_SIGMOID: return fg_sigmoid(a);
    default: return a;
    }
}
__device__ __forceinline__ float gen_act_deriv(float h, int act)
{
    switch (act) {
    case FGNN_ACT_TANH: return 1.0f - h * h;
    case FGNN_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
    case FGNN_ACT_SIGMOID: return h * (1.0f - h);
    default: return 1.0f;
    }
}
__device__ __forceinline__ void gen_dense_bw(const GnnGeneralDev& w, int li, const float* in, float* out)
{
    const int K = w.K[li], J = w.J[li], act = w.act_l[li];
    const float* W = w.W[li];
    const float* b = w.b[li];
    for (int j = 0; j < J; ++j) {
        float a = 0.0f;
        for (int k = 0; k < K; ++k) a = FG_FMA(in[k], W[k * J + j], a);
        if (b) a = a + b[j];
        out[j] = gen_act_bw(a, act);
    }
}
// d in[k] = sum_j W[k][j] delta[j]
__device__ __forceinline__ void gen_dense_back(const GnnGeneralDev& w, int li, const float* delta, float* din)
{
    const int K = w.K[li], J = w.J[li];
    const float* W = w.W[li];
    for (int k = 0; k < K; ++k) {
        float a = 0.0f;
        for (int j = 0; j < J; ++j) a = FG_FMA(W[k * J + j], delta[j], a);
        din[k] = a;
    }
}

constexpr int GBW_W = FGNN_GEN_MAX_W > 2 * FGNN_GEN_MAX_D + 3 ? FGNN_GEN_MAX_W : 2 * FGNN_GEN_MAX_D + 3;  // widest activation vector

__global__ void __launch_bounds__(256) gnn_general_backward_kernel(GraphDev g, GnnGeneralDev w, GnnGenBwArgs a)
{
    extern __shared__ float lds[];
    float* gcn = lds;  // [m_x] then [m_z]: h_cn of :168-172
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, n = g.n;
    for (int c = tid; c < g.m_x; c += nt)
        gcn[c] = a.logit_hx[(size_t)b * g.m_x + c] * ((a.synd_x[(size_t)b * g.m_x + c] & 1) ? -1.0f : 1.0f);
    for (int c = tid; c < g.m_z; c += nt)
        gcn[g.m_x + c] = a.logit_hz[(size_t)b * g.m_z + c] * ((a.synd_z[(size_t)b * g.m_z + c] & 1) ? -1.0f : 1.0f);
    __syncthreads();
    const int D = w.D, L = w.L, rop = w.reduce_op;
    const bool extremum = rop == FGNN_REDUCE_MAX || rop == FGNN_REDUCE_MIN;
    const float* in = a.llr + (size_t)b * 3 * n;
    const float* go = a.gout + (size_t)b * 3 * n;
    float ha[5][GBW_W];                 // activations of the MLP at hand: ha[0] = its input, ha[l + 1] = output of its layer l
    float z[2 * FGNN_GEN_MAX_D + 3];    // [reduced_x | reduced_z | X Y Z]
    float cnt[2][FGNN_GEN_MAX_D];       // max / min: how many edges attain the extremum, per component
    float dz[GBW_W], dA[GBW_W], dB[GBW_W];
    for (int v = tid; v < n; v += nt) {
        const float X = in[v], Y = in[n + v], Z = in[2 * n + v];
        // ---- forward of the message MLPs and the reduce (gnn_general_kernel's operations) ----
        for (int s = 0; s < 2; ++s) {
            const int* vptr = s == 0 ? g.vptr_x : g.vptr_z;
            const float* gc = s == 0 ? gcn : gcn + g.m_x;
            const int e0 = vptr[v], e1 = vptr[v + 1];
            float* acc = z + s * D;
            for (int i = 0; i < D; ++i) acc[i] = 0.0f;
            for (int e = e0; e < e1; ++e) {
                ha[0][0] = gc[g.vchk[e]];
                ha[0][1] = X;
                ha[0][2] = Y;
                ha[0][3] = Z;
                for (int l = 0; l < L; ++l) gen_dense_bw(w, s * L + l, ha[l], ha[l + 1]);
                for (int i = 0; i < D; ++i) {
                    const float m = ha[L][i];
                    float r;
                    if (e == e0) r = m;
                    else if (rop == FGNN_REDUCE_MAX) r = FG_MAX(acc[i], m);
                    else if (rop == FGNN_REDUCE_MIN) r = FG_MIN(acc[i], m);
                    else r = acc[i] + m;
                    acc[i] = r;
                }
            }
            if (rop == FGNN_REDUCE_MEAN && e1 > e0) {
                const float fd = (float)(e1 - e0);
                for (int i = 0; i < D; ++i) acc[i] = acc[i] / fd;
            }
            if (extremum) {  // ties share the gradient: count the edges that attain the extremum
                for (int i = 0; i < D; ++i) cnt[s][i] = 0.0f;
                for (int e = e0; e < e1; ++e) {
                    ha[0][0] = gc[g.vchk[e]];
                    ha[0][1] = X;
                    ha[0][2] = Y;
                    ha[0][3] = Z;
                    for (int l = 0; l < L; ++l) gen_dense_bw(w, s * L + l, ha[l], ha[l + 1]);
                    for (int i = 0; i < D; ++i) cnt[s][i] += (ha[L][i] == acc[i]) ? 1.0f : 0.0f;
                }
            }
        }
        z[2 * D] = X;
        z[2 * D + 1] = Y;
        z[2 * D + 2] = Z;
        // ---- embed MLP + _llr_inv_embed: forward with every activation kept, then backward ----
        const size_t nrow = (size_t)b * n + v;
        for (int k = 0; k < 2 * D + 3; ++k) ha[0][k] = z[k];
        for (int l = 0; l < L - 1; ++l) gen_dense_bw(w, 2 * L + l, ha[l], ha[l + 1]);
        for (int l = 0; l < L; ++l) {  // inputs of the L - 1 embed layers and of _llr_inv_embed (layer 3L - 1, input ha[L - 1])
            const int li = l < L - 1 ? 2 * L + l : 3 * L - 1, K = w.K[li];
            float* o = a.acts[li] + nrow * K;
            for (int k = 0; k < K; ++k) o[k] = ha[l][k];
        }
        {
            const int li = 3 * L - 1;
            float* d = a.deltas[li] + nrow * 3;
            dA[0] = go[v];
            dA[1] = go[n + v];
            dA[2] = go[2 * n + v];
            d[0] = dA[0];
            d[1] = dA[1];
            d[2] = dA[2];
            gen_dense_back(w, li, dA, dB);  // d ha[L - 1]
        }
        float* dcur = dB;
        float* dnxt = dA;
        for (int l = L - 2; l >= 0; --l) {
            const int li = 2 * L + l, J = w.J[li];
            float* d = a.deltas[li] + nrow * J;
            for (int j = 0; j < J; ++j) {
                dnxt[j] = dcur[j] * gen_act_deriv(ha[l + 1][j], w.act_l[li]);
                d[j] = dnxt[j];
            }
            gen_dense_back(w, li, dnxt, dcur);  // d ha[l] (dcur is free again: its values went into dnxt)
        }
        for (int k = 0; k < 2 * D; ++k) dz[k] = dcur[k];
        // ---- message MLPs backward (activations recomputed per edge) ----
        for (int s = 0; s < 2; ++s) {
            const int* vptr = s == 0 ? g.vptr_x : g.vptr_z;
            const float* gc = s == 0 ? gcn : gcn + g.m_x;
            const int e0 = vptr[v], e1 = vptr[v + 1], Es = s == 0 ? g.E_x : g.E_z, ebase = s == 0 ? 0 : g.E_x;
            const float fd = (float)(e1 - e0);
            for (int e = e0; e < e1; ++e) {
                const size_t row = (size_t)b * Es + (e - ebase);
                ha[0][0] = gc[g.vchk[e]];
                ha[0][1] = X;
                ha[0][2] = Y;
                ha[0][3] = Z;
                for (int l = 0; l < L; ++l) gen_dense_bw(w, s * L + l, ha[l], ha[l + 1]);
                for (int l = 0; l < L; ++l) {
                    const int li = s * L + l, K = w.K[li];
                    float* o = a.acts[li] + row * K;
                    for (int k = 0; k < K; ++k) o[k] = ha[l][k];
                }
                // gradient at the message (output of the last, linear layer) through the reduce
                for (int i = 0; i < D; ++i) {
                    const float gr = dz[s * D + i];
                    float dm;
                    if (rop == FGNN_REDUCE_MEAN) dm = gr / fd;
                    else if (extremum) dm = (ha[L][i] == z[s * D + i]) ? gr / cnt[s][i] : 0.0f;
                    else dm = gr;
                    dA[i] = dm;
                }
                float* dc = dA;
                float* dn = dB;
                for (int l = L - 1; l >= 0; --l) {
                    const int li = s * L + l, J = w.J[li];
                    float* d = a.deltas[li] + row * J;
                    if (l < L - 1)
                        for (int j = 0; j < J; ++j) dc[j] = dc[j] * gen_act_deriv(ha[l + 1][j], w.act_l[li]);
                    for (int j = 0; j < J; ++j) d[j] = dc[j];
                    if (l > 0) {
                        gen_dense_back(w, li, dc, dn);
                        float* t = dc;
                        dc = dn;
                        dn = t;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int fgnn_bp4_backward(const fgnn_graph* g, int num_iter, float normalization_factor, const float* llr_ch,
                                 const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* tape_x,
                                 const float* tape_z, const float* grad_x_logit, const float* grad_z_logit,
                                 const uint8_t* has_grad, float* grad_llr_ch, void* stream)
{
    if (!g) return fgnn_fail(FGNN_ERR_ARG, "graph is NULL");
    if (B < 0 || num_iter < 0) return fgnn_fail(FGNN_ERR_ARG, "B and num_iter must be >= 0");
    if (!llr_ch || !synd_x || !synd_z || !tape_x || !tape_z || !has_grad || !grad_llr_ch)
        return fgnn_fail(FGNN_ERR_ARG, "required buffer is NULL");
    if (!g->d.rptr[0] || !g->d.rptr[1] || !g->d.tptr[0] || !g->d.tptr[1])
        return fgnn_fail(FGNN_ERR_STATE, "logit row sets not installed (fgnn_graph_set_rows)");
    if (B == 0) return FGNN_OK;
    FGNN_DEVICE_GUARD(g->device);
    const GraphDev& d = g->d;
    BwArgs a;
    a.B = B;
    a.T = num_iter;
    a.nu_sz = d.E > 2 * d.n ? d.E : 2 * d.n;
    a.factor = normalization_factor;
    a.llr_ch = llr_ch;
    a.synd_x = synd_x;
    a.synd_z = synd_z;
    a.tape_x = tape_x;
    a.tape_z = tape_z;
    a.gx = grad_x_logit;
    a.gz = grad_z_logit;
    a.has_g = has_grad;
    a.dllr = grad_llr_ch;
    const size_t floats = (size_t)2 * d.E + a.nu_sz + (size_t)6 * d.n + d.rows[0] + d.rows[1];
    const size_t lds_bytes = floats * sizeof(float);
    if (lds_bytes > FGNN_LDS_BUDGET) return fgnn_fail(FGNN_ERR_ARG, "code too large for the LDS-resident backward kernel");
    auto kern = bp4_backward_kernel;
    if (lds_bytes > 48 * 1024)
        FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_feedback_gnn_backward(const fgnn_graph* g, const fgnn_weights* w, const float* llr,
                                          const float* logit_hx, const float* logit_hz, const uint8_t* synd_x,
                                          const uint8_t* synd_z, int B, const float* grad_out, float* node_in,
                                          float* node_h2, float* node_d2, float* const edge_feat[2],
                                          float* const edge_h1[2], float* const edge_d1[2], float* const edge_dm[2],
                                          void* stream)
{
    if (!g || !w) return fgnn_fail(FGNN_ERR_ARG, "graph or weights is NULL");
    if (B == 0) return FGNN_OK;  // an empty batch needs no buffers
    if (!llr || !logit_hx || !logit_hz || !synd_x || !synd_z || !grad_out || !node_in || !node_h2 || !node_d2 ||
        !edge_feat || !edge_h1 || !edge_d1 || !edge_dm)
        return fgnn_fail(FGNN_ERR_ARG, "buffer is NULL");
    for (int s = 0; s < 2; ++s)
        if (!edge_feat[s] || !edge_h1[s] || !edge_d1[s] || !edge_dm[s]) return fgnn_fail(FGNN_ERR_ARG, "buffer is NULL");
    if (B < 0) return fgnn_fail(FGNN_ERR_ARG, "B must be >= 0");
    if (w->device != g->device) return fgnn_fail(FGNN_ERR_ARG, "weights and graph live on different devices");
    if (w->general)
        return fgnn_fail(FGNN_ERR_ARG, "the reverse pass exists for the num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, "
                                       "mean, tanh, use_bias configuration only");
    if (B == 0) return FGNN_OK;
    FGNN_DEVICE_GUARD(g->device);
    GnnBwArgs a;
    a.B = B;
    a.llr = llr;
    a.logit_hx = logit_hx;
    a.logit_hz = logit_hz;
    a.synd_x = synd_x;
    a.synd_z = synd_z;
    a.gout = grad_out;
    a.node_in = node_in;
    a.node_h2 = node_h2;
    a.node_d2 = node_d2;
    for (int s = 0; s < 2; ++s) {
        a.e_feat[s] = edge_feat[s];
        a.e_h1[s] = edge_h1[s];
        a.e_d1[s] = edge_d1[s];
        a.e_dm[s] = edge_dm[s];
    }
    const size_t lds_bytes = (size_t)g->d.m * sizeof(float);
    if (lds_bytes > FGNN_LDS_BUDGET) return fgnn_fail(FGNN_ERR_ARG, "too many checks for the LDS-resident kernel");
    auto kern = gnn_backward_kernel;
    if (lds_bytes > 48 * 1024)
        FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), g->d, w->d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_feedback_gnn_backward_general(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                                                  const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B,
                                                  const float* grad_out, float* const* acts, float* const* deltas, int num_layers,
                                                  void* stream)
{
    if (!g || !w) return fgnn_fail(FGNN_ERR_ARG, "graph or weights is NULL");
    if (!w->general) return fgnn_fail(FGNN_ERR_ARG, "fgnn_feedback_gnn_backward_general takes weights made by fgnn_weights_create_general");
    if (B < 0) return fgnn_fail(FGNN_ERR_ARG, "B must be >= 0");
    if (B == 0) return FGNN_OK;  // an empty batch needs no buffers (an empty torch tensor's data pointer is NULL)
    if (!llr || !logit_hx || !logit_hz || !synd_x || !synd_z || !grad_out || !acts || !deltas) return fgnn_fail(FGNN_ERR_ARG, "buffer is NULL");
    if (num_layers != w->gen.nl) return fgnn_fail(FGNN_ERR_ARG, "num_layers must be 3 * num_mlp_layers (one activation / delta pair per Dense layer)");
    for (int li = 0; li < num_layers; ++li)
        if (!acts[li] || !deltas[li]) return fgnn_fail(FGNN_ERR_ARG, "buffer is NULL");
    if (w->device != g->device) return fgnn_fail(FGNN_ERR_ARG, "weights and graph live on different devices");
    FGNN_DEVICE_GUARD(g->device);
    GnnGenBwArgs a;
    std::memset(&a, 0, sizeof(a));
    a.B = B;
    a.llr = llr;
    a.logit_hx = logit_hx;
    a.logit_hz = logit_hz;
    a.synd_x = synd_x;
    a.synd_z = synd_z;
    a.gout = grad_out;
    for (int li = 0; li < num_layers; ++li) {
        a.acts[li] = acts[li];
        a.deltas[li] = deltas[li];
    }
    const size_t lds_bytes = (size_t)g->d.m * sizeof(float);
    if (lds_bytes > FGNN_LDS_BUDGET) return fgnn_fail(FGNN_ERR_ARG, "too many checks for the LDS-resident kernel");
    auto kern = gnn_general_backward_kernel;
    if (lds_bytes > 48 * 1024)
        FGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), g->d, w->gen, a);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}
