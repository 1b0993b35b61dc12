// fgnn_gnn.hip — the feedback GNN: one CN->VN message-passing layer that turns the marginals and
// soft syndromes of a BP run into the channel LLRs of the next one.
//
// Replaces Feedback_GNN.call of /root/reference sionna/fec/ldpc/feedback_gnn.py:161-188 (with
// reduce_msg :130-150 and MLP.call, gnn.py:63-69) for the configuration the reference trains and
// ships: num_msg_dims=20, num_hidden_units=40, num_mlp_layers=2, reduce_op="mean",
// activation="tanh", use_bias=True (n882.py:45-51).
//
// The reference materialises [bs,E,4], [bs,E,40] and [bs,E,20] tensors in HBM (27 GB at
// bs=65 536).  Here one thread owns one qubit: it walks the qubit's edges, runs the 4->40->20 edge
// MLP in registers, averages, and runs the 43->40->3 node MLP; the only HBM traffic is the
// [B,3,n] input/output and the two soft-syndrome vectors.  Weights are wave-uniform, so they
// arrive through scalar loads (SGPR operands of v_fmac_f32), transposed at upload so that each
// hidden unit's row is contiguous.
//
// Arithmetic order = oracle/fgnn_oracle.c (gnn_edge_side / gnn_one): every dot product is an fmaf
// chain in ascending k starting from 0, then + bias; edge messages are summed in ascending check
// order and divided by the degree.
#include <cstring>

#include "fgnn_internal.h"
#include "fgnn_math.h"

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
};

// vn_msg_mlp_{x,z} on one edge: feature [g, X, Y, Z] -> Dense(40,tanh) -> Dense(20)  (:175-181)
__device__ __forceinline__ void edge_mlp(float gv, float X, float Y, float Z, const float* __restrict__ w1t,
                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                         const float* __restrict__ b2, float (&msg)[MSG])
{
#pragma unroll
    for (int i = 0; i < MSG; ++i) msg[i] = 0.0f;
#pragma unroll 2
    for (int j = 0; j < HID; ++j) {
        const float* r = w1t + j * 4;
        float a = 0.0f;
        a = FG_FMA(gv, r[0], a);
        a = FG_FMA(X, r[1], a);
        a = FG_FMA(Y, r[2], a);
        a = FG_FMA(Z, r[3], a);
        const float h = fg_tanh(a + b1[j]);
        const float* r2 = w2 + j * MSG;
#pragma unroll
        for (int i = 0; i < MSG; ++i) msg[i] = FG_FMA(h, r2[i], msg[i]);
    }
#pragma unroll
    for (int i = 0; i < MSG; ++i) msg[i] = msg[i] + b2[i];
}

// mean over the qubit's edges of one side (:183-184, reduce_msg :139-141)
__device__ __forceinline__ void side_mean(const GraphDev& g, const int* __restrict__ vptr, int v, const float* gcn,
                                          float X, float Y, float Z, const float* w1t, const float* b1, const float* w2,
                                          const float* b2, float (&mean)[MSG])
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
            side_mean(g, g.vptr_x, v, gcn, X, Y, Z, w.w1t[0], w.b1[0], w.w2[0], w.b2[0], mean);
#pragma unroll
            for (int i = 0; i < MSG; ++i) feat[i] = mean[i];
            // hz slots start at E_x in vchk, check ids are side-local: g_z lives at gcn + m_x
            side_mean(g, g.vptr_z, v, gcn + g.m_x, X, Y, Z, w.w1t[1], w.b1[1], w.w2[1], w.b2[1], mean);
#pragma unroll
            for (int i = 0; i < MSG; ++i) feat[MSG + i] = mean[i];
        }
        // vn_embed_mlp Dense(40,tanh) on [m_x | m_z | X,Y,Z], then _llr_inv_embed Dense(3)  (:186)
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll 2
        for (int j = 0; j < HID; ++j) {
            const float* r = w.wet + j * 44;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 2 * MSG; ++k) acc = FG_FMA(feat[k], r[k], acc);
            acc = FG_FMA(X, r[40], acc);
            acc = FG_FMA(Y, r[41], acc);
            acc = FG_FMA(Z, r[42], acc);
            const float h = fg_tanh(acc + w.be[j]);
            const float* ro = w.wout + j * 4;
            o0 = FG_FMA(h, ro[0], o0);
            o1 = FG_FMA(h, ro[1], o1);
            o2 = FG_FMA(h, ro[2], o2);
        }
        out[v] = o0 + w.bout[0];
        out[n + v] = o1 + w.bout[1];
        out[2 * n + v] = o2 + w.bout[2];
    }
}

}  // namespace

extern "C" int fgnn_weights_create(const float* const host_arrays[12], int device, fgnn_weights** out)
{
    if (!host_arrays || !out) return fgnn_fail(FGNN_ERR_ARG, "NULL argument");
    for (int i = 0; i < 12; ++i)
        if (!host_arrays[i]) return fgnn_fail(FGNN_ERR_ARG, "weight array is NULL");
    FGNN_HIP_CHECK(hipSetDevice(device));
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
    *out = w;
    return FGNN_OK;
}

extern "C" void fgnn_weights_destroy(fgnn_weights* w)
{
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->blob) (void)hipFree(w->blob);
    delete w;
}

int fgnn_feedback_gnn_impl(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                           const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B, float* out,
                           const int* index, void* stream)
{
    if (!g || !w) return fgnn_fail(FGNN_ERR_ARG, "graph or weights is NULL");
    if (!llr || !logit_hx || !logit_hz || !synd_x || !synd_z || !out) return fgnn_fail(FGNN_ERR_ARG, "buffer is NULL");
    if (B < 0) return fgnn_fail(FGNN_ERR_ARG, "B must be >= 0");
    if (w->device != g->device) return fgnn_fail(FGNN_ERR_ARG, "weights and graph live on different devices");
    if (B == 0) return FGNN_OK;
    FGNN_HIP_CHECK(hipSetDevice(g->device));
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
    size_t lds_bytes = (size_t)a.lds_per_cw * sizeof(float) * (size_t)L.cpb;
    hipLaunchKernelGGL(gnn_kernel, dim3(L.blocks), dim3(L.threads), lds_bytes, static_cast<hipStream_t>(stream), g->d, w->d, a);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_feedback_gnn(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                                 const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B, float* out,
                                 void* stream)
{
    return fgnn_feedback_gnn_impl(g, w, llr, logit_hx, logit_hz, synd_x, synd_z, B, out, nullptr, stream);
}
