// tests/abi_host.cpp — the C ABI of include/fgnn.h driven from a plain C++ host program: no Python, no torch.
//
// What a non-Python caller of the drop-in boundary does: build the Tanner graphs of the [[882,24]] QC-GHP
// code from COO lists, install the stage-one soft-syndrome row sets, upload a feedback GNN, then per batch
//   fgnn_pauli_noise -> fgnn_syndrome -> fgnn_sandwich_decode (BP4-64, GNN, BP4-16) -> fgnn_flag_update
// on buffers it hipMalloc'ed itself, on its own stream.  Every output is compared bit for bit with the CPU oracle's C entry
// points (oracle/fgnn_oracle.c, test infrastructure) on the same Philox samples.
//
// Built by __graft_entry__.build() (hipcc, host code only) into tests/_build/abi_host; run by tests/test_abi.py (-m gpu).
//   usage: abi_host [batch] [p]      |      abi_host --graph   (no GPU: prints the edge lists' checksum for tests/test_abi.py)
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/fgnn.h"

extern "C" {  // the oracle's C API (declared here: the oracle ships no header, it is not a product interface)
struct og_graph;
og_graph* og_graph_create(int n, int m_x, int m_z, int E_x, const int32_t* chk_x, const int32_t* var_x, int E_z,
                          const int32_t* chk_z, const int32_t* var_z);
void og_graph_set_rows(og_graph* g, int which, int rows, int nnz, const int32_t* r, const int32_t* c);
void og_graph_set_gnn_order(og_graph* g, int factored);      // 0 (initial) = the reference's formulas term by term,
void og_graph_set_vn_shared_lse(og_graph* g, int shared);    // 1 = the library's opt-in re-association (fgnn_graph_set_option 4 / 5)
void og_graph_destroy(og_graph* g);
int og_pauli_noise(uint64_t seed, float p, uint64_t first_sample, int B, int n, uint8_t* noise_x, uint8_t* noise_z);
int og_syndrome(const og_graph* g, const uint8_t* ex, const uint8_t* ez, int B, uint8_t* synd_x, uint8_t* synd_z);
int og_bp4_decode(const og_graph* g, int cn_type, int num_iter, float factor, const float* llr_ch, float llr_const,
                  const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* msg_init_x, const float* msg_init_z,
                  float* llr_out, uint8_t* x_hat, uint8_t* z_hat, float* x_logit, float* z_logit, float* msg_out_x,
                  float* msg_out_z);
int og_sandwich_decode(const og_graph* g, int num_layers, const int* iters, const float* factors, const int* cn_types,
                       const float* const* const* weights, float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B,
                       uint8_t* x_hat, uint8_t* z_hat, float* llr_final, uint8_t* rounds);
}

#define HIP_OK(e)                                                                                  \
    do {                                                                                           \
        hipError_t _e = (e);                                                                       \
        if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } \
    } while (0)
#define FG_OK(e)                                                                                   \
    do {                                                                                           \
        int _r = (e);                                                                              \
        if (_r != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #e, _r, fgnn_last_error()); return 3; } \
    } while (0)

namespace {

struct Coo {
    std::vector<int32_t> r, c;
    void add(int row, int col) { r.push_back(row); c.push_back(col); }
};

// [[882,24]] = QC-GHP(l = 63, A = 7x7 array of circulant permutations with shifts {27,54,0} on wrapped diagonals,
// b = {0,1,6}): hx = [A | I (x) C], hz = [I (x) C^T | A^T] with P^s[(i+s)%l, i] = 1  (reference codes_q.py:84-89,208-234; n882.py:34).
void build_ghp882(Coo& hx, Coo& hz, int& n, int& m)
{
    const int l = 63, k = 7, shifts[3] = {27, 54, 0}, b[3] = {0, 1, 6};
    std::vector<int> a(k * k, -1);
    for (int i = 0; i < k; ++i)
        for (int t = 0; t < 3; ++t) a[i * k + (i - t + k) % k] = shifts[t];  // create_cyclic_permuting_matrix (:229-234): A[j, (j-i)%n] = s_i
    m = k * l;
    n = 2 * k * l;
    for (int br = 0; br < k; ++br)
        for (int bc = 0; bc < k; ++bc) {
            const int s = a[br * k + bc];
            if (s < 0) continue;
            for (int i = 0; i < l; ++i) {
                const int row = br * l + (i + s) % l, col = bc * l + i;  // block (br,bc) of A = P^s
                hx.add(row, col);
                hz.add(col, k * l + row);  // A^T occupies the right half of hz
            }
        }
    for (int blk = 0; blk < k; ++blk)
        for (int t = 0; t < 3; ++t)
            for (int i = 0; i < l; ++i) {
                const int row = blk * l + (i + b[t]) % l, col = blk * l + i;  // C = sum_c P^c
                hx.add(row, k * l + col);                                    // I (x) C, right half of hx
                hz.add(col, row);                                            // I (x) C^T, left half of hz
            }
}

template <typename T>
size_t mismatches(const std::vector<T>& a, const std::vector<T>& b)
{
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += std::memcmp(&a[i], &b[i], sizeof(T)) != 0;
    return bad;
}

}  // namespace

int main(int argc, char** argv)
{
    const bool graph_only = argc > 1 && std::strcmp(argv[1], "--graph") == 0;
    const int B = argc > 1 && !graph_only ? std::atoi(argv[1]) : 96;
    const float p = argc > 2 ? (float)std::atof(argv[2]) : 0.10f;
    const uint64_t seed = 0x5EED, first = 1234567;
    Coo hx, hz;
    int n = 0, m = 0;
    build_ghp882(hx, hz, n, m);
    const int E = (int)hx.r.size();
    if (n != 882 || m != 441 || E != 2646 || (int)hz.r.size() != 2646) { std::fprintf(stderr, "construction is off\n"); return 1; }
    if (graph_only) {  // sum over the edges of (row * 2654435761 + col) ^ side tag, order-independent
        uint64_t h = 0;
        for (int e = 0; e < E; ++e) {
            h += ((uint64_t)hx.r[e] * 2654435761ull + (uint64_t)hx.c[e]) ^ 0x1111ull;
            h += ((uint64_t)hz.r[e] * 2654435761ull + (uint64_t)hz.c[e]) ^ 0x2222ull;
        }
        std::printf("%d %d %d %llu\n", n, m, E, (unsigned long long)h);
        return 0;
    }
    std::vector<std::vector<int>> rowx(m), rowz(m);  // qubits of each check
    for (int e = 0; e < E; ++e) { rowx[hx.r[e]].push_back(hx.c[e]); rowz[hz.r[e]].push_back(hz.c[e]); }
    {   // CSS condition hx hz^T = 0 (mod 2)
        std::vector<uint8_t> Hz((size_t)m * n, 0);
        for (int e = 0; e < E; ++e) Hz[(size_t)hz.r[e] * n + hz.c[e]] ^= 1;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                int acc = 0;
                for (int v : rowx[i]) acc ^= Hz[(size_t)j * n + v];
                if (acc) { std::fprintf(stderr, "hx hz^T != 0 at (%d,%d)\n", i, j); return 1; }
            }
    }

    // a feedback GNN with pseudo-random parameters of the shipped shapes (reference file order, fgnn.h)
    const int shape[12] = {40 * 3, 3, 4 * 40, 40, 40 * 20, 20, 4 * 40, 40, 40 * 20, 20, 43 * 40, 40};
    std::vector<std::vector<float>> W(12);
    uint32_t lcg = 2463534242u;
    for (int i = 0; i < 12; ++i) {
        W[i].resize(shape[i]);
        for (float& x : W[i]) {
            lcg = lcg * 1664525u + 1013904223u;
            x = ((float)(lcg >> 8) / 16777216.0f - 0.5f) * ((i & 1) ? 0.2f : 0.7f);
        }
        if (i == 1) W[i] = {1.6f, 2.4f, 1.5f};  // output bias near the range of the trained networks
    }
    const float* wptr[12];
    for (int i = 0; i < 12; ++i) wptr[i] = W[i].data();

    const float p0 = 0.05f;
    const float L0 = std::log(3.0f * (1.0f - p0) / p0);  // feedback_gnn.py:311-312; 4.0430512
    const int iters[2] = {64, 16}, cn[2] = {FGNN_CN_BOXPLUS_PHI, FGNN_CN_BOXPLUS_PHI};
    const float factors[2] = {1.0f, 1.0f};

    // ---- oracle (host) ----
    og_graph* og = og_graph_create(n, m, m, E, hx.r.data(), hx.c.data(), E, hz.r.data(), hz.c.data());
    og_graph_set_rows(og, 0, m, E, hz.r.data(), hz.c.data());  // stage_one: pcm_x_perp = hz, pcm_z_perp = hx (decoding_q.py:35-37)
    og_graph_set_rows(og, 1, m, E, hx.r.data(), hx.c.data());
    // the checker restates the forms the library runs by default: the reference's formulas term by term
    // (FGNN_OPT_GNN_FACTORED = FGNN_OPT_BP4_SHARED_LSE = 0, the oracle's initial state — set explicitly)
    og_graph_set_gnn_order(og, 0);
    og_graph_set_vn_shared_lse(og, 0);
    std::vector<uint8_t> ex((size_t)B * n), ez((size_t)B * n), sx((size_t)B * m), sz((size_t)B * m), oxh((size_t)B * n),
        ozh((size_t)B * n), bxh((size_t)B * n), bzh((size_t)B * n);
    std::vector<float> ollr((size_t)B * 3 * n), bllr((size_t)B * 3 * n), bxl((size_t)B * m), bzl((size_t)B * m);
    og_pauli_noise(seed, p, first, B, n, ex.data(), ez.data());
    og_syndrome(og, ex.data(), ez.data(), B, sx.data(), sz.data());
    og_bp4_decode(og, FGNN_CN_BOXPLUS_PHI, 64, 1.0f, nullptr, L0, sx.data(), sz.data(), B, nullptr, nullptr, bllr.data(),
                  bxh.data(), bzh.data(), bxl.data(), bzl.data(), nullptr, nullptr);
    const float* const* wl[1] = {wptr};
    if (og_sandwich_decode(og, 2, iters, factors, cn, wl, L0, sx.data(), sz.data(), B, oxh.data(), ozh.data(), ollr.data(), nullptr)) {
        std::fprintf(stderr, "oracle sandwich failed\n");
        return 1;
    }

    // ---- device, through the C ABI only ----
    HIP_OK(hipSetDevice(0));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    fgnn_graph* g = nullptr;
    FG_OK(fgnn_graph_create(n, m, m, E, hx.r.data(), hx.c.data(), E, hz.r.data(), hz.c.data(), 0, &g));
    FG_OK(fgnn_graph_set_rows(g, FGNN_ROWS_X_LOGIT, m, E, hz.r.data(), hz.c.data()));
    FG_OK(fgnn_graph_set_rows(g, FGNN_ROWS_Z_LOGIT, m, E, hx.r.data(), hx.c.data()));
    fgnn_weights* gw = nullptr;
    FG_OK(fgnn_weights_create(wptr, 0, &gw));
    uint8_t *d_ex, *d_ez, *d_sx, *d_sz, *d_xh, *d_zh, *d_err;
    float *d_llr, *d_xl, *d_zl;
    void* d_ws;
    HIP_OK(hipMalloc((void**)&d_ex, (size_t)B * n));
    HIP_OK(hipMalloc((void**)&d_ez, (size_t)B * n));
    HIP_OK(hipMalloc((void**)&d_sx, (size_t)B * m));
    HIP_OK(hipMalloc((void**)&d_sz, (size_t)B * m));
    HIP_OK(hipMalloc((void**)&d_xh, (size_t)B * n));
    HIP_OK(hipMalloc((void**)&d_zh, (size_t)B * n));
    HIP_OK(hipMalloc((void**)&d_err, (size_t)B));
    HIP_OK(hipMalloc((void**)&d_llr, sizeof(float) * (size_t)B * 3 * n));
    HIP_OK(hipMalloc((void**)&d_xl, sizeof(float) * (size_t)B * m));
    HIP_OK(hipMalloc((void**)&d_zl, sizeof(float) * (size_t)B * m));
    const size_t ws_bytes = fgnn_sandwich_workspace_bytes(g, B);
    HIP_OK(hipMalloc(&d_ws, ws_bytes));

    FG_OK(fgnn_pauli_noise(seed, p, first, B, n, d_ex, d_ez, st));
    FG_OK(fgnn_syndrome(g, d_ex, d_ez, B, d_sx, d_sz, st));
    std::vector<uint8_t> gex(ex.size()), gez(ez.size()), gsx(sx.size()), gsz(sz.size()), gxh(oxh.size()), gzh(ozh.size());
    std::vector<float> gllr(ollr.size()), gxl(bxl.size()), gzl(bzl.size());
    size_t bad = 0;
    // (1) plain BP4-64: marginals, decisions, soft syndromes
    FG_OK(fgnn_bp4_decode(g, FGNN_CN_BOXPLUS_PHI, 64, 1.0f, nullptr, L0, d_sx, d_sz, B, nullptr, nullptr, d_llr, d_xh, d_zh, d_xl, d_zl,
                          nullptr, nullptr, st));
    HIP_OK(hipMemcpyAsync(gex.data(), d_ex, gex.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gez.data(), d_ez, gez.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gsx.data(), d_sx, gsx.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gsz.data(), d_sz, gsz.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gxh.data(), d_xh, gxh.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gzh.data(), d_zh, gzh.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gllr.data(), d_llr, gllr.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gxl.data(), d_xl, gxl.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(gzl.data(), d_zl, gzl.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    bad += mismatches(ex, gex) + mismatches(ez, gez) + mismatches(sx, gsx) + mismatches(sz, gsz);
    bad += mismatches(bxh, gxh) + mismatches(bzh, gzh) + mismatches(bllr, gllr) + mismatches(bxl, gxl) + mismatches(bzl, gzl);
    std::printf("BP4-64: %zu mismatching values (noise, syndromes, marginals, decisions, soft syndromes)\n", bad);
    // (2) the sandwich, with and without compaction of the feedback round
    for (int compact = 0; compact < 2; ++compact) {
        FG_OK(fgnn_sandwich_decode(g, 2, iters, factors, cn, &gw, L0, d_sx, d_sz, B, compact, d_xh, d_zh, compact ? nullptr : d_llr,
                                   nullptr, d_ws, ws_bytes, st));
        HIP_OK(hipMemcpyAsync(gxh.data(), d_xh, gxh.size(), hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(gzh.data(), d_zh, gzh.size(), hipMemcpyDeviceToHost, st));
        if (!compact) HIP_OK(hipMemcpyAsync(gllr.data(), d_llr, gllr.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        const size_t b2 = mismatches(oxh, gxh) + mismatches(ozh, gzh) + (compact ? 0 : mismatches(ollr, gllr));
        std::printf("sandwich BP4-64 + GNN + BP4-16 (compact=%d): %zu mismatching values\n", compact, b2);
        bad += b2;
    }
    // (3) still-flagged samples, device-side check of the final estimate against the syndrome
    HIP_OK(hipMemsetAsync(d_err, 1, (size_t)B, st));
    FG_OK(fgnn_flag_update(g, d_xh, d_zh, d_sx, d_sz, B, d_err, st));
    std::vector<uint8_t> err(B);
    HIP_OK(hipMemcpyAsync(err.data(), d_err, (size_t)B, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    int flagged = 0, flagged_ref = 0;
    for (int b = 0; b < B; ++b) {
        flagged += err[b] != 0;
        bool f = false;  // host recomputation: syndrome of the oracle's estimate vs the true syndrome
        for (int c = 0; c < m && !f; ++c) {
            int ax = 0, az = 0;
            for (int v : rowx[c]) ax ^= ozh[(size_t)b * n + v];
            for (int v : rowz[c]) az ^= oxh[(size_t)b * n + v];
            f = ax != sx[(size_t)b * m + c] || az != sz[(size_t)b * m + c];
        }
        flagged_ref += f;
    }
    std::printf("flagged after the sandwich: %d of %d (host recomputation: %d)\n", flagged, B, flagged_ref);
    bad += flagged != flagged_ref;
    // (3b) the stage_two / trainable trace in one launch: slot k of the trace = the soft syndromes of a k-iteration decode
    {
        const int T = 3;
        float *d_tx, *d_tz;
        HIP_OK(hipMalloc((void**)&d_tx, sizeof(float) * (size_t)(T + 1) * B * m));
        HIP_OK(hipMalloc((void**)&d_tz, sizeof(float) * (size_t)(T + 1) * B * m));
        FG_OK(fgnn_bp4_decode_trace(g, FGNN_CN_BOXPLUS_PHI, T, 1.0f, nullptr, L0, d_sx, d_sz, B, nullptr, nullptr, d_llr, d_xh, d_zh, d_tx,
                                    d_tz, nullptr, nullptr, st));
        std::vector<float> tx((size_t)(T + 1) * B * m), tz(tx.size()), rxl((size_t)B * m), rzl((size_t)B * m), rllr(ollr.size());
        std::vector<uint8_t> rxh(oxh.size()), rzh(ozh.size());
        HIP_OK(hipMemcpyAsync(tx.data(), d_tx, tx.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(tz.data(), d_tz, tz.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        size_t b3 = 0;
        for (int k = 0; k <= T; ++k) {
            og_bp4_decode(og, FGNN_CN_BOXPLUS_PHI, k, 1.0f, nullptr, L0, sx.data(), sz.data(), B, nullptr, nullptr, rllr.data(), rxh.data(),
                          rzh.data(), rxl.data(), rzl.data(), nullptr, nullptr);
            for (size_t i = 0; i < rxl.size(); ++i)
                b3 += std::memcmp(&rxl[i], &tx[(size_t)k * B * m + i], 4) != 0 || std::memcmp(&rzl[i], &tz[(size_t)k * B * m + i], 4) != 0;
        }
        std::printf("one-launch trace, %d iterations: %zu mismatching soft syndromes\n", T, b3);
        bad += b3;
        (void)hipFree(d_tx);
        (void)hipFree(d_tz);
    }
    // (3c) an empty batch needs no buffers: every entry point returns FGNN_OK for B = 0 with NULL data pointers
    {
        int rc = 0;
        rc |= fgnn_pauli_noise(seed, p, first, 0, n, nullptr, nullptr, st);
        rc |= fgnn_syndrome(g, nullptr, nullptr, 0, nullptr, nullptr, st);
        rc |= fgnn_bp4_decode(g, FGNN_CN_BOXPLUS_PHI, 4, 1.0f, nullptr, L0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, st);
        rc |= fgnn_bp4_decode_trace(g, FGNN_CN_BOXPLUS_PHI, 4, 1.0f, nullptr, L0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, st);
        rc |= fgnn_feedback_gnn(g, gw, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, st);
        rc |= fgnn_sandwich_decode(g, 2, iters, factors, cn, &gw, L0, nullptr, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, st);
        rc |= fgnn_flag_update(g, nullptr, nullptr, nullptr, nullptr, 0, nullptr, st);
        rc |= fgnn_merge(nullptr, nullptr, nullptr, 0, n, nullptr, nullptr, st);
        if (rc != 0) {
            std::printf("an empty batch with NULL buffers was rejected: %s\n", fgnn_last_error());
            ++bad;
        }
        if (fgnn_version() != 2) {
            std::printf("fgnn_version() = %d, this host was written against 2\n", fgnn_version());
            ++bad;
        }
    }
    // (4) errors cross the ABI as codes
    if (fgnn_bp4_decode(g, 7, 1, 1.0f, nullptr, L0, d_sx, d_sz, B, nullptr, nullptr, d_llr, d_xh, d_zh, nullptr, nullptr, nullptr,
                        nullptr, st) != FGNN_ERR_ARG || std::strstr(fgnn_last_error(), "Unknown node type") == nullptr) {
        std::printf("unknown cn_type was not rejected\n");
        ++bad;
    }
    fgnn_weights_destroy(gw);
    fgnn_graph_destroy(g);
    og_graph_destroy(og);
    for (void* ptr : {(void*)d_ex, (void*)d_ez, (void*)d_sx, (void*)d_sz, (void*)d_xh, (void*)d_zh, (void*)d_err, (void*)d_llr,
                      (void*)d_xl, (void*)d_zl, d_ws})
        (void)hipFree(ptr);
    (void)hipStreamDestroy(st);
    std::printf(bad ? "abi_host FAILED\n" : "abi_host ok: bit-identical to the oracle\n");
    return bad ? 1 : 0;
}
