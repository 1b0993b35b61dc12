// fgnn_internal.h — shared declarations of libfgnn_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/fgnn.h"

// Device view of one CSS code.  Messages of one codeword live in ONE array of E = E_x + E_z floats:
// slots [0,E_x) are the hx edges, [E_x,E) the hz edges, each block sorted by (qubit, check)
// ("VN-major"), so a variable node owns a contiguous run per side and a check node gathers.
struct GraphDev {
    int n, m_x, m_z, m, E_x, E_z, E;
    const int* vptr_x;  // [n+1] first hx slot of qubit v
    const int* vptr_z;  // [n+1] first hz slot of qubit v (already offset by E_x)
    const int* vchk;    // [E]   check id (0..m_x-1 / 0..m_z-1) of each VN-major slot
    const int* cptr;    // [m+1] combined checks: hx rows 0..m_x-1 then hz rows; offsets into cslot/cvn
    const int* cslot;   // [E]   message slot of the k-th edge of a check (ascending qubit)
    const int* cvn;     // [E]   qubit of that edge
    const uint16_t* cslot16;  // [m][8] packed rows of slot BYTE offsets (4 * slot) for DC-regular graphs with DC <= 8 and 4E < 65536, else null
    // CSR row sets (fgnn_graph_set_rows)
    int rows[6];
    const int* rptr[6];
    const int* rcol[6];
    // column-major view of the two soft-syndrome row sets (which = 0, 1): rows containing qubit v, ascending; used by
    // the backward kernel to gather row gradients per qubit in a fixed order
    const int* tptr[2];  // [n+1]
    const int* trow[2];  // [nnz]
    // uniform degrees, 0 if irregular
    int dvx, dvz, dc;
    int max_vdeg;  // largest number of edges (both sides) at one qubit: the fixed-point detector packs one sign bit per edge
    int max_cdeg, max_cdeg_x;  // largest check degree over all checks / over the hx checks (binary BP): the predicated register-resident
                               // check updates of the runtime-degree kernels are compiled for degrees up to 8 and 16
};

struct fgnn_graph {
    GraphDev d;
    int device;
    int tpc, cpb;  // threads per codeword, codewords per block
    bool user_launch;
    bool shortcut = true;        // exact wave-uniform shortcuts for saturated nodes (fgnn_graph_set_option)
    bool early_exit = true;      // exact fixed-point exit of the iteration loop (needs shortcut; fgnn_graph_set_option)
    bool hw_transcendentals = false;  // opt-in: phi-rule BP4 on v_exp_f32 / v_log_f32, fixed dataflow, NOT bit-exact (fgnn_graph_set_option 3)
    bool bp4_shared_lse = false; // opt-in (fgnn_graph_set_option 5): qubit update with the (a - b) part of the log-sum-exp once per qubit and side; default = one per edge, decoding_q.py:254-273 term by term
    int gnn_stream = 1;          // factored feedback GNN of a regular graph on the streaming VALU kernel (fgnn_graph_set_option 6): 0 never (MFMA tiles), 1 where it is the faster one, 2 always
    bool gnn_factored = false;   // opt-in (fgnn_graph_set_option 4): feedback GNN in the factored association (same function, 2/3 of the 40->20 layer gone); default = one Dense per edge, feedback_gnn.py:175-184 term by term
    bool force_generic = false;  // testing: run the runtime-degree kernel even on a regular graph
    std::vector<void*> allocs;
    // host copies of the canonical edge lists (fgnn_graph_edges)
    std::vector<int32_t> h_chk[2], h_var[2];
    void* row_alloc[6][4];
    void* basis_dev[2] = {nullptr, nullptr};  // pivot rows of hx / hz (fgnn_graph_set_basis, OSD)
    int basis_rank[2] = {0, 0};
    // optional per-launch timing of the BP4 kernel with HIP events on the launch stream (fgnn_profile_*)
    mutable bool prof_on = false;
    mutable int prof_n = 0;
    mutable std::vector<hipEvent_t> prof_ev;  // 2 per launch
    mutable std::vector<int> prof_iters, prof_batch;
};

// Device layout of one feedback GNN (transposed where that makes the scalar loads contiguous).
// Wave-uniform weights are read through the CONSTANT address space: uniform addresses there become s_load (scalar cache -> SGPR operands
// of v_fma); through a plain global pointer hipcc issues per-lane global_load_dwordx4 of the one address and parks the row in VGPRs.
#if defined(__HIPCC__)
typedef const float __attribute__((address_space(4)))* scalar_fp;
__device__ __forceinline__ scalar_fp as_scalar(const float* p) { return (scalar_fp)(unsigned long long)p; }
#endif

struct WeightsDev {
    const float* w1t[2];  // [40][4]  W1^T of vn_msg_mlp_{x,z}
    const float* b1[2];   // [40]
    const float* w2[2];   // [40][20]
    const float* b2[2];   // [20]
    const float* wet;     // [40][44] We^T (k padded 43 -> 44 with 0)
    const float* be;      // [40]
    const float* wout;    // [40][4]  (3 padded to 4)
    const float* bout;    // [4]
    const float* lane_tab;  // [132][64] per-lane MFMA operand / bias tables (fgnn_gnn.hip, mfma path)
    const float* msg_rows[2];  // [40][32] per hidden unit: W1[0..3][j], b1[j], 0,0,0, W2[j][0..19], 0 x 4 (gnn_stream_kernel)
    const float* emb_rows;     // [40][48] per hidden unit: We[0..42][j], be[j], Wout[j][0..2], 0
    const float* emb_quads;    // [10][192] per four hidden units j0..j0+3: We[k][j0..j0+3] for k = 0..42, be[j0..j0+3], Wout[j0+u][0..2] for u = 0..3, 0 x 4
};

// Runtime-shaped feedback GNN (fgnn_weights_create_general): Dense layers in execution order
//   [0, L) vn_msg_mlp_x | [L, 2L) vn_msg_mlp_z | [2L, 3L-1) vn_embed_mlp | 3L-1 _llr_inv_embed
constexpr int FGNN_GEN_MAX_LAYERS = 12;
constexpr int FGNN_GEN_MAX_D = 32, FGNN_GEN_MAX_W = 96;
struct GnnGeneralDev {
    int D, H, L, reduce_op, act, bias, nl;
    int K[FGNN_GEN_MAX_LAYERS], J[FGNN_GEN_MAX_LAYERS], act_l[FGNN_GEN_MAX_LAYERS];
    const float* W[FGNN_GEN_MAX_LAYERS];  // [K][J]
    const float* b[FGNN_GEN_MAX_LAYERS];  // [J] or null
};

struct fgnn_weights {
    WeightsDev d;
    int device;
    void* blob;
    bool general = false;
    GnnGeneralDev gen;
};

void fgnn_set_error(const std::string& s);
int fgnn_fail(int code, const std::string& s);
#define FGNN_HIP_CHECK(expr)                                                                          \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fgnn_fail(FGNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

// Every ABI entry point runs on its handle's device and leaves the caller's current device as it found it.
struct fgnn_device_guard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit fgnn_device_guard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else prev = -1;  // nothing to restore
    }
    ~fgnn_device_guard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    fgnn_device_guard(const fgnn_device_guard&) = delete;
    fgnn_device_guard& operator=(const fgnn_device_guard&) = delete;
};
#define FGNN_DEVICE_GUARD(device)                                                                     \
    fgnn_device_guard _dev_guard(device);                                                             \
    if (_dev_guard.err != hipSuccess)                                                                 \
        return fgnn_fail(FGNN_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_dev_guard.err))

// Optional per-launch timing (fgnn_profile_*): HIP events on the launch stream around one kernel launch.  `tag` is what
// fgnn_profile_read reports as `iters`: the BP4 iteration count, FGNN_PROF_TAG_GNN for a feedback-GNN launch, FGNN_PROF_TAG_GNNBP4
// for a GNN_BP4 launch.
constexpr int FGNN_PROF_TAG_GNN = -1;
constexpr int FGNN_PROF_TAG_GNNBP4 = -2;  // a GNN_BP4 launch (fgnn_gnnbp4_decode)
struct fgnn_prof_scope {
    const fgnn_graph* g;
    hipStream_t st;
    bool on;
    fgnn_prof_scope(const fgnn_graph* g_, hipStream_t st_) : g(g_), st(st_)
    {
        on = g->prof_on && (size_t)(2 * g->prof_n + 1) < g->prof_ev.size();
        if (on && hipEventRecord(g->prof_ev[2 * g->prof_n], st) != hipSuccess) on = false;
    }
    void done(int tag, int B)
    {
        if (!on) return;
        if (hipEventRecord(g->prof_ev[2 * g->prof_n + 1], st) != hipSuccess) return;
        g->prof_iters[g->prof_n] = tag;
        g->prof_batch[g->prof_n] = B;
        g->prof_n++;
    }
};

// dynamic LDS a kernel may ask for: the CU's 160 KB minus the 256-byte log table of fgnn_math.h (static LDS)
constexpr size_t FGNN_LDS_BUDGET = 160 * 1024 - 256;

// launch geometry shared by the per-codeword kernels
struct LaunchGeom {
    int tpc, cpb, threads, blocks;
};
LaunchGeom fgnn_geom(const fgnn_graph* g, int B);

// internal entry points with an optional slot->sample indirection (compacted sandwich rounds); bp4: flagged[b] (optional) = the
// decision's syndrome differs from the measured one — the sandwich's flag test fused into the decoder's epilogue
int fgnn_bp4_decode_impl(const fgnn_graph* g, int cn_type, int num_iter, float normalization_factor, const float* llr_ch,
                         float llr_const, const uint8_t* synd_x, const uint8_t* synd_z, int B, const float* msg_init_x,
                         const float* msg_init_z, float* llr_out, uint8_t* x_hat, uint8_t* z_hat, float* x_logit,
                         float* z_logit, float* msg_out_x, float* msg_out_z, const int* index, uint8_t* flagged, void* stream);
int fgnn_feedback_gnn_impl(const fgnn_graph* g, const fgnn_weights* w, const float* llr, const float* logit_hx,
                           const float* logit_hz, const uint8_t* synd_x, const uint8_t* synd_z, int B, float* out,
                           const int* index, void* stream);
// flag update / masked merge restricted to a list of samples (index[0..B), null = samples 0..B-1)
int fgnn_flag_update_impl(const fgnn_graph* g, const uint8_t* x_hat, const uint8_t* z_hat, const uint8_t* synd_x,
                          const uint8_t* synd_z, int B, uint8_t* errors, const int* index, void* stream);
int fgnn_merge_impl(const uint8_t* errors, const uint8_t* x_upd, const uint8_t* z_upd, int B, int n, uint8_t* x_hat, uint8_t* z_hat,
                    const int* index, void* stream);
