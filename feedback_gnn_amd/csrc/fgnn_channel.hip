// fgnn_channel.hip — the integer/byte stages around the decoder: depolarizing noise, syndromes,
// the per-round flag/merge logic and the residual check of
// Sandwich_BP_GNN_Evaluation_Model.call (/root/reference sionna/fec/ldpc/feedback_gnn.py:293-361),
// plus the block-error counting of sim_ber (sionna/utils/misc.py:647-669).
//
// The reference does all GF(2) products as dense int64 tf.matmul followed by `& 1`
// (feedback_gnn.py:308-309, 324-325, 349-353).  Here they are XORs over CSR rows of uint8 vectors
// held in LDS; every kernel is one workgroup per codeword-group with coalesced byte I/O.
#include "fgnn_internal.h"
#include "fgnn_math.h"
#include "fgnn_rng.h"

namespace {

// Pauli.call, sionna/channel/pauli.py:98-108.  One thread = 4 qubits of one sample (one Philox block).
// `base` (optional, device): the stream position the launch starts from is *base + first — a hipGraph that replays a Monte-Carlo loop
// keeps its position in device memory and advances it itself (fgnn_pauli_noise_dev), so every replay draws fresh samples.
// `thr_in` (optional): the thresholds of a general (px, py, pz) channel, formed on the host by fg_pauli_thresholds_xyz (additions and
// subtractions only); without it the kernel forms the depolarizing split of `p` itself, as the oracle does.
template <bool XYZ>
__global__ void __launch_bounds__(256) pauli_kernel(uint64_t seed, float p, fg_pauli_thr thr_in, uint64_t first,
                                                    const unsigned long long* __restrict__ base, int B, int n, int nblk,
                                                    uint8_t* __restrict__ ex, uint8_t* __restrict__ ez)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * nblk) return;
    if (base) first += (uint64_t)*base;
    const int b = (int)(t / nblk), blk = (int)(t - (long long)b * nblk);
    const fg_pauli_thr thr = XYZ ? thr_in : fg_pauli_thresholds(p);
    float u[4];
    fg_uniform4(seed, first + (uint64_t)b, (uint32_t)blk, u);
    const int q0 = blk * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q0 + k < n) {
            ex[(size_t)b * n + q0 + k] = fg_pauli_x(u[k], thr);
            ez[(size_t)b * n + q0 + k] = fg_pauli_z(u[k], thr);
        }
}

// Pauli.call with wt=True (pauli.py:80-97).  One wave per sample: its lanes zero the two rows and fill the identity
// permutation in LDS, lane 0 runs the `wt` Fisher-Yates steps (tens of dependent LDS accesses — noise generation is
// < 0.1 % of a Monte-Carlo step) and marks the chosen qubits.
__global__ void __launch_bounds__(256) pauli_wt_kernel(uint64_t seed, int wt, uint64_t first, int B, int n, uint8_t* __restrict__ ex,
                                                       uint8_t* __restrict__ ez)
{
    extern __shared__ unsigned short perm_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    unsigned short* perm = perm_all + (size_t)wave * n;
    uint8_t* rx = ex + (size_t)b * n;
    uint8_t* rz = ez + (size_t)b * n;
    for (int v = lane; v < n; v += 64) {
        perm[v] = (unsigned short)v;
        rx[v] = 0;
        rz[v] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane != 0) return;
    float up[4] = {0, 0, 0, 0}, ut[4] = {0, 0, 0, 0};
    for (int i = 0; i < wt; ++i) {
        if ((i & 3) == 0) {
            fg_uniform4s(seed, first + (uint64_t)b, (uint32_t)(i >> 2), 1u, up);
            fg_uniform4s(seed, first + (uint64_t)b, (uint32_t)(i >> 2), 2u, ut);
        }
        const int j = i + fg_fy_pick(up[i & 3], n - i);
        const unsigned short t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
        const float u = ut[i & 3];
        rx[perm[i]] = (uint8_t)(u < (2.0f / 3.0f));
        rz[perm[i]] = (uint8_t)(u > (1.0f / 3.0f));
    }
}

__device__ __forceinline__ unsigned row_parity(const uint8_t* x, const int* __restrict__ col, int p0, int p1)
{
    unsigned a = 0;
    for (int j = p0; j < p1; ++j) a ^= x[col[j]];
    return a & 1u;
}

// syndrome_x = hx noise_z ; syndrome_z = hz noise_x  (feedback_gnn.py:305-309)
__global__ void __launch_bounds__(1024) syndrome_kernel(GraphDev g, int B, int tpc, int cpb, const uint8_t* __restrict__ ex,
                                                        const uint8_t* __restrict__ ez, uint8_t* __restrict__ sx,
                                                        uint8_t* __restrict__ sz)
{
    extern __shared__ uint8_t sm[];
    const int cwl = threadIdx.x / tpc, lane = threadIdx.x - cwl * tpc, b = blockIdx.x * cpb + cwl;
    const bool active = b < B;
    const int n = g.n;
    uint8_t* lx = sm + (size_t)cwl * 2 * n;
    uint8_t* lz = lx + n;
    if (active)
        for (int v = lane; v < n; v += tpc) {
            lx[v] = ex[(size_t)b * n + v];
            lz[v] = ez[(size_t)b * n + v];
        }
    __syncthreads();
    if (!active) return;
    for (int c = lane; c < g.m; c += tpc) {
        const int p0 = g.cptr[c], p1 = g.cptr[c + 1];
        if (c < g.m_x) sx[(size_t)b * g.m_x + c] = (uint8_t)row_parity(lz, g.cvn, p0, p1);
        else sz[(size_t)b * g.m_z + (c - g.m_x)] = (uint8_t)row_parity(lx, g.cvn, p0, p1);
    }
}

// errors &= any([hz x_hat ; hx z_hat] != [synd_z ; synd_x])   (feedback_gnn.py:324-330)
__global__ void __launch_bounds__(1024) flag_kernel(GraphDev g, int B, int tpc, int cpb, const uint8_t* __restrict__ xh,
                                                    const uint8_t* __restrict__ zh, const uint8_t* __restrict__ sx,
                                                    const uint8_t* __restrict__ sz, uint8_t* __restrict__ errors,
                                                    const int* __restrict__ index)
{
    extern __shared__ uint8_t sm[];
    const int cwl = threadIdx.x / tpc, lane = threadIdx.x - cwl * tpc, slot = blockIdx.x * cpb + cwl;
    const bool active = slot < B;
    const int b = (active && index) ? index[slot] : slot;  // index: only the listed samples (compacted rounds)
    const int n = g.n;
    unsigned* neq = reinterpret_cast<unsigned*>(sm);  // [cpb]
    uint8_t* lx = sm + ((cpb * sizeof(unsigned) + 15) & ~size_t(15)) + (size_t)cwl * 2 * n;
    uint8_t* lz = lx + n;
    if (lane == 0) neq[cwl] = 0;
    if (active)
        for (int v = lane; v < n; v += tpc) {
            lx[v] = xh[(size_t)b * n + v];
            lz[v] = zh[(size_t)b * n + v];
        }
    __syncthreads();
    unsigned mine = 0;
    if (active)
        for (int c = lane; c < g.m; c += tpc) {
            const int p0 = g.cptr[c], p1 = g.cptr[c + 1];
            if (c < g.m_x) mine |= row_parity(lz, g.cvn, p0, p1) ^ (sx[(size_t)b * g.m_x + c] & 1u);
            else mine |= row_parity(lx, g.cvn, p0, p1) ^ (sz[(size_t)b * g.m_z + (c - g.m_x)] & 1u);
        }
    if (mine) atomicOr(&neq[cwl], 1u);
    __syncthreads();
    if (active && lane == 0) errors[b] = (uint8_t)((errors[b] != 0) && (neq[cwl] != 0));
}

// masked overwrite of the estimates (feedback_gnn.py:339-340)
__global__ void __launch_bounds__(256) merge_kernel(const uint8_t* __restrict__ errors, const uint8_t* __restrict__ xu,
                                                    const uint8_t* __restrict__ zu, long long total, int n,
                                                    uint8_t* __restrict__ xh, uint8_t* __restrict__ zh,
                                                    const int* __restrict__ index)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (index) {  // only the listed samples (compacted rounds): slot -> sample
        const long long slot = i / n;
        i = (long long)index[slot] * n + (i - slot * n);
    }
    if (errors[i / n]) {
        xh[i] = xu[i];
        zh[i] = zu[i];
    }
}

// residual check (feedback_gnn.py:343-361) + row-wise any() (metrics.py:221-223)
__global__ void __launch_bounds__(1024) residual_kernel(GraphDev g, int RX, int RZ, int B, int tpc, int cpb, const uint8_t* __restrict__ ex,
                                                        const uint8_t* __restrict__ ez, const uint8_t* __restrict__ xh,
                                                        const uint8_t* __restrict__ zh, uint8_t* __restrict__ s_hat,
                                                        uint8_t* __restrict__ ls_hat, uint8_t* __restrict__ flags)
{
    extern __shared__ uint8_t sm[];
    const int cwl = threadIdx.x / tpc, lane = threadIdx.x - cwl * tpc, b = blockIdx.x * cpb + cwl;
    const bool active = b < B;
    const int n = g.n;
    unsigned* fl = reinterpret_cast<unsigned*>(sm);  // [cpb]
    uint8_t* xd = sm + ((cpb * sizeof(unsigned) + 15) & ~size_t(15)) + (size_t)cwl * 2 * n;
    uint8_t* zd = xd + n;
    if (lane == 0) fl[cwl] = 0;
    __syncthreads();
    unsigned any_diff = 0;
    if (active)
        for (int v = lane; v < n; v += tpc) {
            xd[v] = ex[(size_t)b * n + v] ^ xh[(size_t)b * n + v];  // (:346)
            zd[v] = ez[(size_t)b * n + v] ^ zh[(size_t)b * n + v];  // (:347)
            any_diff |= xd[v] | zd[v];
        }
    // bit 2 of the flag word: the decision differs from the error somewhere.  When it does not (the usual case below
    // threshold) every parity below is a parity of zeros: the row loops are skipped unless the arrays were asked for.
    if (any_diff) atomicOr(&fl[cwl], 4u);
    __syncthreads();
    unsigned mine = 0;
    if (active && ((fl[cwl] & 4u) || s_hat || ls_hat)) {
        const int ms = g.m_z + g.m_x;
        // s_hat = [hz xd ; hx zd]  (:349-350,:355): combined check c: hx rows first, so remap
        for (int c = lane; c < g.m; c += tpc) {
            const int p0 = g.cptr[c], p1 = g.cptr[c + 1];
            unsigned bit;
            int pos;
            if (c < g.m_x) { bit = row_parity(zd, g.cvn, p0, p1); pos = g.m_z + c; }
            else { bit = row_parity(xd, g.cvn, p0, p1); pos = c - g.m_x; }
            if (s_hat) s_hat[(size_t)b * ms + pos] = (uint8_t)bit;
            mine |= bit;
        }
        // ls_hat = [hx_perp xd ; hz_perp zd]  (:352-353,:356)
        // RX / RZ = the row sets applied to xd / zd: (hx_perp, hz_perp) for the sandwich model, (lz, lx) for BP4_OSD_Model
        const int r0 = g.rows[RX], r1 = g.rows[RZ];
        for (int r = lane; r < r0 + r1; r += tpc) {
            unsigned bit;
            if (r < r0) bit = row_parity(xd, g.rcol[RX], g.rptr[RX][r], g.rptr[RX][r + 1]);
            else bit = row_parity(zd, g.rcol[RZ], g.rptr[RZ][r - r0], g.rptr[RZ][r - r0 + 1]);
            if (ls_hat) ls_hat[(size_t)b * (r0 + r1) + r] = (uint8_t)bit;
            mine |= bit << 1;
        }
    }
    if (mine) atomicOr(&fl[cwl], mine);
    __syncthreads();
    if (active && lane == 0 && flags) flags[b] = (uint8_t)(fl[cwl] & 3u);
}

__global__ void __launch_bounds__(256) count_kernel(const uint8_t* __restrict__ flags, int B, unsigned long long* counts)
{
    __shared__ unsigned sh[2];
    if (threadIdx.x < 2) sh[threadIdx.x] = 0;
    __syncthreads();
    unsigned f = 0, l = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        f += flags[i] & 1u;
        l += (flags[i] >> 1) & 1u;
    }
    if (f) atomicAdd(&sh[0], f);
    if (l) atomicAdd(&sh[1], l);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sh[0]) atomicAdd(&counts[0], (unsigned long long)sh[0]);
        if (sh[1]) atomicAdd(&counts[1], (unsigned long long)sh[1]);
        if (blockIdx.x == 0) atomicAdd(&counts[2], (unsigned long long)B);
    }
}

// Counters of num_batches consecutive Monte-Carlo batches decoded as ONE launch (sim_ber's fused read-back, utils.py): block j sums
// the flags of batch j into part[j]; one thread then walks the batches in order: ring[j] = the cumulative counters after batch j,
// i.e. exactly what num_batches sequential fgnn_count_flags calls would have left behind one after the other.
__global__ void __launch_bounds__(256) count_batches_kernel(const uint8_t* __restrict__ flags, int batch, unsigned* __restrict__ part)
{
    __shared__ unsigned sh[2];
    if (threadIdx.x < 2) sh[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* f0 = flags + (size_t)blockIdx.x * batch;
    unsigned f = 0, l = 0;
    for (int i = threadIdx.x; i < batch; i += blockDim.x) {
        f += f0[i] & 1u;
        l += (f0[i] >> 1) & 1u;
    }
    if (f) atomicAdd(&sh[0], f);
    if (l) atomicAdd(&sh[1], l);
    __syncthreads();
    if (threadIdx.x < 2) part[2 * blockIdx.x + threadIdx.x] = sh[threadIdx.x];
}

__global__ void count_scan_kernel(const unsigned* __restrict__ part, int num_batches, int batch, unsigned long long* counts,
                                  unsigned long long* ring)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long c0 = counts[0], c1 = counts[1], c2 = counts[2];
    for (int j = 0; j < num_batches; ++j) {
        c0 += part[2 * j];
        c1 += part[2 * j + 1];
        c2 += (unsigned long long)batch;
        ring[3 * j] = c0;
        ring[3 * j + 1] = c1;
        ring[3 * j + 2] = c2;
    }
    counts[0] = c0;
    counts[1] = c1;
    counts[2] = c2;
}

// Bit-packed decisions for the all-gather of SURVEY §8(e): row b = the 2n bits [x_hat[b,:] | z_hat[b,:]], most significant bit
// first inside a byte (numpy.packbits order), ceil(2n/8) bytes per codeword.  One thread per output byte; the 8 input bytes of
// a thread are contiguous except across the x|z seam, a wave reads 512 consecutive input bytes and writes 64 consecutive bytes.
__global__ void __launch_bounds__(256) pack_kernel(const uint8_t* __restrict__ x_hat, const uint8_t* __restrict__ z_hat,
                                                   long long total, int n, int nb, uint8_t* __restrict__ packed)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long b = i / nb;
    const int byte = (int)(i - b * nb);
    const uint8_t* xr = x_hat + b * n;
    const uint8_t* zr = z_hat + b * n;
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int pos = byte * 8 + k;
        unsigned bit = 0;
        if (pos < n) bit = xr[pos] & 1u;
        else if (pos < 2 * n) bit = zr[pos - n] & 1u;
        v |= bit << (7 - k);
    }
    packed[i] = (uint8_t)v;
}

__global__ void __launch_bounds__(256) unpack_kernel(const uint8_t* __restrict__ packed, long long total, int n, int nb,
                                                     uint8_t* __restrict__ x_hat, uint8_t* __restrict__ z_hat)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B * 2n decisions
    if (i >= total) return;
    const long long b = i / (2 * n);
    const int pos = (int)(i - b * 2 * n);
    const uint8_t v = (uint8_t)((packed[b * nb + (pos >> 3)] >> (7 - (pos & 7))) & 1u);
    if (pos < n) x_hat[b * n + pos] = v;
    else z_hat[b * n + pos - n] = v;
}

}  // namespace

extern "C" int fgnn_pack_decisions(const uint8_t* x_hat, const uint8_t* z_hat, int B, int n, uint8_t* packed, void* stream)
{
    if (B < 0 || n <= 0) return fgnn_fail(FGNN_ERR_ARG, "bad pack arguments");
    if (B == 0) return FGNN_OK;  // an empty batch needs no buffers (an empty torch tensor's data pointer is NULL)
    if (!x_hat || !z_hat || !packed) return fgnn_fail(FGNN_ERR_ARG, "bad pack arguments");
    const int nb = (2 * n + 7) / 8;
    const long long total = (long long)B * nb;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x_hat,
                       z_hat, total, n, nb, packed);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_unpack_decisions(const uint8_t* packed, int B, int n, uint8_t* x_hat, uint8_t* z_hat, void* stream)
{
    if (B < 0 || n <= 0) return fgnn_fail(FGNN_ERR_ARG, "bad unpack arguments");
    if (B == 0) return FGNN_OK;
    if (!x_hat || !z_hat || !packed) return fgnn_fail(FGNN_ERR_ARG, "bad unpack arguments");
    const int nb = (2 * n + 7) / 8;
    const long long total = (long long)B * 2 * n;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), packed,
                       total, n, nb, x_hat, z_hat);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_pauli_noise(uint64_t seed, float p, uint64_t first_sample, int B, int n, uint8_t* noise_x,
                                uint8_t* noise_z, void* stream)
{
    if (B < 0 || n <= 0) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    if (B == 0) return FGNN_OK;
    if (!noise_x || !noise_z) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    const int nblk = (n + 3) / 4;
    const long long total = (long long)B * nblk;
    hipLaunchKernelGGL(pauli_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                       p, fg_pauli_thr{0.0f, 0.0f, 0.0f}, first_sample, static_cast<const unsigned long long*>(nullptr), B, n, nblk,
                       noise_x, noise_z);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_pauli_noise_xyz(uint64_t seed, float px, float py, float pz, uint64_t first_sample, int B, int n,
                                    uint8_t* noise_x, uint8_t* noise_z, void* stream)
{
    // pauli.py:98-108 compares u with px, px - py and (px + pz) - py for ANY triple and validates nothing; only NaNs are refused here
    // (X, Y and Z are the disjoint events of probability px - py, py, pz - py when 0 <= py <= min(px, pz) and px + pz - py <= 1)
    if (B < 0 || n <= 0 || !(px == px && py == py && pz == pz)) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    if (B == 0) return FGNN_OK;
    if (!noise_x || !noise_z) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    const int nblk = (n + 3) / 4;
    const long long total = (long long)B * nblk;
    hipLaunchKernelGGL(pauli_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                       0.0f, fg_pauli_thresholds_xyz(px, py, pz), first_sample, static_cast<const unsigned long long*>(nullptr), B, n,
                       nblk, noise_x, noise_z);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_pauli_noise_dev(uint64_t seed, float p, const uint64_t* first_sample_dev, uint64_t offset, int B, int n,
                                    uint8_t* noise_x, uint8_t* noise_z, void* stream)
{
    if (B < 0 || n <= 0 || !(p >= 0.0f && p <= 1.0f)) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    if (B == 0) return FGNN_OK;
    if (!noise_x || !noise_z || !first_sample_dev) return fgnn_fail(FGNN_ERR_ARG, "bad noise arguments");
    const int nblk = (n + 3) / 4;
    const long long total = (long long)B * nblk;
    hipLaunchKernelGGL(pauli_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                       p, fg_pauli_thr{0.0f, 0.0f, 0.0f}, offset, reinterpret_cast<const unsigned long long*>(first_sample_dev), B, n,
                       nblk, noise_x, noise_z);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_pauli_noise_wt(uint64_t seed, int wt, uint64_t first_sample, int B, int n, uint8_t* noise_x, uint8_t* noise_z,
                                   void* stream)
{
    if (B < 0 || n <= 0 || n > 65535 || wt < 0 || wt > n) return fgnn_fail(FGNN_ERR_ARG, "bad fixed-weight noise arguments");
    if (B == 0) return FGNN_OK;
    if (!noise_x || !noise_z) return fgnn_fail(FGNN_ERR_ARG, "bad fixed-weight noise arguments");
    hipLaunchKernelGGL(pauli_wt_kernel, dim3((B + 3) / 4), dim3(256), (size_t)4 * n * sizeof(unsigned short),
                       static_cast<hipStream_t>(stream), seed, wt, first_sample, B, n, noise_x, noise_z);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_syndrome(const fgnn_graph* g, const uint8_t* noise_x, const uint8_t* noise_z, int B, uint8_t* synd_x,
                             uint8_t* synd_z, void* stream)
{
    if (!g || B < 0) return fgnn_fail(FGNN_ERR_ARG, "bad syndrome arguments");
    if (B == 0) return FGNN_OK;
    if (!noise_x || !noise_z || !synd_x || !synd_z) return fgnn_fail(FGNN_ERR_ARG, "bad syndrome arguments");
    FGNN_DEVICE_GUARD(g->device);
    LaunchGeom L = fgnn_geom(g, B);
    size_t lds = (size_t)L.cpb * 2 * g->d.n;
    hipLaunchKernelGGL(syndrome_kernel, dim3(L.blocks), dim3(L.threads), lds, static_cast<hipStream_t>(stream), g->d, B, L.tpc,
                       L.cpb, noise_x, noise_z, synd_x, synd_z);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

int fgnn_flag_update_impl(const fgnn_graph* g, const uint8_t* x_hat, const uint8_t* z_hat, const uint8_t* synd_x,
                          const uint8_t* synd_z, int B, uint8_t* errors, const int* index, void* stream)
{
    if (!g || B < 0) return fgnn_fail(FGNN_ERR_ARG, "bad flag arguments");
    if (B == 0) return FGNN_OK;
    if (!x_hat || !z_hat || !synd_x || !synd_z || !errors) return fgnn_fail(FGNN_ERR_ARG, "bad flag arguments");
    FGNN_DEVICE_GUARD(g->device);
    LaunchGeom L = fgnn_geom(g, B);
    size_t lds = ((L.cpb * sizeof(unsigned) + 15) & ~size_t(15)) + (size_t)L.cpb * 2 * g->d.n;
    hipLaunchKernelGGL(flag_kernel, dim3(L.blocks), dim3(L.threads), lds, static_cast<hipStream_t>(stream), g->d, B, L.tpc, L.cpb,
                       x_hat, z_hat, synd_x, synd_z, errors, index);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_flag_update(const fgnn_graph* g, const uint8_t* x_hat, const uint8_t* z_hat, const uint8_t* synd_x,
                                const uint8_t* synd_z, int B, uint8_t* errors, void* stream)
{
    return fgnn_flag_update_impl(g, x_hat, z_hat, synd_x, synd_z, B, errors, nullptr, stream);
}

int fgnn_merge_impl(const uint8_t* errors, const uint8_t* x_upd, const uint8_t* z_upd, int B, int n, uint8_t* x_hat, uint8_t* z_hat,
                    const int* index, void* stream)
{
    if (B < 0 || n <= 0) return fgnn_fail(FGNN_ERR_ARG, "bad merge arguments");
    if (B == 0) return FGNN_OK;
    if (!errors || !x_upd || !z_upd || !x_hat || !z_hat) return fgnn_fail(FGNN_ERR_ARG, "bad merge arguments");
    const long long total = (long long)B * n;
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), errors,
                       x_upd, z_upd, total, n, x_hat, z_hat, index);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_merge(const uint8_t* errors, const uint8_t* x_upd, const uint8_t* z_upd, int B, int n, uint8_t* x_hat,
                          uint8_t* z_hat, void* stream)
{
    return fgnn_merge_impl(errors, x_upd, z_upd, B, n, x_hat, z_hat, nullptr, stream);
}

extern "C" int fgnn_residual_rows(const fgnn_graph* g, int rows_x, int rows_z, const uint8_t* noise_x, const uint8_t* noise_z,
                                  const uint8_t* x_hat, const uint8_t* z_hat, int B, uint8_t* s_hat, uint8_t* ls_hat, uint8_t* flags,
                                  void* stream)
{
    if (!g || B < 0) return fgnn_fail(FGNN_ERR_ARG, "bad residual arguments");
    if (rows_x < 0 || rows_x > 5 || rows_z < 0 || rows_z > 5 || !g->d.rptr[rows_x] || !g->d.rptr[rows_z])
        return fgnn_fail(FGNN_ERR_STATE, "row sets for the residual check not installed (fgnn_graph_set_rows)");
    if (B == 0) return FGNN_OK;
    if (!noise_x || !noise_z || !x_hat || !z_hat) return fgnn_fail(FGNN_ERR_ARG, "bad residual arguments");
    FGNN_DEVICE_GUARD(g->device);
    LaunchGeom L = fgnn_geom(g, B);
    size_t lds = ((L.cpb * sizeof(unsigned) + 15) & ~size_t(15)) + (size_t)L.cpb * 2 * g->d.n;
    hipLaunchKernelGGL(residual_kernel, dim3(L.blocks), dim3(L.threads), lds, static_cast<hipStream_t>(stream), g->d, rows_x, rows_z, B,
                       L.tpc, L.cpb, noise_x, noise_z, x_hat, z_hat, s_hat, ls_hat, flags);
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_residual(const fgnn_graph* g, const uint8_t* noise_x, const uint8_t* noise_z, const uint8_t* x_hat,
                             const uint8_t* z_hat, int B, uint8_t* s_hat, uint8_t* ls_hat, uint8_t* flags, void* stream)
{
    return fgnn_residual_rows(g, FGNN_ROWS_HX_PERP, FGNN_ROWS_HZ_PERP, noise_x, noise_z, x_hat, z_hat, B, s_hat, ls_hat, flags, stream);
}

extern "C" int fgnn_count_flags_batches(const uint8_t* flags, int num_batches, int batch, uint64_t* counts, uint64_t* ring,
                                        uint32_t* scratch, void* stream)
{
    if (!flags || !counts || !ring || !scratch || num_batches < 0 || batch <= 0) return fgnn_fail(FGNN_ERR_ARG, "bad count arguments");
    if (num_batches == 0) return FGNN_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(count_batches_kernel, dim3(num_batches), dim3(256), 0, st, flags, batch, scratch);
    FGNN_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(count_scan_kernel, dim3(1), dim3(64), 0, st, scratch, num_batches, batch,
                       reinterpret_cast<unsigned long long*>(counts), reinterpret_cast<unsigned long long*>(ring));
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}

extern "C" int fgnn_count_flags(const uint8_t* flags, int B, uint64_t* counts, void* stream)
{
    if (!counts || B < 0) return fgnn_fail(FGNN_ERR_ARG, "bad count arguments");
    if (B == 0) return FGNN_OK;
    if (!flags) return fgnn_fail(FGNN_ERR_ARG, "bad count arguments");
    int blocks = (B + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), flags, B,
                       reinterpret_cast<unsigned long long*>(counts));
    FGNN_HIP_CHECK(hipGetLastError());
    return FGNN_OK;
}
