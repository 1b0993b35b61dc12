// tests/math_bits_exhaustive.hip — the hipcc / gfx950 half of the exhaustive bit comparison of feedback_gnn_amd/csrc/fgnn_math.h and fgnn_rng.h.
//
// The oracle (gcc, x86) and the kernels (hipcc, gfx950) compile the SAME header, so "HIP == oracle" on decoder outputs cannot see
// a place where the two compilers, or the device paths of the header (FG_CLAMP = v_med3_f32, the rcp-based divisions, the LDS log
// table), turn it into different functions — unless BP happens to visit the input.  This program visits every input: for each
// function it evaluates the DEVICE build on every float32 bit pattern of a range and writes, per aligned window of 2^chunk_log2
// inputs, sum(result bits) and sum(result bits * (input bits | 1)) mod 2^64.  tests/test_gpu_math_bits.py forms the same sums
// with the gcc build (og_math_checksums in oracle/fgnn_oracle.c) and compares window by window.
//   usage: math_bits_exhaustive OUT.bin chunk_log2 fn:lo:hi [fn:lo:hi ...]      (lo, hi = uint32 bit patterns, hex or decimal)
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tests/math_bits_exhaustive.hip -o tests/_build/math_bits_exhaustive
// The compile flags are those of feedback_gnn_amd/csrc/Makefile (what the shipped kernels are built with).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../feedback_gnn_amd/csrc/fgnn_math.h"
#include "../feedback_gnn_amd/csrc/fgnn_rng.h"

// same numbering as og_math_fn (oracle/fgnn_oracle.c)
static __device__ __forceinline__ float apply(int fn, float v)
{
    switch (fn) {
    case 0: return fg_exp(v);
    case 1: return fg_log(v);
    case 2: return fg_log1p(v);
    case 3: return fg_softplus(v);
    case 4: return fg_phi(v);
    case 5: return fg_tanh(v);
    case 6: return fg_atanh(v);
    case 7: return fg_phi_gnn(v);
    case 8: return fg_lse2_corr(v, 0.0f);
    case 9: return fg_sigmoid(v);
    case 10: return fg_div3(v);
    case 11: return fg_rcp_unit(v);
    case 12: return fg_div_atanh(v);
    case 13: return fg_lse2(v, 1.0f);
    default: return 0.0f;
    }
}

// result bits of probe FN on input bits u; ids >= 14 are the integer-valued probes of fgnn_rng.h (same folding as og_math_bits)
static __device__ __forceinline__ uint32_t rotl(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
static __device__ __forceinline__ uint32_t apply_bits(int fn, uint32_t u)
{
    if (fn == 14) {
        uint32_t r[4];
        fg_philox4x32_10(u, ~u, u * 2654435761u, u >> 3, 0x5EEDu ^ (u << 5), u >> 7, r);
        return r[0] ^ rotl(r[1], 8) ^ rotl(r[2], 16) ^ rotl(r[3], 24);
    }
    if (fn == 15) return fg_f2u(fg_u32_to_unit(u));
    if (fn == 16) {
        const fg_pauli_thr t = fg_pauli_thresholds(fg_u2f(u));
        return fg_f2u(t.px) ^ rotl(fg_f2u(t.lo), 8) ^ rotl(fg_f2u(t.hi), 16);
    }
    return fg_f2u(apply(fn, fg_u2f(u)));
}

// Random123 known-answer vectors for philox4x32-10, evaluated on the device
__global__ void philox_kat(uint32_t* out)
{
    const uint32_t ctr[3][4] = {{0, 0, 0, 0}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u}};
    const uint32_t key[3][2] = {{0, 0}, {0xffffffffu, 0xffffffffu}, {0xa4093822u, 0x299f31d0u}};
    if (threadIdx.x < 3) {
        uint32_t r[4];
        const int i = threadIdx.x;
        fg_philox4x32_10(ctr[i][0], ctr[i][1], ctr[i][2], ctr[i][3], key[i][0], key[i][1], r);
        for (int j = 0; j < 4; ++j) out[4 * i + j] = r[j];
    }
}

constexpr int SPAN_LOG2 = 16;  // one workgroup walks 2^16 consecutive bit patterns, aligned: it lies inside one window (chunk_log2 >= 16)

template <int FN>
__global__ void __launch_bounds__(256) sums(uint32_t lo, uint32_t hi, int chunk_log2, unsigned long long* out)
{
    FG_LOG_TAB_SETUP();
    const uint64_t base = ((uint64_t)(lo >> SPAN_LOG2) + blockIdx.x) << SPAN_LOG2;
    unsigned long long s1 = 0, s2 = 0;
    for (int it = 0; it < (1 << SPAN_LOG2) / 256; ++it) {
        const uint64_t u = base + (uint64_t)it * 256 + threadIdx.x;
        if (u < lo || u > hi) continue;
        const uint32_t y = apply_bits(FN, (uint32_t)u);
        s1 += y;
        s2 += (unsigned long long)y * (unsigned long long)((uint32_t)u | 1u);
    }
    __shared__ unsigned long long red[2][256];
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint64_t k = (base >> chunk_log2) - (uint64_t)(lo >> chunk_log2);
        atomicAdd(&out[2 * k], red[0][0]);
        atomicAdd(&out[2 * k + 1], red[1][0]);
    }
}

typedef void (*kern_t)(uint32_t, uint32_t, int, unsigned long long*);
static const kern_t KERNELS[] = {sums<0>, sums<1>, sums<2>, sums<3>, sums<4>, sums<5>, sums<6>, sums<7>, sums<8>, sums<9>, sums<10>, sums<11>, sums<12>, sums<13>, sums<14>, sums<15>, sums<16>};

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s OUT.bin chunk_log2 fn:lo:hi [fn:lo:hi ...]\n", argv[0]);
        return 2;
    }
    const int cl = std::atoi(argv[2]);
    if (cl < SPAN_LOG2 || cl > 31) {
        std::fprintf(stderr, "chunk_log2 must be in [%d, 31]\n", SPAN_LOG2);
        return 2;
    }
    FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 2;
    {   // the device's Philox4x32-10 on the Random123 known-answer inputs, printed for the test to compare
        uint32_t* d_kat;
        uint32_t h_kat[12];
        if (hipMalloc(&d_kat, sizeof(h_kat)) != hipSuccess) return 3;
        hipLaunchKernelGGL(philox_kat, dim3(1), dim3(64), 0, 0, d_kat);
        if (hipMemcpy(h_kat, d_kat, sizeof(h_kat), hipMemcpyDeviceToHost) != hipSuccess) return 3;
        (void)hipFree(d_kat);
        for (int i = 0; i < 3; ++i) std::printf("philox_kat %d: %08x %08x %08x %08x\n", i, h_kat[4 * i], h_kat[4 * i + 1], h_kat[4 * i + 2], h_kat[4 * i + 3]);
    }
    for (int a = 3; a < argc; ++a) {
        unsigned fn;
        unsigned long long lo, hi;
        char* p = argv[a];
        fn = (unsigned)std::strtoul(p, &p, 0);
        if (*p++ != ':') return 2;
        lo = std::strtoull(p, &p, 0);
        if (*p++ != ':') return 2;
        hi = std::strtoull(p, &p, 0);
        if (fn >= sizeof(KERNELS) / sizeof(KERNELS[0]) || lo > hi || hi > 0xffffffffull) {
            std::fprintf(stderr, "bad range %s\n", argv[a]);
            return 2;
        }
        const size_t nwin = (size_t)((hi >> cl) - (lo >> cl) + 1);
        const unsigned nblk = (unsigned)((hi >> SPAN_LOG2) - (lo >> SPAN_LOG2) + 1);
        unsigned long long* d_out;
        if (hipMalloc(&d_out, nwin * 16) != hipSuccess || hipMemset(d_out, 0, nwin * 16) != hipSuccess) return 3;
        hipLaunchKernelGGL(KERNELS[fn], dim3(nblk), dim3(256), 0, 0, (uint32_t)lo, (uint32_t)hi, cl, d_out);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            std::fprintf(stderr, "kernel for %s failed\n", argv[a]);
            return 3;
        }
        std::vector<unsigned long long> h(nwin * 2);
        if (hipMemcpy(h.data(), d_out, nwin * 16, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        (void)hipFree(d_out);
        if (std::fwrite(h.data(), 16, nwin, f) != nwin) return 3;
        std::printf("fn %u bits 0x%08llx..0x%08llx: %llu inputs, %zu windows\n", fn, lo, hi, hi - lo + 1, nwin);
    }
    std::fclose(f);
    return 0;
}
