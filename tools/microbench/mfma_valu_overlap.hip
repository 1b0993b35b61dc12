// Does v_mfma_f32_16x16x4_f32 overlap with f32 VALU work on gfx950?  One wave per SIMD (256-thread workgroups, one per CU) and
// four waves per SIMD run (a) MFMAs only, (b) v_fma_f32 only, (c) both interleaved in every wave.  Prints shader cycles (s_memtime) per wave.   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 2048;  // MFMAs per wave; 8 VALU per MFMA in the mixed modes

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, float seed)
{
    const int wave = threadIdx.x >> 6;
    f4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    constexpr bool do_mfma = MODE == 0 || MODE == 2;
    constexpr bool do_valu = MODE == 1 || MODE == 2;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N / 4; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (do_mfma) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads)
{
    const int blocks = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * (threads / 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= h.size();
    std::printf("%-44s waves/SIMD=%d  %8.0f cycles/wave (s_memtime ticks)  %.3f ms  -> %.1f per MFMA-slot\n", name, threads / 256, mean, ms,
                mean / N);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int threads : {256, 1024}) {
        run<0>("MFMA f32 16x16x4 only (4 chains)", threads);
        run<1>("v_fma_f32 only (8 per slot, 8 chains)", threads);
        run<2>("both, interleaved in every wave", threads);
    }
    return 0;
}
