// Do f32 MFMAs of ONE wave overlap with f32 VALU work of ANOTHER wave on the same SIMD (gfx950)?  1024-thread workgroups, one per CU:
// 16 waves = 4 per SIMD.  role(w) = (w >> 2) & 1: waves 0-3 and 8-11 run MFMAs only, waves 4-7 and 12-15 run v_fma_f32 only, so every
// SIMD hosts two MFMA waves and two VALU waves.  Compared with the same MFMA waves alone and the same VALU waves alone (the other
// role idle).  If the two pipes were independent, "both" would take max(a, b); if they share the FP32 lanes, a + b.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096;

template <bool DO_MFMA, bool DO_VALU>
__global__ void __launch_bounds__(1024) k(float* out, float seed)
{
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = ((wave >> 2) & 1) == 0;
    float s = 0;
    if (mfma_role) {
        if (DO_MFMA) {
            f4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
            const float a = seed * 1.0001f, b = seed * 0.9999f;
            for (int it = 0; it < N / 4; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            }
            for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        }
    } else {
        if (DO_VALU) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
            const float a = seed * 1.0001f, b = seed * 0.9999f;
            for (int it = 0; it < N * 2; ++it) {  // 16 v_fma per MFMA of the other role: equal FP32-lane time if they share lanes
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
            }
            for (int i = 0; i < 8; ++i) s += v[i];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool M, bool V>
float run(float* out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<M, V>), dim3(256), dim3(1024), 0, 0, out, 1.0f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<M, V>), dim3(256), dim3(1024), 0, 0, out, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 1024);
    const float a = run<true, false>(out), b = run<false, true>(out), c = run<true, true>(out);
    std::printf("two MFMA-only waves per SIMD (%d x v_mfma_f32_16x16x4_f32 each):   %.4f ms\n", N, a);
    std::printf("two VALU-only waves per SIMD (%d x v_fma_f32 each):               %.4f ms\n", N * 16, b);
    std::printf("both kinds of waves together on every SIMD:                        %.4f ms   (max %.4f, sum %.4f)\n", c, a > b ? a : b, a + b);
    return 0;
}
