// Cost of an IEEE float division (the v_div_scale / v_rcp / fma / v_div_fmas / v_div_fixup sequence hipcc emits without fast-math)
// relative to v_fma_f32, four waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize div_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 4096;
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, float seed)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 0.001f;
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) v[i] = __builtin_fmaf(v[i], a, b);                 // 1 fma
            if (MODE == 1) v[i] = v[i] / (v[i] + 2.0f) + b;                   // add, IEEE div, add
            if (MODE == 2) v[i] = __builtin_amdgcn_rcpf(v[i] + 2.0f) * v[i] + b;  // add, v_rcp, mul, add
            if (MODE == 3) {                                                   // add + 10 dependent fmas: a division's worth of plain FMAs
                float d = v[i] + 2.0f;
#pragma unroll
                for (int q = 0; q < 10; ++q) d = __builtin_fmaf(d, a, b);
                v[i] = d + b;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name)
{
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per SIMD, each N*8 items
    std::printf("%-52s %.3f ms  = %.2f ns per item per SIMD\n", name, ms, ms * 1e6 / (4.0 * N * 8));
    (void)hipFree(out);
}
int main()
{
    run<0>("v_fma_f32");
    run<1>("add + IEEE division + add");
    run<2>("add + v_rcp_f32 + mul + add");
    run<3>("add + 10 dependent v_fma_f32 + add");
    return 0;
}
