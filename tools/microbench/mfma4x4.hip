// v_mfma_f32_4x4x1_16b_f32 on gfx950: (1) operand / result lane map, checked with asymmetric integer data; (2) issue rate with independent
// accumulators and with ONE dependent accumulator; (3) that the accumulation is an fmaf chain.   hipcc --offload-arch=gfx950 -O3 mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(1024) rate(float* out, unsigned long long* cyc, float seed, int n)
{
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f4{seed, 0, 0, (float)j};
    const float a = seed * 1.0001f + threadIdx.x * 1e-7f, b = seed * 0.9999f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC>
void run_rate(int threads)
{
    const int blocks = 256, n = 4096 / NACC;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f, n);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f, n);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)n * NACC * (threads / 256.0);
    std::printf("4x4x1 f32, %d accumulators, %d wave(s)/SIMD: %.4f ms, %.2f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz); %.1f TFLOP/s chip\n", NACC,
                threads / 256, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, 512.0 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) {
        a[l] = (float)(1 + l);          // distinct per lane
        b[l] = (float)(1000 + 7 * l);
    }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l = 4*blk + j, register r:  D = A[lane 4*blk + r] * B[lane l]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float want = a[4 * (l / 4) + r] * b[l];
            if (d[l * 4 + r] != want) ++bad;
        }
    std::printf("layout hypothesis D[lane 4b+j][reg r] = A[lane 4b+r] * B[lane 4b+j]: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    if (bad) {
        for (int l = 0; l < 8; ++l) std::printf("lane %d: %g %g %g %g\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
    }
    for (int threads : {256, 512, 768, 1024}) {
        run_rate<1>(threads);
        run_rate<2>(threads);
        run_rate<5>(threads);
        run_rate<10>(threads);
    }
    return 0;
}
