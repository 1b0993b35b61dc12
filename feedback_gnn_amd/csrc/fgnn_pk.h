// fgnn_pk.h — packed-f32 building blocks of the streaming kernels (gnn_stream_kernel, gnn_bp4_stream_kernel).
//
// A wave-uniform weight multiplies a per-lane value: as the SGPR operand of v_fmac_f32 that issues at half rate in bursts on gfx950;
// as an SGPR PAIR of v_pk_fma_f32 — two output units per instruction, the per-lane input broadcast to both halves by op_sel — the
// instruction issues in about the time of one such v_fmac (tools/microbench, round 3: 1.85-1.93 ns per wave-instruction and SIMD against
// 1.79), i.e. uniform-weight multiply-adds at the full f32 rate.  Each half is an IEEE fma: the bits of the scalar instruction.
#ifndef FGNN_PK_H
#define FGNN_PK_H

#include "fgnn_math.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const f2 __attribute__((address_space(4)))* scalar_f2p;
__device__ __forceinline__ scalar_f2p as_scalar2(const float* p) { return (scalar_f2p)(unsigned long long)p; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bc2(float x) { return f2{x, x}; }

// acc[jp] = fmaf(in[k], W[k][j0 + 2 jp + {0, 1}], acc[jp]) for k = 0 .. K-1 in this order; the K inputs arrive as (K + 1) / 2 pairs of
// consecutive elements (element k = half k & 1 of pair k >> 1, picked by the instruction's op_sel: no broadcast copies); w = &W[0][j0],
// row stride in floats.  The weight rows arrive through the scalar cache in groups of KG rows, one group ahead: scalar loads return
// out of order, so the only wait there is is "all of them" (s_waitcnt lgkmcnt(0)); the group in use is therefore waited for FIRST (the
// empty asm that names it), then the next group's loads are issued, then the group's packed fmas run — and the scheduler may not move
// anything across the group boundaries (left alone it hoists every s_load of a layer to the top and spills hundreds of SGPRs).  The
// input pairs pass through an empty asm at the top: a loop-invariant input would otherwise have its {x, x} broadcasts hoisted out of
// the caller's block loop as twice as many live registers.
// (LAUNDER = false: the caller has passed its pairs through such an asm itself, in place — no copies at all.)
template <int K, int JP, int KG, bool LAUNDER = true>
__device__ __forceinline__ void dense_pk(const f2 (&in2)[(K + 1) / 2], const float* w, int stride, f2 (&acc)[JP])
{
    constexpr int NG = (K + KG - 1) / KG, KP = (K + 1) / 2;
    f2 x[KP];
#pragma unroll
    for (int q = 0; q < KP; ++q) {
        x[q] = in2[q];
        if constexpr (LAUNDER) asm volatile("" : "+v"(x[q]));
    }
    f2 buf[2][KG][JP];
#pragma unroll
    for (int kk = 0; kk < KG; ++kk)
        if (kk < K) {
            scalar_f2p r = as_scalar2(w + kk * stride);
#pragma unroll
            for (int jp = 0; jp < JP; ++jp) buf[0][kk][jp] = r[jp];
        }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        asm volatile("" ::"s"(buf[g & 1][0][0]));  // the group's rows are in their SGPRs from here on
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) {
#pragma unroll
            for (int kk = 0; kk < KG; ++kk)
                if ((g + 1) * KG + kk < K) {
                    scalar_f2p r = as_scalar2(w + ((g + 1) * KG + kk) * stride);
#pragma unroll
                    for (int jp = 0; jp < JP; ++jp) buf[(g + 1) & 1][kk][jp] = r[jp];
                }
        }
#pragma unroll
        for (int kk = 0; kk < KG; ++kk)
            if (g * KG + kk < K) {
                const int k = g * KG + kk;
                const f2 pr = x[k >> 1];
                const f2 xb = (k & 1) ? __builtin_shufflevector(pr, pr, 1, 1) : __builtin_shufflevector(pr, pr, 0, 0);
#pragma unroll
                for (int jp = 0; jp < JP; ++jp) acc[jp] = pk_fma(xb, buf[g & 1][kk][jp], acc[jp]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ f2 tanh2(f2 a) { return f2{fg_tanh(a.x), fg_tanh(a.y)}; }
// {x, x} from a lone register without a copy: the instruction's op_sel reads the low half twice (the high half is never read)
__device__ __forceinline__ f2 bc_lo(float x)
{
    f2 t;
    t.x = x;
    t.y = x;
    return __builtin_shufflevector(t, t, 0, 0);
}

#endif  // FGNN_PK_H
