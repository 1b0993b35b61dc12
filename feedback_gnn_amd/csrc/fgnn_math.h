/* fgnn_math.h — deterministic float32 elementary functions for the BP4 / feedback-GNN path.
 *
 * Every routine is built ONLY from IEEE-754 binary32 fma / add / mul / div, comparisons and
 * 32-bit integer operations.  No libm call, no hardware transcendental (v_exp_f32/v_log_f32 are
 * 1-ulp table approximations whose bits a CPU cannot reproduce).  The same header is compiled
 *   - by hipcc  (-ffp-contract=off) into the gfx950 kernels  (the .hip files next to this header), and
 *   - by gcc    (-ffp-contract=off -mfma) into the CPU oracle (the .c files under oracle),
 * so a GPU result and an oracle result are the same bits, for every sample and every iteration.
 * Accuracy (tools/check_math.c, exhaustive over the domains used): exp <= 0.9 ulp,
 * log <= 0.93 ulp, log1p <= 1.5 ulp, tanh <= 2.7 ulp — the same class as TensorFlow's own Eigen / XLA kernels,
 * which are not correctly rounded either (SURVEY.md §8c "Third-party arithmetic").
 *
 * The composite functions restate TensorFlow op semantics used by the reference:
 *   fg_softplus  : tf.math.softplus  (tf2xla Softplus: thresholds +-(log(eps)+2))
 *   fg_lse2      : tf.math.reduce_logsumexp over a stacked pair (max-shifted log-sum-exp)
 *   fg_phi       : QLDPCBPDecoder._phi, /root/reference sionna/fec/ldpc/decoding_q.py:365-373
 */
#ifndef FGNN_MATH_H
#define FGNN_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FG_FN static __device__ __host__ __forceinline__
#define FG_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define FG_MAX(a, b) __builtin_fmaxf((a), (b))
#define FG_MIN(a, b) __builtin_fminf((a), (b))
#define FG_ABS(a) __builtin_fabsf((a))
/* clamp to [lo, hi], lo <= hi, no NaNs: one v_med3_f32 (fmaxf/fminf would first canonicalise the operand: one extra VALU op) */
#if defined(__HIP_DEVICE_COMPILE__)
#define FG_CLAMP(x, lo, hi) __builtin_amdgcn_fmed3f((x), (lo), (hi))
#else
#define FG_CLAMP(x, lo, hi) FG_MIN(FG_MAX((x), (lo)), (hi))
#endif
FG_FN uint32_t fg_f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
FG_FN float fg_u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
#else
#define FG_FN static inline __attribute__((always_inline))
#define FG_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
/* no NaNs on this path and the sign of a zero never reaches a result, so the
 * ternary forms (maxss/minss) agree with v_max_f32/v_min_f32 numerically. */
#define FG_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define FG_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define FG_ABS(a) __builtin_fabsf((a))
#define FG_CLAMP(x, lo, hi) FG_MIN(FG_MAX((x), (lo)), (hi))
FG_FN uint32_t fg_f2u(float x) { union { float f; uint32_t u; } c; c.f = x; return c.u; }
FG_FN float fg_u2f(uint32_t x) { union { float f; uint32_t u; } c; c.u = x; return c.f; }
#endif

/* ---- constants -------------------------------------------------------------------------- */
#define FG_LOG2E 1.44269502e+00f        /* 0x3fb8aa3b */
#define FG_LN2_HI 6.93145752e-01f       /* 0x3f317200: 15 significant bits, k*LN2_HI exact for |k|<512 */
#define FG_LN2_LO 1.42860677e-06f       /* ln2 - LN2_HI */
#define FG_RND_MAGIC 12582912.0f        /* 1.5*2^23: (x + M) - M == rint(x) for |x| < 2^22 */
#define FG_SQRT_HALF_BITS 0x3f3504f3u   /* bits of sqrt(0.5) */
/* tf2xla Softplus threshold: log(FLT_EPSILON) + 2 in float32 */
#define FG_SOFTPLUS_THRESH 13.9423847f
/* decoding_q.py:372 clip constants ("optimized for tf.float32") */
#define FG_PHI_MIN 8.5e-8f
#define FG_PHI_MAX 16.635532f

/* ---- exp -------------------------------------------------------------------------------- */
/* e^x for x in [-87, 88].  k = rint(x*log2e); r = x - k*ln2 (two-term); degree-6 polynomial;
 * scale by adding k to the exponent field.  11 VALU ops on gfx950. */
FG_FN float fg_exp(float x)
{
    float t = FG_FMA(x, FG_LOG2E, FG_RND_MAGIC); /* low mantissa bits of t hold k */
    float k = t - FG_RND_MAGIC;
    float r = FG_FMA(k, -FG_LN2_HI, x);
    r = FG_FMA(k, -FG_LN2_LO, r);
    float p = 1.381461043e-03f;
    p = FG_FMA(p, r, 8.368710056e-03f);
    p = FG_FMA(p, r, 4.166838899e-02f);
    p = FG_FMA(p, r, 1.666652113e-01f);
    p = FG_FMA(p, r, 4.999999404e-01f);
    p = FG_FMA(p, r, 1.0f);
    p = FG_FMA(p, r, 1.0f);
    /* bits(t) = 0x4B400000 + k; shifting left by 23 leaves exactly k<<23 (mod 2^32) */
    return fg_u2f(fg_f2u(p) + (fg_f2u(t) << 23));
}

/* ---- log -------------------------------------------------------------------------------- */
/* log(2^e * (1+f)) for f in [sqrt(.5)-1, sqrt(2)-1], ef = (float)e.
 * log1p(f) = f - f^2/2 + f^3*P7(f).  12 VALU ops. */
FG_FN float fg_log_core(float f, float ef)
{
    float z = f * f;
    float p = -7.634429634e-02f;
    p = FG_FMA(p, f, 1.276154965e-01f);
    p = FG_FMA(p, f, -1.316019446e-01f);
    p = FG_FMA(p, f, 1.420176178e-01f);
    p = FG_FMA(p, f, -1.662335694e-01f);
    p = FG_FMA(p, f, 2.000122666e-01f);
    p = FG_FMA(p, f, -2.500082254e-01f);
    p = FG_FMA(p, f, 3.333333135e-01f);
    float u = FG_FMA(f, p, -0.5f);
    float r = FG_FMA(z, u, f);
    r = FG_FMA(ef, FG_LN2_LO, r);
    return FG_FMA(ef, FG_LN2_HI, r);
}

/* log(x) for normal positive x. */
FG_FN float fg_log(float x)
{
    uint32_t ix = fg_f2u(x) - FG_SQRT_HALF_BITS;
    int32_t e = (int32_t)ix >> 23;
    float m = fg_u2f((ix & 0x007fffffu) + FG_SQRT_HALF_BITS); /* m in [sqrt(.5), sqrt(2)) */
    return fg_log_core(m - 1.0f, (float)e);
}

/* log(w) for w in [1, 2]: same bits as fg_log(w) (same reduced argument, same core), cheaper split. */
FG_FN float fg_log_1to2(float w)
{
    int big = w >= 1.41421354f; /* bits 0x3fb504f3 = 2*sqrt(.5): fg_log switches exponent here too */
    float m = big ? 0.5f * w : w;
    return fg_log_core(m - 1.0f, big ? 1.0f : 0.0f);
}

/* log(1+u) for u in [0, 2^24].  The exponent e is read from RN(1+u); the reduced argument
 * f = (1+u)*2^-e - 1 is then formed by ONE fma from u itself, so no bits of u are lost. */
FG_FN float fg_log1p(float u)
{
    float w = 1.0f + u;
    uint32_t ix = fg_f2u(w) - FG_SQRT_HALF_BITS;
    int32_t e = (int32_t)ix >> 23;
    float sc = fg_u2f(0x3f800000u - (ix & 0xff800000u)); /* 2^-e */
    float f = FG_FMA(u, sc, sc - 1.0f);
    return fg_log_core(f, (float)e);
}

/* ---- TensorFlow op restatements ----------------------------------------------------------- */
/* tf.math.softplus(t) as lowered by tf2xla: t > -thr -> t ; t < thr -> exp(t) ; else
 * log1p(exp(t)), thr = log(eps)+2.  exp underflow (t < -87) is flushed to 0. */
FG_FN float fg_softplus(float t)
{
    float tc = FG_CLAMP(t, -87.0f, FG_SOFTPLUS_THRESH);
    float y = fg_exp(tc);
    float l = fg_log1p(y);
    float small = (t < -87.0f) ? 0.0f : y;
    float r = (t < -FG_SOFTPLUS_THRESH) ? small : l;
    return (t > FG_SOFTPLUS_THRESH) ? t : r;
}

/* tf.math.reduce_logsumexp(stack([a, b]), axis=-1) = log(exp(a-m) + exp(b-m)) + m, m = max(a,b).
 * One of the two exponentials is exp(0) = 1; the other underflows the sum for |a-b| > 20. */
FG_FN float fg_lse2(float a, float b)
{
    float m = FG_MAX(a, b);
    float d = FG_ABS(a - b);
    float y = fg_exp(-FG_MIN(d, 20.0f));
    return fg_log_1to2(1.0f + y) + m;
}

/* QLDPCBPDecoder._phi, decoding_q.py:365-373:
 *   x = clip(x, 8.5e-8, 16.635532); softplus(x) - log(exp(x) - 1). */
FG_FN float fg_phi(float x)
{
    float xc = FG_CLAMP(x, FG_PHI_MIN, FG_PHI_MAX);
    float y = fg_exp(xc);
    float sp = fg_log1p(y);
    sp = (xc > FG_SOFTPLUS_THRESH) ? xc : sp;
    return sp - fg_log(y - 1.0f);
}

/* GNN_BP4._phi, sionna/fec/ldpc/gnn.py:333-338: same clip, but log(exp(x)+1) - log(exp(x)-1) (no softplus). */
FG_FN float fg_phi_gnn(float x)
{
    float xc = FG_CLAMP(x, FG_PHI_MIN, FG_PHI_MAX);
    float y = fg_exp(xc);
    return fg_log(y + 1.0f) - fg_log(y - 1.0f);
}

/* x / 3 (the mean over a qubit's three edges, feedback_gnn.py:139-141) for any finite x: the CPU divides, the device multiplies by
 * RN(1/3) and corrects with one exact remainder — the correctly rounded quotient for all 2^32 - 2^24 finite inputs
 * (tests/div_exhaustive.hip). */
FG_FN float fg_div3(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float c = 0.333333343f; /* 0x3eaaaaab */
    const float q = x * c;
    const float y = FG_FMA(FG_FMA(-3.0f, q, x), c, q);
    return (x == 0.0f) ? x : y; /* -0 / 3 = -0: the remainder step would return +0 */
#else
    return x / 3.0f;
#endif
}

/* ---- tanh / atanh (feedback-GNN activations, 'boxplus' check-node rule) ------------------ */
/* a / b for b = a + 2, a in [0, 2^58] (the quotient of fg_tanh).  The CPU divides.  The device refines v_rcp_f32 with exact
 * remainders; the result is the correctly rounded quotient — the CPU's — for every float in that range, which
 * tests/div_exhaustive.hip checks one by one on the GPU (tests/test_math.py).  About half the slots of the general division
 * sequence (no v_div_scale / v_div_fixup: nothing here needs rescaling). */
FG_FN float fg_div_em1(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(b);
    float q = a * r;
    float e = FG_FMA(-b, q, a);
    q = FG_FMA(e, r, q);
    return q;
#else
    return a / b;
#endif
}

FG_FN float fg_tanh(float x)
{
    /* tanh|x| = em1/(em1+2), em1 = expm1(2|x|) = 2^k*(e^r - 1) + (2^k - 1) with the reduction and the
     * polynomial of fg_exp.  For k = 0 this is r + r^2*q(r) with r = 2|x| exactly, so small arguments keep
     * full relative accuracy without a second branch. */
    float t = FG_MIN(FG_ABS(x) + FG_ABS(x), 40.0f);
    float tt = FG_FMA(t, FG_LOG2E, FG_RND_MAGIC);
    float k = tt - FG_RND_MAGIC;
    float r = FG_FMA(k, -FG_LN2_HI, t);
    r = FG_FMA(k, -FG_LN2_LO, r);
    float q = 1.381461043e-03f;
    q = FG_FMA(q, r, 8.368710056e-03f);
    q = FG_FMA(q, r, 4.166838899e-02f);
    q = FG_FMA(q, r, 1.666652113e-01f);
    q = FG_FMA(q, r, 4.999999404e-01f);
    float pm1 = FG_FMA(r * r, q, r);                         /* e^r - 1 */
    float sc = fg_u2f((fg_f2u(tt) << 23) + 0x3f800000u);     /* 2^k, k in [0, 58] */
    float em1 = FG_FMA(sc, pm1, sc - 1.0f);
    float y = fg_div_em1(em1, em1 + 2.0f);
    return fg_u2f(fg_f2u(y) | (fg_f2u(x) & 0x80000000u));
}

/* 2a / (1 - a) for a in [0, 1 - 2^-23] (the quotient of fg_atanh) and 1 / t for 2^-126 <= |t| <= 1 (the 'boxplus' rule divides the
 * product of tanh values by each factor, decoding_q.py:340-345): device = v_rcp_f32 + exact-remainder fma steps, CPU = division;
 * equal for every input of those ranges (tests/div_exhaustive.hip).  Outside its range fg_rcp_unit falls back to the division. */
FG_FN float fg_div_atanh(float a)
{
    const float num = a + a, den = 1.0f - a;
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(den);
    float q = num * r;
    q = FG_FMA(FG_FMA(-den, q, num), r, q);
    return q;
#else
    return num / den;
#endif
}
FG_FN float fg_rcp_unit(float t)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float at = FG_ABS(t);
    if (at >= 1.17549435e-38f && at <= 1.0f) {
        float r = __builtin_amdgcn_rcpf(t);
        r = FG_FMA(FG_FMA(-t, r, 1.0f), r, r);
        return r;
    }
#endif
    return 1.0f / t;
}

/* atanh(x) = 0.5*log1p(2|x|/(1-|x|)), |x| <= 1-2^-23 */
FG_FN float fg_atanh(float x)
{
    float ax = FG_ABS(x);
    float r = 0.5f * fg_log1p(fg_div_atanh(ax));
    return fg_u2f(fg_f2u(r) | (fg_f2u(x) & 0x80000000u));
}

/* logistic sigmoid 1/(1+e^-x) (Keras 'sigmoid' activation of the general feedback GNN), stable on both sides */
FG_FN float fg_sigmoid(float x)
{
    float e = fg_exp(-FG_MIN(FG_ABS(x), 87.0f));
    float d = 1.0f + e;
    return (x >= 0.0f) ? 1.0f / d : e / d;
}

#endif /* FGNN_MATH_H */
