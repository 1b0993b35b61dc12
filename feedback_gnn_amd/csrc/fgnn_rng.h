/* fgnn_rng.h — counter-based noise source for the depolarizing channel.
 *
 * The reference draws one tf.random.uniform float32 per qubit (sionna/channel/pauli.py:100) from
 * TensorFlow's unseeded global generator, so its sample stream is not reproducible.  The build
 * defines the stream instead: Philox4x32-10 (Salmon et al., SC'11; same generator family as
 * tf.random's stateless ops), key = 64-bit seed, counter = (global sample index, word block), so
 * the noise of sample i is the same on the CPU oracle, on one GPU and on any shard of 8 GPUs.
 * Integer arithmetic only (plus the float32 thresholds of the channel); shared by the HIP kernels and the oracle.
 */
#ifndef FGNN_RNG_H
#define FGNN_RNG_H

#include <stdint.h>
#include "fgnn_math.h"

FG_FN void fg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                            uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uint32 -> float32 in [0,1): 23 random mantissa bits, as TensorFlow's Uint32ToFloat. */
FG_FN float fg_u32_to_unit(uint32_t x) { return fg_u2f((x >> 9) | 0x3f800000u) - 1.0f; }

/* uniforms for qubits 4*block .. 4*block+3 of one sample */
FG_FN void fg_uniform4(uint64_t seed, uint64_t sample, uint32_t block, float u[4])
{
    uint32_t r[4];
    fg_philox4x32_10((uint32_t)sample, (uint32_t)(sample >> 32), block, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    u[0] = fg_u32_to_unit(r[0]);
    u[1] = fg_u32_to_unit(r[1]);
    u[2] = fg_u32_to_unit(r[2]);
    u[3] = fg_u32_to_unit(r[3]);
}

/* same with an explicit stream id in the 4th counter word (0 = i.i.d. channel noise, 1 = fixed-weight positions,
 * 2 = fixed-weight Pauli types) */
FG_FN void fg_uniform4s(uint64_t seed, uint64_t sample, uint32_t block, uint32_t stream, float u[4])
{
    uint32_t r[4];
    fg_philox4x32_10((uint32_t)sample, (uint32_t)(sample >> 32), block, stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    u[0] = fg_u32_to_unit(r[0]);
    u[1] = fg_u32_to_unit(r[1]);
    u[2] = fg_u32_to_unit(r[2]);
    u[3] = fg_u32_to_unit(r[3]);
}

/* Fixed-weight errors, Pauli.call wt branch (pauli.py:80-97): `wt` distinct positions (first wt entries of a uniform
 * shuffle) and per position u ~ U[0,1): X component iff u < 2/3, Z component iff u > 1/3.  The build's stream:
 * partial Fisher-Yates, step i swaps perm[i] with perm[i + min(floor(u_i*(n-i)), n-i-1)], u_i from stream 1,
 * the type uniform of position i from stream 2.  `perm` is caller scratch of n entries (identity on entry). */
FG_FN int fg_fy_pick(float u, int remaining)
{
    int j = (int)(u * (float)remaining);
    return j < remaining - 1 ? j : remaining - 1;
}

/* Pauli.call thresholds (pauli.py:100-108) for any triple (px, py, pz), evaluated in float32 like the reference's tf.float32 graph:
 *   noise_x = u < px ;  noise_z = (u >= px - py) & (u < (px + pz) - py),
 * i.e. u in [0, px - py): X, [px - py, px): Y, [px, px + pz - py): Z.  Additions and subtractions only: the same bits on any IEEE host
 * or device.  fg_pauli_thresholds(p) is the depolarizing split px = pz = 2p/3, py = p/3 of every caller in the reference
 * (feedback_gnn.py:298, bp_osd.py:107). */
typedef struct { float px, lo, hi; } fg_pauli_thr;

FG_FN fg_pauli_thr fg_pauli_thresholds_xyz(float px, float py, float pz)
{
    fg_pauli_thr t;
    t.px = px;
    t.lo = px - py;
    t.hi = (px + pz) - py;
    return t;
}
FG_FN fg_pauli_thr fg_pauli_thresholds(float p)
{
    return fg_pauli_thresholds_xyz((2.0f * p) / 3.0f, p / 3.0f, (2.0f * p) / 3.0f);
}
FG_FN uint8_t fg_pauli_x(float u, fg_pauli_thr t) { return (uint8_t)(u < t.px); }
FG_FN uint8_t fg_pauli_z(float u, fg_pauli_thr t) { return (uint8_t)((u >= t.lo) && (u < t.hi)); }

#endif /* FGNN_RNG_H */
