/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Software IEEE-754 binary16 for the CPU restatement of piet-metal's
 * renderKernel, whose accumulators are `half` (reference
 * TestApp/PietRender.metal:470-472, :526, :532, :537).
 *
 * gcc 11 on x86-64 has no _Float16, so every half operation is done as
 * "widen to f32 (exact), operate in f32, round once to binary16 (RNE)".
 * For +,-,* and / of two binary16 values this is identical to a correctly
 * rounded binary16 operation (f32 carries 24 >= 2*11+2 significand bits, so
 * the double rounding is innocuous).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.
 */
#ifndef PMO_HALF_H
#define PMO_HALF_H

#include <stdint.h>
#include <string.h>

typedef uint16_t pmo_half; /* raw binary16 bits */

static inline uint32_t pmo_f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static inline float pmo_bits_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* f32 -> binary16, round-to-nearest-even, denormals kept. */
static inline pmo_half pmo_f2h(float f) {
    uint32_t x = pmo_f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    uint32_t o;
    if (x >= ((127u + 16u) << 23)) {
        /* |f| >= 65536, Inf or NaN */
        o = (x > (255u << 23)) ? 0x7e00u : 0x7c00u;
    } else if (x < (113u << 23)) {
        /* result is a binary16 subnormal (or zero): let the f32 adder round */
        const uint32_t magic = ((127u - 15u) + (23u - 10u) + 1u) << 23;
        float t = pmo_bits_f32(x) + pmo_bits_f32(magic);
        o = pmo_f32_bits(t) - magic;
    } else {
        uint32_t mant_odd = (x >> 13) & 1u;
        x += ((uint32_t)(15 - 127) << 23) + 0xfffu;
        x += mant_odd;
        o = x >> 13;
    }
    return (pmo_half)(sign | o);
}

/* binary16 -> f32, exact. */
static inline float pmo_h2f(pmo_half h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t em = (uint32_t)h & 0x7fffu;
    uint32_t o;
    if (em >= 0x7c00u) {
        o = 0x7f800000u | ((em & 0x3ffu) << 13);
    } else if (em >= 0x0400u) {
        o = (em << 13) + ((127u - 15u) << 23);
    } else {
        /* subnormal: em * 2^-24 */
        float t = (float)em * (1.0f / 16777216.0f);
        o = pmo_f32_bits(t);
    }
    return pmo_bits_f32(sign | o);
}

static inline pmo_half pmo_hadd(pmo_half a, pmo_half b) { return pmo_f2h(pmo_h2f(a) + pmo_h2f(b)); }
static inline pmo_half pmo_hsub(pmo_half a, pmo_half b) { return pmo_f2h(pmo_h2f(a) - pmo_h2f(b)); }
static inline pmo_half pmo_hmul(pmo_half a, pmo_half b) { return pmo_f2h(pmo_h2f(a) * pmo_h2f(b)); }

/* MSL mix(x, y, a) = x + (y - x) * a, every operation rounded to binary16
 * (decision D1 in SURVEY.md section 3.3). */
static inline pmo_half pmo_hmix(pmo_half x, pmo_half y, pmo_half a) {
    return pmo_hadd(x, pmo_hmul(pmo_hsub(y, x), a));
}

#define PMO_H_ZERO ((pmo_half)0x0000)
#define PMO_H_ONE ((pmo_half)0x3c00)

#endif /* PMO_HALF_H */
