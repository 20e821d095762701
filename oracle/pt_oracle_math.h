/*
 * TEST INFRASTRUCTURE ONLY -- part of the CPU oracle, never linked into or called by the
 * product library.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.
 *
 * Counter-based random numbers and the deterministic sine/cosine pair that the oracle, the
 * host build of the reference and the HIP kernels all evaluate identically.
 *
 * The reference fills a device buffer with cuRAND XORWOW seeded by time(NULL)
 * (pathtracer.cu:215-216,224), so its random floats are not reproducible by design.  This
 * build replaces the buffer by Philox4x32-10 (Salmon et al., "Parallel Random Numbers: As
 * Easy as 1, 2, 3", SC'11; Random123 v1.09 known-answer vectors are checked in
 * tests/test_oracle_rng.py) addressed by (pixel, iteration, slot, stream), which reproduces
 * the buffer's *layout* contract (pathtracer.cu:70,92,131,141,155):
 *
 *   stream 0 (camera), slot 0 -> r0..r3, slot 1 -> r4..r7  : the 8 jitter floats of a pixel
 *   stream 1 (path),   slot b -> { u_bsdf, v_bsdf, rr, - } : the 3 floats of bounce b
 *
 * All four sub-pixel paths of a pixel share the path stream, as in RenderKernel
 * (pathtracer.cu:92,99).
 */
#ifndef PT_ORACLE_MATH_H
#define PT_ORACLE_MATH_H

#include <stdint.h>

#define PTOR_STREAM_CAMERA 0u
#define PTOR_STREAM_PATH   1u
#define PTOR_STREAM_LIGHT  2u   /* direct-light sampling: slot = bounce -> (u, v, light choice, -) */

/* Philox4x32-10: multipliers and Weyl key increments from the Random123 paper. */
static inline void ptor_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
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

/* 24-bit uniform in [0,1): every value is exactly representable, so host and device agree. */
static inline float ptor_u01(uint32_t x)
{
    return (float)(x >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
}

static inline void ptor_rng4(uint64_t seed, uint32_t pixel, uint32_t iteration, uint32_t slot,
                             uint32_t stream, float out[4])
{
    uint32_t ctr[4] = { pixel, iteration, slot, stream };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t r[4];
    ptor_philox4x32_10(ctr, key, r);
    out[0] = ptor_u01(r[0]); out[1] = ptor_u01(r[1]);
    out[2] = ptor_u01(r[2]); out[3] = ptor_u01(r[3]);
}

/*
 * Deterministic sinf/cosf for arguments in [0, ~8]: the only transcendental pair on the hot
 * path (ConcentricSampleDisk, montecarlo.h:115-117, theta in [0, 2*pi]).  libm's and the GPU
 * math library's sinf differ in the last bit, which would make paths diverge after the first
 * diffuse bounce, so both sides evaluate this fixed sequence of fp32 multiplies and adds
 * (no fused multiply-add; compile with -ffp-contract=off).
 *
 * Reduction: k = nearest multiple of pi/2, three-term Cody-Waite (k <= 5 keeps k*P1 and k*P2
 * exact).  Kernels: the Cephes single-precision minimax polynomials on [-pi/4, pi/4].
 */
static inline void ptor_sincosf(float x, float *s_out, float *c_out)
{
    const float two_over_pi = 0.636619772367581343f;
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188216e-8f;
    int k = (int)(x * two_over_pi + 0.5f);
    float kf = (float)k;
    float r = ((x - kf * P1) - kf * P2) - kf * P3;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z;
    float s = r + r * ps;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f);
    float c = (1.0f - 0.5f * z) + (z * z) * pc;
    float ss, cc;
    switch (k & 3) {
    case 0:  ss =  s; cc =  c; break;
    case 1:  ss =  c; cc = -s; break;
    case 2:  ss = -s; cc = -c; break;
    default: ss = -c; cc =  s; break;
    }
    *s_out = ss;
    *c_out = cc;
}

static inline float pt_sinf(float x) { float s, c; ptor_sincosf(x, &s, &c); return s; }
static inline float pt_cosf(float x) { float s, c; ptor_sincosf(x, &s, &c); return c; }

#endif /* PT_ORACLE_MATH_H */
