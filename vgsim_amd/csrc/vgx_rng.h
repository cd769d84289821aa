// vgx_rng.h — device/host random streams and the portable logarithm used by the kernels.
//
// PCG64 (XSL-RR 128/64) seeded through numpy's SeedSequence(entropy, spawn_key=(k,)): the stream that
// the reference's RndmWrapper(seed=(user_seed, k)) hands out (mc_lib.rndm, call sites
// src/_BirthDeath.pyx:403, 477, 488, 2310); uniform() = (next64 >> 11) * 2^-53.
// Philox4x32-10 (Salmon et al., SC'11): counter-based stream for the lane-parallel tau-leap kernel.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VGX_HD __host__ __device__ __forceinline__
#else
#define VGX_HD inline
#endif

struct VgxPcg64 { uint64_t sh, sl, ih, il; };

VGX_HD uint64_t vgx_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

VGX_HD void vgx_pcg64_step(VgxPcg64 &g) {  // state = state * MULT + inc (mod 2^128)
    const uint64_t MH = 0x2360ED051FC65DA4ull, ML = 0x4385DF649FCCF645ull;
    uint64_t lo = g.sl * ML;
    uint64_t hi = vgx_mulhi64(g.sl, ML) + g.sh * ML + g.sl * MH;
    uint64_t nl = lo + g.il;
    uint64_t carry = nl < lo ? 1u : 0u;
    g.sh = hi + g.ih + carry;
    g.sl = nl;
}

VGX_HD uint32_t vgx_ss_hashmix(uint32_t value, uint32_t &hc) {
    value ^= hc;
    hc *= 0x931e8875u;
    value *= hc;
    value ^= value >> 16;
    return value;
}
VGX_HD uint32_t vgx_ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    r ^= r >> 16;
    return r;
}

// SeedSequence(entropy, spawn_key=(spawn_key,)).generate_state(4, uint64) -> PCG64 (state, inc)
VGX_HD void vgx_pcg64_seed(VgxPcg64 &g, uint64_t entropy, uint32_t spawn_key) {
    uint32_t ent[5];
    ent[0] = (uint32_t)entropy;
    ent[1] = (uint32_t)(entropy >> 32);   // a zero high word equals the zero padding of a one-word entropy
    ent[2] = 0; ent[3] = 0; ent[4] = spawn_key;
    uint32_t pool[4];
    uint32_t hc = 0x43b0d7e5u;
#pragma unroll
    for (int i = 0; i < 4; i++) pool[i] = vgx_ss_hashmix(ent[i], hc);
#pragma unroll
    for (int is = 0; is < 4; is++)
#pragma unroll
        for (int id = 0; id < 4; id++)
            if (is != id) pool[id] = vgx_ss_mix(pool[id], vgx_ss_hashmix(pool[is], hc));
#pragma unroll
    for (int id = 0; id < 4; id++) pool[id] = vgx_ss_mix(pool[id], vgx_ss_hashmix(ent[4], hc));
    uint32_t w[8];
    uint32_t hb = 0x8b51f9ddu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= 0x58f38dedu;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t s0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), s1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    uint64_t s2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), s3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    // inc = (initseq << 1) | 1 ; state = 0 ; step ; state += initstate ; step
    g.ih = (s2 << 1) | (s3 >> 63);
    g.il = (s3 << 1) | 1u;
    g.sh = 0; g.sl = 0;
    vgx_pcg64_step(g);
    uint64_t nl = g.sl + s1;
    uint64_t carry = nl < g.sl ? 1u : 0u;
    g.sh = g.sh + s0 + carry;
    g.sl = nl;
    vgx_pcg64_step(g);
}

VGX_HD uint64_t vgx_pcg64_next(VgxPcg64 &g) {
    vgx_pcg64_step(g);
    uint64_t x = g.sh ^ g.sl;
    unsigned rot = (unsigned)(g.sh >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
VGX_HD double vgx_pcg64_double(VgxPcg64 &g) {
    return (double)(vgx_pcg64_next(g) >> 11) * (1.0 / 9007199254740992.0);
}

// 128-bit helpers (mod 2^128)
VGX_HD void vgx_mul128(uint64_t ah, uint64_t al, uint64_t bh, uint64_t bl, uint64_t &rh, uint64_t &rl) {
    rl = al * bl;
    rh = vgx_mulhi64(al, bl) + ah * bl + al * bh;
}
VGX_HD void vgx_add128(uint64_t &ah, uint64_t &al, uint64_t bh, uint64_t bl) {
    uint64_t nl = al + bl;
    ah = ah + bh + (nl < al ? 1u : 0u);
    al = nl;
}
VGX_HD double vgx_pcg64_output_double(uint64_t sh, uint64_t sl) {  // XSL-RR of a state, as a double in [0,1)
    uint64_t x = sh ^ sl;
    unsigned rot = (unsigned)(sh >> 58);
    uint64_t o = (x >> rot) | (x << ((64u - rot) & 63u));
    return (double)(o >> 11) * (1.0 / 9007199254740992.0);
}

// ---- Philox4x32-10 -------------------------------------------------------------------------------
struct VgxPhilox { uint32_t c[4]; uint32_t k[2]; };
VGX_HD void vgx_philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// 128 random bits for (counter, key)
VGX_HD void vgx_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
    for (int r = 0; r < 10; r++) {
        vgx_philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// ---- counter-based uniform stream (FAST mode with vgx_run_opts.mode = 2) ---------------------------
// Output number n = 0, 1, ... of the stream of (seed, attempt): the low (n even) / high (n odd) 64 bits of Philox4x32-10
// with counter (n >> 1 as two words, attempt, 'VGXs') and the seed's two halves as key; as a double like PCG64's outputs
// (top 53 bits).  Any output can be formed independently of the others: a wavefront's lanes fill a batch without the
// 128-bit jump-ahead arithmetic of the PCG64 stream, and the host clock regenerates single uniforms.
VGX_HD uint64_t vgx_philox_stream_u64(uint64_t seed, uint32_t attempt, uint64_t n) {
    const uint64_t blk = n >> 1;
    const uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), attempt, 0x56475873u};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    vgx_philox4x32(ctr, key, o);
    return (n & 1) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
}
VGX_HD double vgx_philox_stream_double(uint64_t seed, uint32_t attempt, uint64_t n) {
    return (double)(vgx_philox_stream_u64(seed, attempt, n) >> 11) * (1.0 / 9007199254740992.0);
}

// ---- portable natural logarithm ------------------------------------------------------------------
// fdlibm e_log.c algorithm (argument reduction to [sqrt(2)/2, sqrt(2)], s = f/(2+f), degree-14 even
// polynomial; < 1 ulp).  Only +,-,*,/ on binary64 and no contraction, so host and device agree bit for
// bit; it differs from glibc's log (what the reference calls, pyx:477) by at most 1 ulp.
VGX_HD double vgx_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                 Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                 Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    union { double d; uint64_t u; } cv;
    cv.d = x;
    int32_t hx = (int32_t)(cv.u >> 32);
    uint32_t lx = (uint32_t)cv.u;
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) { cv.u = 0xfff0000000000000ull; return cv.d; }  // log(0) = -inf
        if (hx < 0) { cv.u = 0x7ff8000000000000ull; return cv.d; }
        k -= 54;
        x *= two54;
        cv.d = x;
        hx = (int32_t)(cv.u >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    cv.u = (cv.u & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    x = cv.d;
    k += (i >> 20);
    double f = x - 1.0;
    double dk;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    dk = (double)k;
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    } else {
        if (k == 0) return f - s * (f - R);
        return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
    }
}
