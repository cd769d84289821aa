// vgx_tau_rng.h — random streams and samplers shared by the tau-leaping kernels (vgx_tau.hip: step kernels of large models;
// vgx_taus.hip: the on-device step loop of small models).  Philox4x32-10 keyed by (seed, attempt) with the counter
// (compartment, step, retry): every compartment owns an independent, reproducible stream, whatever the launch geometry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_rng.h"

// The tau path is validated distributionally, so FMA contraction is allowed in its translation units (the library is built
// with -ffp-contract=off for the bit-exact direct kernels); the samplers below have always been compiled that way.
#pragma clang fp contract(fast)

struct TauRng {
    uint32_t c0, c1, c2, c3, k0, k1;
    uint64_t spare;
    bool have;
    __device__ void init(uint64_t seed, uint32_t attempt, uint64_t cell, uint32_t step, uint32_t retry) {
        k0 = (uint32_t)seed ^ (attempt * 0x9E3779B9u);
        k1 = (uint32_t)(seed >> 32) ^ 0x85EBCA6Bu;
        c0 = (uint32_t)cell;
        c1 = (uint32_t)(cell >> 32);
        c2 = step;
        c3 = retry << 20;  // low 20 bits: block counter of this stream
        have = false;
        spare = 0;
    }
    __device__ double uniform() {  // (0,1): 52 random bits + half an ulp, never 0
        uint64_t x;
        if (have) {
            x = spare;
            have = false;
        } else {
            uint32_t ctr[4] = {c0, c1, c2, c3}, key[2] = {k0, k1}, out[4];
            vgx_philox4x32(ctr, key, out);
            c3 += 1;
            x = ((uint64_t)out[1] << 32) | out[0];
            spare = ((uint64_t)out[3] << 32) | out[2];
            have = true;
        }
        return ((double)(x >> 12) + 0.5) * (1.0 / 4503599627370496.0);
    }
};

// log Gamma for PTRS (Stirling series with upward recurrence below 7, as in numpy's random_loggam)
static __device__ double tau_loggam(double x) {
    const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04,
                          -5.952380952380952e-04, 8.417508417508418e-04, -1.917526917526918e-03,
                          6.410256410256410e-03, -2.955065359477124e-02, 1.796443723688307e-01,
                          -1.39243221690590e+00};
    if (x == 1.0 || x == 2.0) return 0.0;
    int n = x < 7.0 ? (int)(7 - x) : 0;
    double x0 = x + n;
    double x2 = (1.0 / x0) * (1.0 / x0);
    double gl0 = a[9];
    for (int k = 8; k >= 0; k--) gl0 = gl0 * x2 + a[k];
    double gl = gl0 / x0 + 0.5 * 1.8378770664093453e+00 + (x0 - 0.5) * log(x0) - x0;
    for (int k = 1; k <= n; k++) { gl -= log(x0 - 1.0); x0 -= 1.0; }
    return gl;
}

// Poisson(lam).  The reference's sampler (numpy random_poisson, pyx:2532) uses the multiplication method below 10
// and PTRS (Hoermann 1993) from 10 on; here the small-mean branch is inversion by sequential search (one uniform
// per draw instead of k+1: same law, no random numbers inside the divergent loop), the large-mean branch PTRS.
static __device__ int64_t tau_poisson(TauRng &g, double lam) {
    if (!(lam > 0.0)) return 0;
    if (lam < 10.0) {
        double u = g.uniform();
        if (u <= 1.0 - lam) return 0;   // exp(-lam) >= 1 - lam: the search below would stop at 0 (most compartments)
        double pk = exp(-lam), F = pk;
        int64_t X = 0;
        while (u > F && X < 200) {
            X += 1;
            pk *= lam / (double)X;
            F += pk;
        }
        return X;
    }
    double slam = sqrt(lam), loglam = log(lam);
    double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2);
    while (true) {
        double U = g.uniform() - 0.5, V = g.uniform();
        double us = 0.5 - fabs(U);
        int64_t k = (int64_t)floor((2 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0 || (us < 0.013 && V > us)) continue;
        // (log V + log invalpha - log(a / us^2 + b) as one logarithm: the wavefront pays for every log some lane needs)
        if (log(V * invalpha / (a / (us * us) + b)) <= (-lam + k * loglam - tau_loggam((double)k + 1))) return k;
    }
}

static __device__ __forceinline__ int tau_mutate(int sites, int hi, int s, int DS) {  // pyx:2420-2427
    int digit4 = 1 << (2 * (sites - s - 1));
    int AS = (hi / digit4) % 4;
    if (DS >= AS) DS += 1;
    return hi + (DS - AS) * digit4;
}

