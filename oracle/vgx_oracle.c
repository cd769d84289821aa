/*
 * vgx_oracle.c — TEST INFRASTRUCTURE (see vgx_oracle.h for scope and parity status).
 *
 * Op-for-op restatement in plain C of the reference's forward simulators.  Citations: pyx:N =
 * /root/reference/src/_BirthDeath.pyx, fc:N = src/fast_choose.pxi, ev:N = src/events.pxi.
 * Build with -O2 -ffp-contract=off (oracle/Makefile): floating-point operation order is the contract.
 */
#include "vgx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

#define P_ (m->popNum)
#define H_ (m->hapNum)
#define S_ (m->susNum)
#define SITES_ (m->sites)
#define IDX2(a, i, j, nj) ((a)[(i) * (nj) + (j)])
#define IDX3(a, i, j, k, nj, nk) ((a)[((i) * (nj) + (j)) * (nk) + (k)])

/* =====================================================================================
 * RNG: numpy SeedSequence + PCG64 (XSL-RR 128/64), as wrapped by mc_lib.rndm.RndmWrapper
 * (call sites pyx:74,403,2310: RndmWrapper(seed=(user_seed, attempt)); pyx:477,488: uniform()).
 * ===================================================================================== */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L 0xca01f9ddu
#define SS_MIX_R 0x4973f715u
#define SS_XSHIFT 16
#define SS_POOL 4

static uint32_t ss_hashmix(uint32_t value, uint32_t *hash_const) {
    value ^= *hash_const;
    *hash_const *= SS_MULT_A;
    value *= *hash_const;
    value ^= value >> SS_XSHIFT;
    return value;
}
static uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
    r ^= r >> SS_XSHIFT;
    return r;
}

#define PCG_MULT_HI 0x2360ED051FC65DA4ull
#define PCG_MULT_LO 0x4385DF649FCCF645ull

static inline void pcg_step(vgo_pcg64 *g) {
    u128 s = ((u128)g->state_hi << 64) | g->state_lo;
    u128 inc = ((u128)g->inc_hi << 64) | g->inc_lo;
    u128 mult = ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO;
    s = s * mult + inc;
    g->state_hi = (uint64_t)(s >> 64);
    g->state_lo = (uint64_t)s;
}

void vgo_pcg64_seed(vgo_pcg64 *g, uint64_t entropy, uint32_t spawn_key) {
    /* SeedSequence(entropy, spawn_key=(spawn_key,)): entropy as little-endian uint32 words (at least
     * one), padded with zeros to the pool size when a spawn key follows, then the spawn-key words. */
    uint32_t ent[8];
    int n = 0;
    if (entropy == 0) {
        ent[n++] = 0;
    } else {
        while (entropy) { ent[n++] = (uint32_t)entropy; entropy >>= 32; }
    }
    while (n < SS_POOL) ent[n++] = 0;
    ent[n++] = spawn_key;

    uint32_t pool[SS_POOL];
    uint32_t hc = SS_INIT_A;
    for (int i = 0; i < SS_POOL; i++) pool[i] = ss_hashmix(ent[i], &hc);
    for (int is = 0; is < SS_POOL; is++)
        for (int id = 0; id < SS_POOL; id++)
            if (is != id) pool[id] = ss_mix(pool[id], ss_hashmix(pool[is], &hc));
    for (int is = SS_POOL; is < n; is++)
        for (int id = 0; id < SS_POOL; id++) pool[id] = ss_mix(pool[id], ss_hashmix(ent[is], &hc));

    /* generate_state(4, uint64) = 8 uint32 words, viewed little-endian as 4 uint64 */
    uint32_t w[8];
    uint32_t hb = SS_INIT_B;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i % SS_POOL];
        v ^= hb;
        hb *= SS_MULT_B;
        v *= hb;
        v ^= v >> SS_XSHIFT;
        w[i] = v;
    }
    uint64_t s0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), s1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    uint64_t s2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), s3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    /* pcg64_set_seed: initstate = (s0<<64)|s1, initseq = (s2<<64)|s3; pcg_setseq_128_srandom_r */
    u128 initstate = ((u128)s0 << 64) | s1, initseq = ((u128)s2 << 64) | s3;
    u128 inc = (initseq << 1) | 1u;
    g->inc_hi = (uint64_t)(inc >> 64);
    g->inc_lo = (uint64_t)inc;
    g->state_hi = 0;
    g->state_lo = 0;
    pcg_step(g);
    u128 s = (((u128)g->state_hi << 64) | g->state_lo) + initstate;
    g->state_hi = (uint64_t)(s >> 64);
    g->state_lo = (uint64_t)s;
    pcg_step(g);
}

uint64_t vgo_pcg64_next64(vgo_pcg64 *g) {
    pcg_step(g);
    uint64_t hi = g->state_hi, lo = g->state_lo;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

double vgo_pcg64_double(vgo_pcg64 *g) { return (double)(vgo_pcg64_next64(g) >> 11) * (1.0 / 9007199254740992.0); }

void vgo_pcg64_advance(vgo_pcg64 *g, uint64_t delta_hi, uint64_t delta_lo) {
    u128 delta = ((u128)delta_hi << 64) | delta_lo;
    u128 cur_mult = ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO;
    u128 cur_plus = ((u128)g->inc_hi << 64) | g->inc_lo;
    u128 acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    u128 s = ((u128)g->state_hi << 64) | g->state_lo;
    s = acc_mult * s + acc_plus;
    g->state_hi = (uint64_t)(s >> 64);
    g->state_lo = (uint64_t)s;
}

/* =====================================================================================
 * numpy random_poisson (numpy/random/src/distributions/distributions.c, numpy 2.2), the function
 * pyx:2531-2532 calls: lam == 0 -> 0 without drawing; lam < 10 multiplication method; else PTRS
 * (W. Hoermann, "The transformed rejection method for generating Poisson random variables", 1993).
 * ===================================================================================== */
static double np_loggam(double x) {
    static const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04,
                                 -5.952380952380952e-04, 8.417508417508418e-04, -1.917526917526918e-03,
                                 6.410256410256410e-03, -2.955065359477124e-02, 1.796443723688307e-01,
                                 -1.39243221690590e+00};
    double x0, x2, lg2pi, gl, gl0;
    int64_t k, n;
    if ((x == 1.0) || (x == 2.0)) return 0.0;
    else if (x < 7.0) n = (int64_t)(7 - x);
    else n = 0;
    x0 = x + n;
    x2 = (1.0 / x0) * (1.0 / x0);
    lg2pi = 1.8378770664093453e+00;
    gl0 = a[9];
    for (k = 8; k >= 0; k--) { gl0 *= x2; gl0 += a[k]; }
    gl = gl0 / x0 + 0.5 * lg2pi + (x0 - 0.5) * log(x0) - x0;
    if (x < 7.0) {
        for (k = 1; k <= n; k++) { gl -= log(x0 - 1.0); x0 -= 1.0; }
    }
    return gl;
}

int64_t vgo_poisson(vgo_pcg64 *g, double lam) {
    if (lam >= 10) {
        int64_t k;
        double U, V, slam, loglam, a, b, invalpha, vr, us;
        slam = sqrt(lam);
        loglam = log(lam);
        b = 0.931 + 2.53 * slam;
        a = -0.059 + 0.02483 * b;
        invalpha = 1.1239 + 1.1328 / (b - 3.4);
        vr = 0.9277 - 3.6224 / (b - 2);
        while (1) {
            U = vgo_pcg64_double(g) - 0.5;
            V = vgo_pcg64_double(g);
            us = 0.5 - fabs(U);
            k = (int64_t)floor((2 * a / us + b) * U + lam + 0.43);
            if ((us >= 0.07) && (V <= vr)) return k;
            if ((k < 0) || ((us < 0.013) && (V > us))) continue;
            if ((log(V) + log(invalpha) - log(a / (us * us) + b)) <= (-lam + k * loglam - np_loggam(k + 1)))
                return k;
        }
    } else if (lam == 0) {
        return 0;
    } else {
        int64_t X = 0;
        double prod = 1.0, U, enlam = exp(-lam);
        while (1) {
            U = vgo_pcg64_double(g);
            prod *= U;
            if (prod > enlam) X += 1;
            else return X;
        }
    }
}

/* =====================================================================================
 * Portable natural logarithm (fdlibm e_log.c algorithm, Sun Microsystems 1993; < 1 ulp).  Pure
 * +,-,*,/ in binary64 with no contraction, so the HIP kernel's copy gives bit-identical results.
 * ===================================================================================== */
double vgo_portable_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                        Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                        Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    double hfsq, f, s, z, R, w, t1, t2, dk;
    int32_t k, hx, i, j;
    uint32_t lx;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    hx = (int32_t)(bits >> 32);
    lx = (uint32_t)bits;
    k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -INFINITY;
        if (hx < 0) return NAN;
        k -= 54;
        x *= two54;
        memcpy(&bits, &x, 8);
        hx = (int32_t)(bits >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    bits = (bits & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    memcpy(&x, &bits, 8);
    k += (i >> 20);
    f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    } else {
        if (k == 0) return f - s * (f - R);
        return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
    }
}

/* =====================================================================================
 * fastChoose / fastChoose_skip (fc:18-31, fc:36-52), double and int64 instantiations.
 * Zero weight at the chosen index: the reference prints and sys.exit(1)s (fc:5-13); here an error code.
 * ===================================================================================== */
typedef struct { int64_t i; double rn; } choice_t;

static choice_t choose_f64(vgo_model *m, const double *w, int64_t n, double tw, double rn) {
    choice_t c;
    int64_t i = 0;
    double total;
    rn = tw * rn;
    total = w[0];
    while (total < rn && i < n - 1) { i += 1; total += w[i]; }
    if (w[i] == 0.0) m->error = VGO_ERR_ZERO_WEIGHT;
    c.i = i;
    c.rn = (rn - (total - w[i])) / w[i];
    return c;
}

static choice_t choose_i64(vgo_model *m, const int64_t *w, int64_t n, int64_t tw, double rn) {
    choice_t c;
    int64_t i = 0, total;
    rn = (double)tw * rn;
    total = w[0];
    while ((double)total < rn && i < n - 1) { i += 1; total += w[i]; }
    if (w[i] == 0) m->error = VGO_ERR_ZERO_WEIGHT;
    c.i = i;
    c.rn = (rn - (double)(total - w[i])) / (double)w[i];
    return c;
}

static choice_t choose_skip_i64(vgo_model *m, const int64_t *w, int64_t n, int64_t tw, double rn, int64_t skip) {
    choice_t c;
    int64_t i = 0, total;
    rn = (double)tw * rn;
    if (skip == 0) i += 1;
    total = w[i];
    while ((double)total < rn && i < n - 1) {
        i += 1;
        if (i != skip) total += w[i];
    }
    if (w[i] == 0) m->error = VGO_ERR_ZERO_WEIGHT;
    c.i = i;
    c.rn = (rn - (double)(total - w[i])) / (double)w[i];
    return c;
}

/* ---- sparse (occupied-only) variants: zero weights are exact no-ops in the running sum and can never
 * be the stopping index unless the scan clamps at n-1, so visiting only set bits of the
 * occupancy bitmap in index order is bit-identical. ---- */
static inline int64_t occ_words(const vgo_model *m) { return (m->hapNum + 63) / 64; }
static inline void occ_set(vgo_model *m, int64_t pi, int64_t hn, int on) {
    uint64_t *w = &m->occ[pi * occ_words(m) + (hn >> 6)];
    if (on) *w |= (1ull << (hn & 63));
    else *w &= ~(1ull << (hn & 63));
}
/* next occupied index >= from, or H if none */
static int64_t occ_next(const vgo_model *m, int64_t pi, int64_t from) {
    int64_t nw = occ_words(m), wi = from >> 6;
    if (from >= m->hapNum) return m->hapNum;
    const uint64_t *row = &m->occ[pi * nw];
    uint64_t cur = row[wi] & (~0ull << (from & 63));
    while (1) {
        if (cur) {
            int64_t r = (wi << 6) + __builtin_ctzll(cur);
            return r < m->hapNum ? r : m->hapNum;
        }
        wi++;
        if (wi >= nw) return m->hapNum;
        cur = row[wi];
    }
}
static void occ_rebuild(vgo_model *m) {
    memset(m->occ, 0, (size_t)(P_ * occ_words(m)) * 8);
    for (int64_t pn = 0; pn < P_; pn++)
        for (int64_t hn = 0; hn < H_; hn++)
            if (IDX2(m->infectious, pn, hn, H_) != 0) occ_set(m, pn, hn, 1);
}

static choice_t choose_f64_sparse(vgo_model *m, int64_t pi, const double *w, double tw, double rn) {
    /* same as choose_f64 over w[0..H-1], skipping indices with infectious == 0 (w == 0.0 there) */
    choice_t c;
    int64_t n = H_, i, nx;
    double total;
    rn = tw * rn;
    i = 0;
    total = w[0];
    if (!(total < rn && i < n - 1)) goto done;
    nx = occ_next(m, pi, 1);
    while (1) {
        if (nx >= n) { i = n - 1; break; }  /* only zeros remain: the dense loop runs to n-1 */
        i = nx;
        total += w[i];
        if (!(total < rn && i < n - 1)) break;
        nx = occ_next(m, pi, i + 1);
    }
done:
    if (w[i] == 0.0) m->error = VGO_ERR_ZERO_WEIGHT;
    c.i = i;
    c.rn = (rn - (total - w[i])) / w[i];
    return c;
}

static choice_t choose_i64_sparse(vgo_model *m, int64_t pi, const int64_t *w, int64_t tw, double rn) {
    choice_t c;
    int64_t n = H_, i, nx, total;
    rn = (double)tw * rn;
    i = 0;
    total = w[0];
    if (!((double)total < rn && i < n - 1)) goto done;
    nx = occ_next(m, pi, 1);
    while (1) {
        if (nx >= n) { i = n - 1; break; }
        i = nx;
        total += w[i];
        if (!((double)total < rn && i < n - 1)) break;
        nx = occ_next(m, pi, i + 1);
    }
done:
    if (w[i] == 0) m->error = VGO_ERR_ZERO_WEIGHT;
    c.i = i;
    c.rn = (rn - (double)(total - w[i])) / (double)w[i];
    return c;
}

/* =====================================================================================
 * State helpers
 * ===================================================================================== */
typedef struct { vgo_pcg64 g; int philox; uint64_t seed, n; uint32_t att; } rng_t;

/* Philox4x32-10 (Salmon et al. 2011) and the engine's counter-based stream on top of it (FAST mode 2 of libvgx: output n of
 * (seed, attempt) = low / high 64 bits of the block with counter (n >> 1, attempt, 'VGXs') and the seed's halves as key).
 * Test infrastructure only: the reference has no such stream. */
static void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double rng_uniform(rng_t *r) {
    if (!r->philox) return vgo_pcg64_double(&r->g);
    uint64_t blk = r->n >> 1;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), r->att, 0x56475873u}, key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)}, o[4];
    philox4x32(ctr, key, o);
    uint64_t v = (r->n & 1) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
    r->n += 1;
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}

/* memory_optimization (pyx:105-125): program numbers vs haplotypes.  NH = numToHap[hn], HN = hapToNum[h], CUR = the number of
 * program numbers in use (currentHapNum), ROWN = the length of a row indexed by program number as fastChoose sees it. */
#define NH(hn) (m->memory_optimization ? m->numToHap[(hn)] : (int64_t)(hn))
#define HN(h) (m->memory_optimization ? m->hapToNum[(h)] : (int64_t)(h))
#define CUR (m->memory_optimization ? m->currentHapNum : H_)
#define ROWN (m->memory_optimization ? m->maxHapNum : H_)

static inline void NewInfections(vgo_model *m, int64_t pi, int64_t si, int64_t hi, int64_t num) { /* pyx:246-251 */
    IDX2(m->susceptible, pi, si, S_) -= num;
    m->totalSusceptible[pi] -= num;
    IDX2(m->infectious, pi, hi, H_) += num;
    m->totalInfectious[pi] += num;
    m->globalInfectious += num;
    if (m->sparse) occ_set(m, pi, hi, IDX2(m->infectious, pi, hi, H_) != 0);
}
static inline void NewRecoveries(vgo_model *m, int64_t pi, int64_t si, int64_t hi, int64_t num) { /* pyx:255-260 */
    IDX2(m->susceptible, pi, si, S_) += num;
    m->totalSusceptible[pi] += num;
    IDX2(m->infectious, pi, hi, H_) -= num;
    m->totalInfectious[pi] -= num;
    m->globalInfectious -= num;
    if (m->sparse) {
        int on = IDX2(m->infectious, pi, hi, H_) != 0;
        occ_set(m, pi, hi, on);
        if (!on) IDX2(m->hapPopRate, pi, hi, H_) = 0.0; /* what the dense UpdateRates would store: tEvent*0 */
    }
}

static inline void AddEvent(vgo_model *m, double t, int64_t type, int64_t hap, int64_t pop, int64_t nh, int64_t np) { /* ev:37-44 */
    int64_t p = m->ev_ptr;
    m->ev_times[p] = t;
    m->ev_types[p] = type;
    m->ev_haplotypes[p] = hap;
    m->ev_populations[p] = pop;
    m->ev_newHaplotypes[p] = nh;
    m->ev_newPopulations[p] = np;
    m->ev_ptr += 1;
}

static inline double BirthRate(vgo_model *m, int64_t pi, int64_t hi) { /* pyx:382-392 */
    double ps = 0.0;
    for (int64_t sn = 0; sn < S_; sn++) {
        double x = (double)IDX2(m->susceptible, pi, sn, S_) * IDX2(m->susceptibility, NH(hi), sn, S_);
        IDX3(m->susceptHapPopRate, pi, hi, sn, H_, S_) = x;
        for (int64_t pn = 0; pn < P_; pn++)
            ps += x * IDX2(m->migrationRates, pi, pn, P_) * IDX2(m->migrationRates, pi, pn, P_) *
                  m->contactDensity[pn] / m->actualSizes[pn];
    }
    return m->bRate[NH(hi)] * ps;
}

void vgo_update_all_rates(vgo_model *m) { /* pyx:279-351 */
    for (int64_t sn1 = 0; sn1 < S_; sn1++) {
        m->suscepCumulTransition[sn1] = 0;
        for (int64_t sn2 = 0; sn2 < S_; sn2++) m->suscepCumulTransition[sn1] += IDX2(m->suscepTransition, sn1, sn2, S_);
    }
    for (int64_t pn1 = 0; pn1 < P_; pn1++) {
        IDX2(m->migrationRates, pn1, pn1, P_) = 1.0;
        m->actualSizes[pn1] = 0.0;
        for (int64_t pn2 = 0; pn2 < P_; pn2++) {
            if (pn1 == pn2) continue;
            IDX2(m->migrationRates, pn1, pn1, P_) -= IDX2(m->migrationRates, pn1, pn2, P_);
            m->actualSizes[pn1] += IDX2(m->migrationRates, pn2, pn1, P_) * (double)m->sizes[pn2];
        }
        m->actualSizes[pn1] += IDX2(m->migrationRates, pn1, pn1, P_) * (double)m->sizes[pn1];
    }
    if (m->sparse) occ_rebuild(m);

    m->totalRate = 0.0;
    for (int64_t pn = 0; pn < P_; pn++) { m->infectPopRate[pn] = 0; m->immunePopRate[pn] = 0; m->popRate[pn] = 0.; }
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t hn = 0; hn < CUR; hn++) {
            m->tmRate[hn] = 0;
            for (int64_t s = 0; s < SITES_; s++) m->tmRate[hn] += IDX2(m->mRate, NH(hn), s, SITES_);
            if (m->sparse && IDX2(m->infectious, pn, hn, H_) == 0) {
                /* dead state for an unoccupied haplotype, except the static rates and a zero hapPopRate */
                IDX3(m->eventHapPopRate, pn, hn, 1, H_, 4) = m->dRate[hn];
                IDX3(m->eventHapPopRate, pn, hn, 2, H_, 4) = m->sRate[hn] * m->samplingMultiplier[pn];
                IDX3(m->eventHapPopRate, pn, hn, 3, H_, 4) = m->tmRate[hn];
                IDX2(m->hapPopRate, pn, hn, H_) = 0.0;
                continue;
            }
            IDX3(m->eventHapPopRate, pn, hn, 0, H_, 4) = BirthRate(m, pn, hn);
            IDX3(m->eventHapPopRate, pn, hn, 1, H_, 4) = m->dRate[NH(hn)];
            IDX3(m->eventHapPopRate, pn, hn, 2, H_, 4) = m->sRate[NH(hn)] * m->samplingMultiplier[pn];
            IDX3(m->eventHapPopRate, pn, hn, 3, H_, 4) = m->tmRate[hn];
            IDX2(m->tEventHapPopRate, pn, hn, H_) = 0;
            for (int i = 0; i < 4; i++) IDX2(m->tEventHapPopRate, pn, hn, H_) += IDX3(m->eventHapPopRate, pn, hn, i, H_, 4);
            IDX2(m->hapPopRate, pn, hn, H_) = IDX2(m->tEventHapPopRate, pn, hn, H_) * (double)IDX2(m->infectious, pn, hn, H_);
            m->infectPopRate[pn] += IDX2(m->hapPopRate, pn, hn, H_);
        }
        for (int64_t sn = 0; sn < S_; sn++) {
            IDX2(m->immuneSourcePopRate, pn, sn, S_) = m->suscepCumulTransition[sn] * (double)IDX2(m->susceptible, pn, sn, S_);
            m->immunePopRate[pn] += IDX2(m->immuneSourcePopRate, pn, sn, S_);
        }
        m->popRate[pn] = m->infectPopRate[pn] + m->immunePopRate[pn];
        m->totalRate += m->popRate[pn];
    }

    double *maxEffectiveMigration = (double *)calloc((size_t)P_, sizeof(double));
    for (int64_t pn1 = 0; pn1 < P_; pn1++) {
        for (int64_t pn2 = 0; pn2 < P_; pn2++) {
            if (pn1 == pn2) continue;
            double e = 0.0;
            for (int64_t pn3 = 0; pn3 < P_; pn3++)
                e += IDX2(m->migrationRates, pn1, pn3, P_) * IDX2(m->migrationRates, pn2, pn3, P_) *
                     m->contactDensity[pn3] / m->actualSizes[pn3];
            IDX2(m->effectiveMigration, pn1, pn2, P_) = e;
            if (e > maxEffectiveMigration[pn2]) maxEffectiveMigration[pn2] = e;
        }
    }
    double maxEffectiveBirth = 0.0;
    for (int64_t hn = 0; hn < CUR; hn++)
        for (int64_t sn = 0; sn < S_; sn++)
            if (m->bRate[NH(hn)] * IDX2(m->susceptibility, NH(hn), sn, S_) > maxEffectiveBirth)
                maxEffectiveBirth = m->bRate[NH(hn)] * IDX2(m->susceptibility, NH(hn), sn, S_);

    m->totalMigrationRate = 0.0;
    for (int64_t pn = 0; pn < P_; pn++) {
        m->maxEffectiveBirthMigration[pn] = maxEffectiveMigration[pn] * maxEffectiveBirth;
        m->migPopRate[pn] = m->maxEffectiveBirthMigration[pn] * (double)m->totalSusceptible[pn] *
                            (double)(m->globalInfectious - m->totalInfectious[pn]);
        m->totalMigrationRate += m->migPopRate[pn];
    }
    free(maxEffectiveMigration);
}

static void UpdateRates(vgo_model *m, int64_t pi, int infect, int immune, int migration) { /* pyx:516-546 */
    if (infect) {
        m->infectPopRate[pi] = 0.0;
        int64_t hn = m->sparse ? occ_next(m, pi, 0) : 0;
        while (hn < CUR) {
            double tmp;
            IDX3(m->eventHapPopRate, pi, hn, 0, H_, 4) = BirthRate(m, pi, hn);
            tmp = (IDX3(m->eventHapPopRate, pi, hn, 0, H_, 4) + IDX3(m->eventHapPopRate, pi, hn, 1, H_, 4) +
                   IDX3(m->eventHapPopRate, pi, hn, 2, H_, 4) + IDX3(m->eventHapPopRate, pi, hn, 3, H_, 4));
            IDX2(m->tEventHapPopRate, pi, hn, H_) = tmp;
            IDX2(m->hapPopRate, pi, hn, H_) = IDX2(m->tEventHapPopRate, pi, hn, H_) * (double)IDX2(m->infectious, pi, hn, H_);
            m->infectPopRate[pi] += IDX2(m->hapPopRate, pi, hn, H_);
            hn = m->sparse ? occ_next(m, pi, hn + 1) : hn + 1;
        }
    }
    if (immune) {
        m->immunePopRate[pi] = 0;
        for (int64_t sn = 0; sn < S_; sn++) m->immunePopRate[pi] += IDX2(m->immuneSourcePopRate, pi, sn, S_);
    }
    if (infect || immune) {
        m->popRate[pi] = m->infectPopRate[pi] + m->immunePopRate[pi];
        m->totalRate = 0.0;
        for (int64_t pn = 0; pn < P_; pn++) m->totalRate += m->popRate[pn];
    }
    if (migration) {
        m->totalMigrationRate = 0.0;
        for (int64_t pn = 0; pn < P_; pn++) {
            m->migPopRate[pn] = m->maxEffectiveBirthMigration[pn] * (double)m->totalSusceptible[pn] *
                                (double)(m->globalInfectious - m->totalInfectious[pn]);
            m->totalMigrationRate += m->migPopRate[pn];
        }
    }
}

static void AddLockdown(vgo_model *m, int64_t state, int64_t pi, double t) { /* md:60-66 */
    if (m->loc_n >= m->loc_cap) { m->error = VGO_ERR_LOCKDOWN_LOG_FULL; return; }
    m->loc_states[m->loc_n] = state;
    m->loc_populations[m->loc_n] = pi;
    m->loc_times[m->loc_n] = t;
    m->loc_n += 1;
}

static void CheckLockdown(vgo_model *m, int64_t pi) { /* pyx:698-710 */
    if ((double)m->totalInfectious[pi] > m->startLD[pi] * (double)m->sizes[pi] && m->lockdownON[pi] == 0) {
        m->contactDensity[pi] = m->contactDensityAfterLockdown[pi];
        m->swapLockdown += 1;
        m->lockdownON[pi] = 1;
        vgo_update_all_rates(m);
        AddLockdown(m, 1, pi, m->currentTime);
    }
    if ((double)m->totalInfectious[pi] < m->endLD[pi] * (double)m->sizes[pi] && m->lockdownON[pi] == 1) {
        m->contactDensity[pi] = m->contactDensityBeforeLockdown[pi];
        m->swapLockdown += 1;
        m->lockdownON[pi] = 0;
        vgo_update_all_rates(m);
        AddLockdown(m, 0, pi, m->currentTime);
    }
}

static int64_t Mutate(const vgo_model *m, int64_t hi, int64_t s, int64_t DS) { /* pyx:2420-2427 */
    int64_t digit4 = 1;
    for (int64_t k = 0; k < m->sites - s - 1; k++) digit4 *= 4;
    int64_t AS = (hi / digit4) % 4;
    if (DS >= AS) DS += 1;
    return hi + (DS - AS) * digit4;
}

/* =====================================================================================
 * Direct-method events (pyx:550-694)
 * ===================================================================================== */
static void ImmunityTransition(vgo_model *m, int64_t pi) { /* pyx:550-564 */
    choice_t c = choose_f64(m, &IDX2(m->immuneSourcePopRate, pi, 0, S_), S_, m->immunePopRate[pi], m->rn);
    int64_t ssi = c.i;
    m->rn = c.rn;
    c = choose_f64(m, &IDX2(m->suscepTransition, ssi, 0, S_), S_, m->suscepCumulTransition[ssi], m->rn);
    int64_t tsi = c.i;
    m->rn = c.rn;
    IDX2(m->susceptible, pi, ssi, S_) -= 1;
    IDX2(m->susceptible, pi, tsi, S_) += 1;
    IDX2(m->immuneSourcePopRate, pi, ssi, S_) = (double)IDX2(m->susceptible, pi, ssi, S_) * m->suscepCumulTransition[ssi];
    IDX2(m->immuneSourcePopRate, pi, tsi, S_) = (double)IDX2(m->susceptible, pi, tsi, S_) * m->suscepCumulTransition[tsi];
    UpdateRates(m, pi, 0, 1, 0);
    m->iCounter += 1;
    AddEvent(m, m->currentTime, VGO_SUSCCHANGE, ssi, pi, tsi, 0);
}

/* Python-semantics float modulo for non-negative operands (Cython emits __Pyx_mod_double for `%` on C doubles
 * under cdivision=False; for x >= 0, y > 0 that is fmod). */
static inline double py_mod_nonneg(double x, double y) { return fmod(x, y); }

static void Birth(vgo_model *m, int64_t pi, int64_t hi) { /* pyx:568-605 */
    double ws = 0.0, hs = 0.0;
    for (int64_t sn = 0; sn < S_; sn++) ws += IDX3(m->susceptHapPopRate, pi, hi, sn, H_, S_);
    choice_t c = choose_f64(m, &IDX3(m->susceptHapPopRate, pi, hi, 0, H_, S_), S_, ws, m->rn);
    int64_t si = c.i;
    m->rn = c.rn;
    if (m->rn < m->recombination && m->totalInfectious[pi] > 1) { /* pyx:575-596 */
        m->rn = m->rn / m->recombination;
        /* birthInf[hn] = eventHapPopRate[pi, hn, 0] * infectious[pi, hn] with one host of hi set aside; the
         * reference fills a scratch vector of hapNum doubles (pyx:579-581), restated here without the vector:
         * the same products, summed and scanned in the same order */
        double *birthInf = (double *)malloc((size_t)H_ * sizeof(double));
        IDX2(m->infectious, pi, hi, H_) -= 1;
        for (int64_t hn = 0; hn < H_; hn++) {
            birthInf[hn] = IDX3(m->eventHapPopRate, pi, hn, 0, H_, 4) * (double)IDX2(m->infectious, pi, hn, H_);
            hs += birthInf[hn];
        }
        IDX2(m->infectious, pi, hi, H_) += 1;
        c = choose_f64(m, birthInf, H_, hs, m->rn);
        free(birthInf);
        int64_t hi2 = c.i;
        m->rn = c.rn;
        int64_t posRecomb = (int64_t)((double)m->genome_length * m->rn);
        int64_t nhi = 0;
        for (int64_t s = 0; s < m->sites; s++) {
            /* `4**(sites-s-1)` is a C double (Cython 3, cpow=False); `*` binds before `%` (pyx:589/591), so every
             * site but the last contributes (4^k * floor(h / 4^k)) % 4 == 0 */
            double p4 = pow(4.0, (double)(m->sites - s - 1));
            int64_t parent = m->sitesPosition[s] < posRecomb ? hi : hi2;
            nhi += (int64_t)py_mod_nonneg(p4 * floor((double)parent / p4), 4.0);
        }
        if (m->rec_n < m->rec_cap) {
            m->rec_idevents[m->rec_n] = m->ev_ptr;
            m->rec_his[m->rec_n] = hi;
            m->rec_hi2s[m->rec_n] = hi2;
            m->rec_nhis[m->rec_n] = nhi;
            m->rec_posRecombs[m->rec_n] = posRecomb;
        }
        m->rec_n += 1;
        NewInfections(m, pi, si, nhi, 1);
        AddEvent(m, m->currentTime, VGO_BIRTH, NH(hi), pi, si, NH(hi2));
    } else {
        NewInfections(m, pi, si, hi, 1);
        AddEvent(m, m->currentTime, VGO_BIRTH, NH(hi), pi, si, H_);
    }
    IDX2(m->immuneSourcePopRate, pi, si, S_) = m->suscepCumulTransition[si] * (double)IDX2(m->susceptible, pi, si, S_);
    UpdateRates(m, pi, 1, 1, 1);
    m->bCounter += 1;
}

static void Death(vgo_model *m, int64_t pi, int64_t hi, int add_event) { /* pyx:616-626 */
    int64_t st = m->suscType[NH(hi)];
    NewRecoveries(m, pi, st, hi, 1);
    IDX2(m->immuneSourcePopRate, pi, st, S_) = (double)IDX2(m->susceptible, pi, st, S_) * m->suscepCumulTransition[st];
    UpdateRates(m, pi, 1, 1, 1);
    if (add_event) {
        m->dCounter += 1;
        AddEvent(m, m->currentTime, VGO_DEATH, NH(hi), pi, st, 0);
    }
}

static void Sampling(vgo_model *m, int64_t pi, int64_t hi) { /* pyx:630-635 */
    Death(m, pi, hi, 0);
    m->sCounter += 1;
    AddEvent(m, m->currentTime, VGO_SAMPLING, NH(hi), pi, m->suscType[NH(hi)], 0);
}

/* pyx:264-274.  Upstream grows every array indexed by program number by addMemoryNum columns; here those arrays have hapNum
 * columns from the start, so only the bookkeeping remains. */
static void AddMemory(vgo_model *m) {
    if (m->addMemoryNum + m->maxHapNum > H_) m->addMemoryNum = H_ - m->maxHapNum;
    m->maxHapNum += m->addMemoryNum;
}

/* pyx:355-377: insert haplotype nhi into the sorted table, moving the later program numbers up by one — in the table and in
 * the counts of population pi ONLY (as written upstream: the other populations' counts keep their old program numbers). */
static void AddHaplotype(vgo_model *m, int64_t nhi, int64_t pi) {
    int check = 0;
    int64_t mem_h = 0, mem_inf = 0;
    if (m->currentHapNum == m->maxHapNum && m->maxHapNum < H_) AddMemory(m);
    for (int64_t i = 1; i < m->currentHapNum + 1; i++) {
        if (check) {
            m->hapToNum[mem_h] += 1;
            int64_t t = IDX2(m->infectious, pi, i, H_); IDX2(m->infectious, pi, i, H_) = mem_inf; mem_inf = t;
            t = m->numToHap[i]; m->numToHap[i] = mem_h; mem_h = t;
        }
        if ((m->numToHap[i] > nhi || m->numToHap[i] == 0) && m->numToHap[i - 1] < nhi) {
            mem_h = m->numToHap[i];
            check = 1;
            m->numToHap[i] = nhi;
            mem_inf = IDX2(m->infectious, pi, i, H_);
            IDX2(m->infectious, pi, i, H_) = 0;
            m->hapToNum[nhi] = i;
        }
    }
    m->currentHapNum += 1;
    vgo_update_all_rates(m);
}

static void Mutation(vgo_model *m, int64_t pi, int64_t hi) { /* pyx:640-667 */
    int64_t ohi = NH(hi);
    choice_t c = choose_f64(m, &IDX2(m->mRate, ohi, 0, SITES_), SITES_, m->tmRate[hi], m->rn);
    int64_t mi = c.i;
    m->rn = c.rn;
    const double *hm = &IDX3(m->hapMutType, ohi, mi, 0, SITES_, 3);
    c = choose_f64(m, hm, 3, hm[0] + hm[1] + hm[2], m->rn);
    int64_t DS = c.i;
    m->rn = c.rn;
    int64_t nhi = Mutate(m, ohi, mi, DS);
    if (m->memory_optimization) { /* pyx:651-660 */
        int check = 1;
        for (int64_t hn = 0; hn < m->currentHapNum + 1; hn++) {
            if (hn >= m->maxHapNum) break;   /* (upstream reads one slot past numToHap here when the table is full) */
            if (m->numToHap[hn] == nhi) { check = 0; break; }
        }
        if (check) {
            AddHaplotype(m, nhi, pi);
            if (nhi < ohi) hi += 1;
        }
    }
    IDX2(m->infectious, pi, HN(nhi), H_) += 1;
    IDX2(m->infectious, pi, hi, H_) -= 1;
    if (m->sparse) {
        occ_set(m, pi, nhi, 1);
        int on = IDX2(m->infectious, pi, hi, H_) != 0;
        occ_set(m, pi, hi, on);
        if (!on) IDX2(m->hapPopRate, pi, hi, H_) = 0.0;
    }
    UpdateRates(m, pi, 1, 0, 0);
    m->mCounter += 1;
    AddEvent(m, m->currentTime, VGO_MUTATION, ohi, pi, nhi, 0);
}

static int64_t GenerateMigration(vgo_model *m) { /* pyx:672-694 */
    choice_t c = choose_f64(m, m->migPopRate, P_, m->totalMigrationRate, m->rn);
    int64_t tpi = c.i;
    m->rn = c.rn;
    c = choose_skip_i64(m, m->totalInfectious, P_, m->globalInfectious - m->totalInfectious[tpi], m->rn, tpi);
    int64_t spi = c.i;
    m->rn = c.rn;
    if (m->sparse) c = choose_i64_sparse(m, spi, &IDX2(m->infectious, spi, 0, H_), m->totalInfectious[spi], m->rn);
    else c = choose_i64(m, &IDX2(m->infectious, spi, 0, H_), ROWN, m->totalInfectious[spi], m->rn);
    int64_t hi = c.i;
    m->rn = c.rn;
    c = choose_i64(m, &IDX2(m->susceptible, tpi, 0, S_), S_, m->totalSusceptible[tpi], m->rn);
    int64_t si = c.i;
    m->rn = c.rn;
    double p_accept = IDX2(m->effectiveMigration, spi, tpi, P_) * m->bRate[NH(hi)] * IDX2(m->susceptibility, NH(hi), si, S_) /
                      m->maxEffectiveBirthMigration[tpi];
    if (m->rn < p_accept) {
        NewInfections(m, tpi, si, hi, 1);
        UpdateRates(m, tpi, 1, 1, 1);
        m->migPlus += 1;
        AddEvent(m, m->currentTime, VGO_MIGRATION, NH(hi), spi, si, tpi);
    } else {
        m->migNonPlus += 1;
    }
    return tpi;
}

static int64_t GenerateEvent(vgo_model *m, rng_t *r) { /* pyx:483-512 */
    int64_t pi;
    double choose;
    m->rn = rng_uniform(r);
    choose = m->rn * (m->totalRate + m->totalMigrationRate);
    if (m->totalRate > choose) {
        m->rn = choose / m->totalRate;
        choice_t c = choose_f64(m, m->popRate, P_, m->totalRate, m->rn);
        pi = c.i;
        m->rn = c.rn;
        choose = m->rn * m->popRate[pi];
        if (m->immunePopRate[pi] > choose) {
            m->rn = choose / m->immunePopRate[pi];
            ImmunityTransition(m, pi);
        } else {
            m->rn = (choose - m->immunePopRate[pi]) / m->infectPopRate[pi];
            if (m->sparse) c = choose_f64_sparse(m, pi, &IDX2(m->hapPopRate, pi, 0, H_), m->infectPopRate[pi], m->rn);
            else c = choose_f64(m, &IDX2(m->hapPopRate, pi, 0, H_), ROWN, m->infectPopRate[pi], m->rn);
            int64_t hi = c.i;
            m->rn = c.rn;
            c = choose_f64(m, &IDX3(m->eventHapPopRate, pi, hi, 0, H_, 4), 4, IDX2(m->tEventHapPopRate, pi, hi, H_), m->rn);
            int64_t ei = c.i;
            m->rn = c.rn;
            if (ei == VGO_BIRTH) Birth(m, pi, hi);
            else if (ei == VGO_DEATH) Death(m, pi, hi, 1);
            else if (ei == VGO_SAMPLING) Sampling(m, pi, hi);
            else Mutation(m, pi, hi);
        }
    } else {
        m->rn = (choose - m->totalRate) / m->totalMigrationRate;
        pi = GenerateMigration(m);
    }
    return pi;
}

static void FirstInfection(vgo_model *m) { /* pyx:234-242 */
    if (m->globalInfectious == 0) {
        for (int64_t sn = 0; sn < S_; sn++) {
            if (IDX2(m->susceptible, 0, sn, S_) == 0) continue;
            if (m->memory_optimization) AddHaplotype(m, 0, 0);
            NewInfections(m, 0, sn, 0, 1);
            return;
        }
    }
}

static void PrepareParameters(vgo_model *m) { /* pyx:433-451; events.CreateEvents is done by the caller */
    if (!m->first_simulation) {
        FirstInfection(m);
        m->globalInfectious = 0;
        for (int64_t pn = 0; pn < P_; pn++) {
            m->totalSusceptible[pn] = 0;
            for (int64_t sn = 0; sn < S_; sn++) {
                IDX2(m->initial_susceptible, pn, sn, S_) = IDX2(m->susceptible, pn, sn, S_);
                m->totalSusceptible[pn] += IDX2(m->susceptible, pn, sn, S_);
            }
            m->totalInfectious[pn] = 0;
            for (int64_t hn = 0; hn < H_; hn++) {
                IDX2(m->initial_infectious, pn, hn, H_) = IDX2(m->infectious, pn, hn, H_);
                m->totalInfectious[pn] += IDX2(m->infectious, pn, hn, H_);
                m->globalInfectious += IDX2(m->infectious, pn, hn, H_);
            }
        }
        m->first_simulation = 1;
    }
    if (m->sparse) occ_rebuild(m);
    for (int64_t pn = 0; pn < P_; pn++) CheckLockdown(m, pn);
    vgo_update_all_rates(m);
}

static void Restart(vgo_model *m) { /* pyx:714-738 */
    m->ev_ptr = 0;
    m->mev_ptr = 0;
    m->bCounter = m->dCounter = m->sCounter = m->mCounter = m->iCounter = 0;
    m->migPlus = m->migNonPlus = 0;
    m->currentTime = 0.0;
    m->globalInfectious = 0;
    for (int64_t pn = 0; pn < P_; pn++) {
        m->totalSusceptible[pn] = 0;
        m->totalInfectious[pn] = 0;
        for (int64_t sn = 0; sn < S_; sn++) {
            IDX2(m->susceptible, pn, sn, S_) = IDX2(m->initial_susceptible, pn, sn, S_);
            m->totalSusceptible[pn] += IDX2(m->initial_susceptible, pn, sn, S_);
        }
        for (int64_t hn = 0; hn < H_; hn++) {
            IDX2(m->infectious, pn, hn, H_) = IDX2(m->initial_infectious, pn, hn, H_);
            m->totalInfectious[pn] += IDX2(m->initial_infectious, pn, hn, H_);
            m->globalInfectious += IDX2(m->initial_infectious, pn, hn, H_);
        }
    }
    if (m->sparse) occ_rebuild(m);
    for (int64_t pn = 0; pn < P_; pn++) CheckLockdown(m, pn);
    vgo_update_all_rates(m);
}

int vgo_simulate_direct(vgo_model *m, int64_t iterations, int64_t sample_size, float time, int64_t attempts) { /* pyx:396-429 */
    rng_t r;
    m->error = VGO_OK;
    if (m->memory_optimization && (m->sparse || !m->hapToNum || !m->numToHap)) return (int)(m->error = VGO_ERR_BAD_ARG);
    PrepareParameters(m);
    for (int64_t i = 0; i < attempts; i++) {
        vgo_pcg64_seed(&r.g, (uint64_t)m->user_seed, (uint32_t)i);
        r.philox = (m->log_mode & VGO_RNG_PHILOX) ? 1 : 0; r.seed = (uint64_t)m->user_seed; r.att = (uint32_t)i; r.n = 0;
        if (m->totalRate + m->totalMigrationRate != 0.0 && m->globalInfectious != 0) {
            while (m->ev_ptr < m->ev_size && (sample_size == -1 || m->sCounter <= sample_size) &&
                   (time == -1 || m->currentTime < time)) {
                /* SampleTime pyx:476-478 */
                double u = rng_uniform(&r);
                double lg = ((m->log_mode & 15) == VGO_LOG_PORTABLE) ? vgo_portable_log(u) : log(u);
                double tau = -lg / (m->totalRate + m->totalMigrationRate);
                m->currentTime += tau;
                int64_t pi = GenerateEvent(m, &r);
                m->iterations_done += 1;
                if (m->error) return (int)m->error;
                if (m->totalRate == 0.0 || m->globalInfectious == 0) break;
                CheckLockdown(m, pi);
                if (m->error) return (int)m->error;
            }
        }
        if (m->ev_ptr <= 100 && iterations > 100) {
            Restart(m);
        } else {
            m->good_attempt = i + 1;
            break;
        }
    }
    m->rng_state_hi = r.g.state_hi; m->rng_state_lo = r.g.state_lo; m->rng_inc_hi = r.g.inc_hi; m->rng_inc_lo = r.g.inc_lo;
    return (int)m->error;
}

/* =====================================================================================
 * Tau-leaping (pyx:2293-2593).  The dense per-channel arrays of pyx:210-223 are allocated here
 * for the duration of the call (the reference allocates them in __init__).
 * ===================================================================================== */
int64_t vgo_prop_num(const vgo_model *m) { /* pyx:2301 */
    return m->popNum * ((m->popNum - 1) * m->hapNum * m->susNum + m->susNum * (m->susNum - 1) +
                        m->hapNum * (2 + m->sites * 3 + m->susNum));
}

typedef struct {
    double *pMigr, *pSuscep, *pRec, *pSamp, *pMut, *pTrans;
    int64_t *eMigr, *eSuscep, *eRec, *eSamp, *eMut, *eTrans;
} tau_arrays;

#define MIGR(a, s, t, sn, hn) ((a)[(((s) * P_ + (t)) * S_ + (sn)) * H_ + (hn)])
#define SUSC(a, p, s1, s2) ((a)[((p) * S_ + (s1)) * S_ + (s2)])
#define MUT(a, p, h, s, i) ((a)[(((p) * H_ + (h)) * SITES_ + (s)) * 3 + (i)])
#define TRN(a, p, h, sn) ((a)[((p) * H_ + (h)) * S_ + (sn)])

static void Propensities(vgo_model *m, tau_arrays *A) { /* pyx:2351-2417 */
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t hn = 0; hn < H_; hn++) IDX2(m->infectiousAuxTau, pn, hn, H_) = 0.0;
        for (int64_t sn = 0; sn < S_; sn++) IDX2(m->susceptibleAuxTau, pn, sn, S_) = 0.0;
    }
    for (int64_t spn = 0; spn < P_; spn++)
        for (int64_t tpn = 0; tpn < P_; tpn++) {
            if (spn == tpn) continue;
            for (int64_t sn = 0; sn < S_; sn++)
                for (int64_t hn = 0; hn < H_; hn++) {
                    double a = IDX2(m->effectiveMigration, tpn, spn, P_) * (double)IDX2(m->susceptible, tpn, sn, S_) *
                               (double)IDX2(m->infectious, spn, hn, H_) * m->bRate[hn] *
                               IDX2(m->susceptibility, hn, sn, S_) * IDX2(m->migrationRates, spn, spn, P_);
                    MIGR(A->pMigr, spn, tpn, sn, hn) = a;
                    IDX2(m->infectiousAuxTau, tpn, hn, H_) += a;
                    IDX2(m->susceptibleAuxTau, tpn, sn, S_) -= a;
                }
        }
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t ssn = 0; ssn < S_; ssn++)
            for (int64_t tsn = 0; tsn < S_; tsn++) {
                if (ssn == tsn) continue;
                double a = IDX2(m->suscepTransition, ssn, tsn, S_) * (double)IDX2(m->susceptible, pn, ssn, S_);
                SUSC(A->pSuscep, pn, ssn, tsn) = a;
                IDX2(m->susceptibleAuxTau, pn, tsn, S_) += a;
                IDX2(m->susceptibleAuxTau, pn, ssn, S_) -= a;
            }
        for (int64_t hn = 0; hn < H_; hn++) {
            double a = m->dRate[hn] * (double)IDX2(m->infectious, pn, hn, H_);
            IDX2(A->pRec, pn, hn, H_) = a;
            IDX2(m->susceptibleAuxTau, pn, m->suscType[hn], S_) += a;
            IDX2(m->infectiousAuxTau, pn, hn, H_) -= a;
            a = m->sRate[hn] * (double)IDX2(m->infectious, pn, hn, H_) * m->samplingMultiplier[pn];
            IDX2(A->pSamp, pn, hn, H_) = a;
            IDX2(m->susceptibleAuxTau, pn, m->suscType[hn], S_) += a;
            IDX2(m->infectiousAuxTau, pn, hn, H_) -= a;
            for (int64_t s = 0; s < SITES_; s++)
                for (int64_t i = 0; i < 3; i++) {
                    const double *hm = &IDX3(m->hapMutType, hn, s, 0, SITES_, 3);
                    a = IDX2(m->mRate, hn, s, SITES_) * hm[i] / (hm[0] + hm[1] + hm[2]) * (double)IDX2(m->infectious, pn, hn, H_);
                    MUT(A->pMut, pn, hn, s, i) = a;
                    IDX2(m->infectiousAuxTau, pn, Mutate(m, hn, s, i), H_) += a;
                    IDX2(m->infectiousAuxTau, pn, hn, H_) -= a;
                }
        }
    }
    for (int64_t tpn = 0; tpn < P_; tpn++)
        for (int64_t hn = 0; hn < H_; hn++)
            for (int64_t sn = 0; sn < S_; sn++) {
                double a = 0.0;
                for (int64_t spn = 0; spn < P_; spn++)
                    a += m->bRate[hn] * IDX2(m->susceptibility, hn, sn, S_) * IDX2(m->migrationRates, tpn, spn, P_) *
                         IDX2(m->migrationRates, tpn, spn, P_) * m->contactDensity[spn] *
                         (double)IDX2(m->susceptible, tpn, sn, S_) * (double)IDX2(m->infectious, tpn, hn, H_) /
                         m->actualSizes[spn];
                TRN(A->pTrans, tpn, hn, sn) = a;
                IDX2(m->infectiousAuxTau, tpn, hn, H_) += a;
                IDX2(m->susceptibleAuxTau, tpn, sn, S_) -= a;
            }
}

static void ChooseTau(vgo_model *m) { /* pyx:2432-2450, epsilon is a C float */
    const float epsilon = 0.03f;
    m->tau_l = 1.0;
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t hn = 0; hn < H_; hn++) {
            double d = IDX2(m->infectiousAuxTau, pn, hn, H_);
            if (fabs(d) < 1e-8) continue;
            /* epsilon*X is evaluated in single precision (float * int64 -> float), then /2.0 in double */
            double v = (double)(epsilon * (float)IDX2(m->infectious, pn, hn, H_)) / 2.0;
            double tmp = (v > 1.0 ? v : 1.0) / fabs(d);
            if (tmp < m->tau_l) m->tau_l = tmp;
        }
        for (int64_t sn = 0; sn < S_; sn++) {
            double d = IDX2(m->susceptibleAuxTau, pn, sn, S_);
            if (fabs(d) < 1e-8) continue;
            double v = (double)(epsilon * (float)IDX2(m->susceptible, pn, sn, S_)) / 2.0;
            double tmp = (v > 1.0 ? v : 1.0) / fabs(d);
            if (tmp < m->tau_l) m->tau_l = tmp;
        }
    }
}

static int GenerateEvents_tau(vgo_model *m, tau_arrays *A, rng_t *r) { /* pyx:2454-2529 */
    int64_t en;
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t sn = 0; sn < S_; sn++) IDX2(m->susceptibleDelta, pn, sn, S_) = 0;
        for (int64_t hn = 0; hn < H_; hn++) IDX2(m->infectiousDelta, pn, hn, H_) = 0;
    }
    for (int64_t spn = 0; spn < P_; spn++)
        for (int64_t tpn = 0; tpn < P_; tpn++) {
            if (spn == tpn) continue;
            for (int64_t sn = 0; sn < S_; sn++)
                for (int64_t hn = 0; hn < H_; hn++) {
                    en = vgo_poisson(&r->g, MIGR(A->pMigr, spn, tpn, sn, hn) * m->tau_l);
                    MIGR(A->eMigr, spn, tpn, sn, hn) = en;
                    IDX2(m->infectiousDelta, spn, hn, H_) += en; /* sic: source population (pyx:2473) */
                    IDX2(m->susceptibleDelta, tpn, sn, S_) -= en;
                }
        }
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t ssn = 0; ssn < S_; ssn++)
            for (int64_t tsn = 0; tsn < S_; tsn++) {
                if (ssn == tsn) continue;
                en = vgo_poisson(&r->g, SUSC(A->pSuscep, pn, ssn, tsn) * m->tau_l);
                SUSC(A->eSuscep, pn, ssn, tsn) = en;
                IDX2(m->susceptibleDelta, pn, tsn, S_) += en;
                IDX2(m->susceptibleDelta, pn, ssn, S_) -= en;
            }
        for (int64_t hn = 0; hn < H_; hn++) {
            en = vgo_poisson(&r->g, IDX2(A->pRec, pn, hn, H_) * m->tau_l);
            IDX2(A->eRec, pn, hn, H_) = en;
            IDX2(m->susceptibleDelta, pn, m->suscType[hn], S_) += en;
            IDX2(m->infectiousDelta, pn, hn, H_) -= en;
            en = vgo_poisson(&r->g, IDX2(A->pSamp, pn, hn, H_) * m->tau_l);
            IDX2(A->eSamp, pn, hn, H_) = en;
            IDX2(m->susceptibleDelta, pn, m->suscType[hn], S_) += en;
            IDX2(m->infectiousDelta, pn, hn, H_) -= en;
            for (int64_t s = 0; s < SITES_; s++)
                for (int64_t i = 0; i < 3; i++) {
                    en = vgo_poisson(&r->g, MUT(A->pMut, pn, hn, s, i) * m->tau_l);
                    MUT(A->eMut, pn, hn, s, i) = en;
                    IDX2(m->infectiousDelta, pn, Mutate(m, hn, s, i), H_) += en;
                    IDX2(m->infectiousDelta, pn, hn, H_) -= en;
                }
            for (int64_t sn = 0; sn < S_; sn++) {
                en = vgo_poisson(&r->g, TRN(A->pTrans, pn, hn, sn) * m->tau_l);
                TRN(A->eTrans, pn, hn, sn) = en;
                IDX2(m->infectiousDelta, pn, hn, H_) += en;
                IDX2(m->susceptibleDelta, pn, sn, S_) -= en;
            }
        }
    }
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t sn = 0; sn < S_; sn++) {
            int64_t v = IDX2(m->susceptibleDelta, pn, sn, S_) + IDX2(m->susceptible, pn, sn, S_);
            if (v < 0 || v > m->sizes[pn]) return 0;
        }
        for (int64_t hn = 0; hn < H_; hn++) {
            int64_t v = IDX2(m->infectiousDelta, pn, hn, H_) + IDX2(m->infectious, pn, hn, H_);
            if (v < 0 || v > m->sizes[pn]) return 0;
        }
    }
    return 1;
}

static void AddEvents(vgo_model *m, int64_t num, double t, int64_t type, int64_t hap, int64_t pop, int64_t nh, int64_t np) { /* ev:116-125 */
    if (m->mev_num) {
        if (m->mev_ptr >= m->mev_size) { m->error = VGO_ERR_MULTIEVENTS_FULL; m->mev_ptr += 1; return; }
        int64_t p = m->mev_ptr;
        m->mev_num[p] = num;
        m->mev_times[p] = t;
        m->mev_types[p] = type;
        m->mev_haplotypes[p] = hap;
        m->mev_populations[p] = pop;
        m->mev_newHaplotypes[p] = nh;
        m->mev_newPopulations[p] = np;
    }
    m->mev_ptr += 1;
}

static void UpdateCompartmentCounts_tau(vgo_model *m, tau_arrays *A) { /* pyx:2536-2593 */
    int64_t en;
    for (int64_t spn = 0; spn < P_; spn++)
        for (int64_t tpn = 0; tpn < P_; tpn++) {
            if (spn == tpn) continue;
            for (int64_t sn = 0; sn < S_; sn++)
                for (int64_t hn = 0; hn < H_; hn++) {
                    en = MIGR(A->eMigr, spn, tpn, sn, hn);
                    NewInfections(m, tpn, sn, hn, en);
                    AddEvents(m, en, m->currentTime, VGO_MIGRATION, hn, spn, sn, tpn);
                    m->migPlus += en;
                }
        }
    for (int64_t pn = 0; pn < P_; pn++) {
        for (int64_t ssn = 0; ssn < S_; ssn++)
            for (int64_t tsn = 0; tsn < S_; tsn++) {
                if (ssn == tsn) continue;
                en = SUSC(A->eSuscep, pn, ssn, tsn);
                IDX2(m->susceptible, pn, tsn, S_) += en;
                IDX2(m->susceptible, pn, ssn, S_) -= en;
                AddEvents(m, en, m->currentTime, VGO_SUSCCHANGE, ssn, pn, tsn, 0);
                m->iCounter += en;
            }
        for (int64_t hn = 0; hn < H_; hn++) {
            en = IDX2(A->eRec, pn, hn, H_);
            NewRecoveries(m, pn, m->suscType[hn], hn, en);
            AddEvents(m, en, m->currentTime, VGO_DEATH, hn, pn, m->suscType[hn], 0);
            m->dCounter += en;
            en = IDX2(A->eSamp, pn, hn, H_);
            NewRecoveries(m, pn, m->suscType[hn], hn, en);
            AddEvents(m, en, m->currentTime, VGO_SAMPLING, hn, pn, m->suscType[hn], 0);
            m->sCounter += en;
            for (int64_t s = 0; s < SITES_; s++)
                for (int64_t i = 0; i < 3; i++) {
                    int64_t nhn = Mutate(m, hn, s, i);
                    en = MUT(A->eMut, pn, hn, s, i);
                    IDX2(m->infectious, pn, nhn, H_) += en;
                    IDX2(m->infectious, pn, hn, H_) -= en;
                    AddEvents(m, en, m->currentTime, VGO_MUTATION, hn, pn, nhn, 0);
                    m->mCounter += en;
                }
            for (int64_t sn = 0; sn < S_; sn++) {
                en = TRN(A->eTrans, pn, hn, sn);
                NewInfections(m, pn, sn, hn, en);
                AddEvents(m, en, m->currentTime, VGO_BIRTH, hn, pn, sn, 0);
                m->bCounter += en;
            }
        }
    }
}

/* Test hook (not part of the restatement): rejected tries (halvings of tau_l, pyx:2316-2321) of the first steps of the last
 * vgo_simulate_tau call, so that a test can compare the tau ChooseTau gave a step (leap * 2^tries) with another engine's. */
#define VGO_TAU_TRIES_KEPT 4096
static int64_t g_tau_tries[VGO_TAU_TRIES_KEPT];
static int64_t g_tau_steps = 0;
int64_t vgo_tau_tries(int64_t step) { return (step >= 0 && step < g_tau_steps && step < VGO_TAU_TRIES_KEPT) ? g_tau_tries[step] : -1; }


int vgo_simulate_tau(vgo_model *m, int64_t iterations, int64_t sample_size, float time, int64_t attempts) { /* pyx:2293-2346 */
    rng_t r;
    r.philox = 0; r.seed = 0; r.n = 0; r.att = 0;   /* (tau draws from PCG64 through vgo_poisson) */
    tau_arrays A;
    int64_t sparse_saved = m->sparse;
    m->sparse = 0; /* tau walks every channel; the bitmap is not maintained here */
    m->error = VGO_OK;
    PrepareParameters(m);
    int64_t propNum = vgo_prop_num(m);
    if (m->globalInfectious == 0) FirstInfection(m);
    vgo_update_all_rates(m);

    size_t nMigr = (size_t)(P_ * P_ * S_ * H_), nSus = (size_t)(P_ * S_ * S_), nPH = (size_t)(P_ * H_);
    size_t nMut = (size_t)(P_ * H_ * SITES_ * 3), nTr = (size_t)(P_ * H_ * S_);
    A.pMigr = calloc(nMigr ? nMigr : 1, 8); A.eMigr = calloc(nMigr ? nMigr : 1, 8);
    A.pSuscep = calloc(nSus, 8); A.eSuscep = calloc(nSus, 8);
    A.pRec = calloc(nPH, 8); A.eRec = calloc(nPH, 8);
    g_tau_steps = 0;
    A.pSamp = calloc(nPH, 8); A.eSamp = calloc(nPH, 8);
    A.pMut = calloc(nMut ? nMut : 1, 8); A.eMut = calloc(nMut ? nMut : 1, 8);
    A.pTrans = calloc(nTr, 8); A.eTrans = calloc(nTr, 8);

    for (int64_t i = 0; i < attempts; i++) {
        vgo_pcg64_seed(&r.g, (uint64_t)m->user_seed, (uint32_t)i);
        if (m->totalRate + m->totalMigrationRate != 0.0 && m->globalInfectious != 0) {
            while (m->ev_ptr < m->ev_size && (sample_size == -1 || m->sCounter < sample_size) &&
                   (time == -1 || m->currentTime < time)) {
                Propensities(m, &A);
                ChooseTau(m);
                int64_t halvings = 0;
                while (1) {
                    if (GenerateEvents_tau(m, &A, &r)) break;
                    m->tau_l /= 2;
                    halvings += 1;
                }
                if (g_tau_steps < VGO_TAU_TRIES_KEPT) g_tau_tries[g_tau_steps] = halvings;   /* test hook: vgo_tau_tries */
                g_tau_steps += 1;
                m->currentTime += m->tau_l;
                UpdateCompartmentCounts_tau(m, &A);
                AddEvent(m, m->currentTime, VGO_MULTITYPE, m->mev_ptr - propNum, m->mev_ptr, 0, 0);
                m->iterations_done += 1;
                if (m->error) goto out;
                if (m->globalInfectious == 0) break;
                for (int64_t pn = 0; pn < P_; pn++) CheckLockdown(m, pn);
                if (m->error) goto out;
            }
        }
        if (m->ev_ptr <= 100 && iterations > 100) {
            Restart(m);
        } else {
            m->good_attempt = i + 1;
            break;
        }
    }
out:
    free(A.pMigr); free(A.eMigr); free(A.pSuscep); free(A.eSuscep); free(A.pRec); free(A.eRec);
    free(A.pSamp); free(A.eSamp); free(A.pMut); free(A.eMut); free(A.pTrans); free(A.eTrans);
    m->sparse = sparse_saved;
    m->rng_state_hi = r.g.state_hi; m->rng_state_lo = r.g.state_lo; m->rng_inc_hi = r.g.inc_hi; m->rng_inc_lo = r.g.inc_lo;
    return (int)m->error;
}
