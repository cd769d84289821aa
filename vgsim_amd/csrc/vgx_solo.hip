// vgx_solo.hip — direct Gillespie for ONE trajectory of a SMALL model at a time (hapNum <= 64, popNum <= 128): the latency kernel.
//
// What `Simulator.simulate()` is upstream: one sequential event loop (src/_BirthDeath.pyx:396-429).  A wavefront that runs alone
// issues one instruction every four cycles whatever it is, so the only thing that makes one trajectory faster is FEWER INSTRUCTIONS
// PER EVENT and no memory round trip between them.  This kernel therefore drops everything the ensemble kernels carry for large
// haplotype spaces (occupancy lists, rate classes, cold records in L2) and keeps the reference's DENSE arrays — the whole model —
// in LDS and registers for the entire call:
//   * lane h <-> haplotype h: per-haplotype parameters live in registers; the rate caches of ONE population's row
//     (infectious, eventHapPopRate[.,.,0], tEventHapPopRate, hapPopRate and its serial prefix sums: what fastChoose would form
//     again, fast_choose.pxi:22-28) are registers too, the other populations' rows lie in LDS and are swapped in when an event
//     falls there;
//   * lane p <-> population p (two registers beyond 64): popRate and its serial prefix sums (the partial sums of the totalRate
//     loop, pyx:537-539), infect / immune / migration rates, totals, contact densities, lockdown thresholds;
//   * lane s <-> susceptibility group s of the current population: counts, their copy as of the row's last infect-update (what
//     susceptHapPopRate[pi, hi, :] was built from, pyx:385-386) and immuneSourcePopRate;
//   * every sequential f64 sum of the reference is ONE v_fmac_f64 (DPP row_newbcast source) per term: acc = fma(w[k], m, acc) with
//     m = 1.0 rounds exactly like acc + w[k]; serial PREFIX sums use a per-lane multiplier m = (lane >= k ? 1.0 : 0.0) — lane l
//     adds +0.0 from its own step on (x + 0.0 = x), so it ends with w[0] + ... + w[l] and no select sits on the chain;
//   * BirthRate (pyx:382-392): the terms ((S*sig)*m*m*cd)/as of one (group, susceptibility value) pair are formed once, one
//     population per lane, and every haplotype lane adds them through the same chain with multiplier 1.0 if that value is its
//     own and 0.0 if not: each lane sums exactly the reference's non-zero terms in the reference's order (terms with a zero
//     susceptibility are +0.0 upstream);  x / actualSizes is formed through the correctly rounded reciprocal with two exact
//     residual corrections (div_by_const below: the result IS the correctly rounded quotient);
//   * fastChoose's rescaled random number (fast_choose.pxi:31) is formed by every candidate lane at once and read from the
//     chosen one; the rescaling after the last choice of an event is skipped where nothing reads it (Death, Sampling, and Birth
//     without recombination);
//   * uniforms: 64 PCG64 outputs per refill by lane-parallel jump-ahead into LDS, two per iteration as upstream (pyx:477, 488);
//     the logarithm of SampleTime only in calls that need the device clock (a time limit, trajectories, or no event log: event
//     times are rebuilt on the host with libm from the logged rates, vgx_api.hip host_clock);
//   * event records are staged in LDS and written out 64 at a time (2 KB bursts).
// Everything else (Restart, lockdown switches with UpdateAllRates, migration with its rejection step, mutations, immunity
// transitions, trajectories, the logs the host clock needs) follows vgx_direct.hip / vgx_lanes.hip; start and end state are exchanged
// in their layout (occupancy lists, population blocks), so the host side and the other kernels see no difference.
// Exact mode, no recombination.  One wavefront per replicate: small ensembles run as independent wavefronts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_solo.h"

namespace {

enum { ERR_ZERO_WEIGHT = 3, ERR_CAPACITY = 4, ERR_LOOP_GUARD = 5 };
enum { EV_BIRTH = 0, EV_DEATH, EV_SAMPLING, EV_MUTATION, EV_SUSCCHANGE, EV_MIGRATION };

struct Masks { double m[16]; };   // m[k] = (lane & 15) >= k ? 1.0 : 0.0

// ---- chains ---------------------------------------------------------------------------------------------------------------
// One asm statement per row of 16 steps (safe by construction: the leading s_nop 4 covers both hazards the compiler cannot see
// into an asm statement for — a DPP read of a VGPR written by the previous VALU instruction needs 2 wait states, a DPP
// instruction after a write of EXEC needs 5 — and nothing can be scheduled between the steps); steps run in groups of four, a
// row of n <= 4 / 8 / 12 terms leaves early (entries beyond n MUST hold +0.0: they are added when n is not a multiple of four).
#define SOLO_FM(K, MUL, RM) "v_fmac_f64_dpp %[acc], %[v], %[" MUL "] row_newbcast:" #K " row_mask:" RM " bank_mask:0xf\n\t"
#define SOLO_EXIT(N) "s_cmp_le_i32 %[n], " #N "\n\ts_cbranch_scc1 .Lsolo_done%=\n\t"
#define SOLO_SCAN16(RM)                                                                                                        \
    asm volatile("s_nop 4\n\t" SOLO_FM(0, "m0", RM) SOLO_FM(1, "m1", RM) SOLO_FM(2, "m2", RM) SOLO_FM(3, "m3", RM) SOLO_EXIT(4)  \
                 SOLO_FM(4, "m4", RM) SOLO_FM(5, "m5", RM) SOLO_FM(6, "m6", RM) SOLO_FM(7, "m7", RM) SOLO_EXIT(8)                 \
                 SOLO_FM(8, "m8", RM) SOLO_FM(9, "m9", RM) SOLO_FM(10, "m10", RM) SOLO_FM(11, "m11", RM) SOLO_EXIT(12)            \
                 SOLO_FM(12, "m12", RM) SOLO_FM(13, "m13", RM) SOLO_FM(14, "m14", RM) SOLO_FM(15, "m15", RM)                      \
                 ".Lsolo_done%=:\n\t"                                                                                             \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [m0] "v"(M.m[0]), [m1] "v"(M.m[1]), [m2] "v"(M.m[2]), [m3] "v"(M.m[3]), [m4] "v"(M.m[4]),          \
                   [m5] "v"(M.m[5]), [m6] "v"(M.m[6]), [m7] "v"(M.m[7]), [m8] "v"(M.m[8]), [m9] "v"(M.m[9]), [m10] "v"(M.m[10]),  \
                   [m11] "v"(M.m[11]), [m12] "v"(M.m[12]), [m13] "v"(M.m[13]), [m14] "v"(M.m[14]), [m15] "v"(M.m[15]), [n] "s"(nn) \
                 : "scc")
#define SOLO_SUM16(RM)                                                                                                         \
    asm volatile("s_nop 4\n\t" SOLO_FM(0, "mu", RM) SOLO_FM(1, "mu", RM) SOLO_FM(2, "mu", RM) SOLO_FM(3, "mu", RM) SOLO_EXIT(4)  \
                 SOLO_FM(4, "mu", RM) SOLO_FM(5, "mu", RM) SOLO_FM(6, "mu", RM) SOLO_FM(7, "mu", RM) SOLO_EXIT(8)                 \
                 SOLO_FM(8, "mu", RM) SOLO_FM(9, "mu", RM) SOLO_FM(10, "mu", RM) SOLO_FM(11, "mu", RM) SOLO_EXIT(12)              \
                 SOLO_FM(12, "mu", RM) SOLO_FM(13, "mu", RM) SOLO_FM(14, "mu", RM) SOLO_FM(15, "mu", RM)                          \
                 ".Lsolo_done%=:\n\t"                                                                                             \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [mu] "v"(mu), [n] "s"(nn)                                                                          \
                 : "scc")

// the running sum of row r-1 (its lane 15) moves to the lanes of row r
static __device__ __forceinline__ double row_carry(double acc, int which) {
    int lo = __double2loint(acc), hi = __double2hiint(acc);
    if (which == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x2, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x2, 0xf, false); }
    else if (which == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x4, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x4, 0xf, false); }
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x8, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x8, 0xf, false); }
    return __hiloint2double(hi, lo);
}

// SCAN: lane l < n ends with carry + v[0] + ... + v[l] (the serial prefix); lanes >= n-1 of the last row that ran hold the total.
// SUM: every lane of the last row that ran ends with carry + v[0] + ... + v[n-1].  One value per lane, lanes >= n hold +0.0,
// carry wave-uniform, n wave-uniform (1..64), all lanes active.
template <bool SCAN>
static __device__ __forceinline__ double flat_chain(double v, int n, double carry, const Masks &M) {
    double acc = carry;
    const double mu = 1.0;
    int nn = n;
    if (SCAN) SOLO_SCAN16("0x1"); else SOLO_SUM16("0x1");
    if (n > 16) {
        acc = row_carry(acc, 1); nn = n - 16;
        if (SCAN) SOLO_SCAN16("0x2"); else SOLO_SUM16("0x2");
        if (n > 32) {
            acc = row_carry(acc, 2); nn = n - 32;
            if (SCAN) SOLO_SCAN16("0x4"); else SOLO_SUM16("0x4");
            if (n > 48) {
                acc = row_carry(acc, 3); nn = n - 48;
                if (SCAN) SOLO_SCAN16("0x8"); else SOLO_SUM16("0x8");
            }
        }
    }
    return acc;
}

// every lane: acc += mu * (v[lane 0 of its row] + ... in order ... + v[lane n-1 of its row]) term by term, mu = 1.0 or 0.0 per
// lane; v must hold the same 16 values in every row (nn <= 16, entries beyond nn +0.0)
static __device__ __forceinline__ double rows_chain(double acc, double v, double mu, int nn) {
    SOLO_SUM16("0xf");
    return acc;
}

static __device__ __forceinline__ double bperm_f64(double v, int src_lane) {
    int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// n / b for a divisor b whose correctly rounded reciprocal y = RN(1 / b) is known (actualSizes: a parameter).  q0 = RN(n y) is
// within 2 ulp of n / b; q1 = RN(q0 + r0 y) with r0 = RN(n - q0 b) is within 1 ulp (the residual of a quotient that close is
// formed with a relative error of 2^-53 at most, its product with y corrects q0 to 2^-100 before the rounding); r1 = n - q1 b is then
// exact and q2 = RN(q1 + r1 y) is the correctly rounded quotient (Markstein's theorem on the correction of a faithful quotient
// with a correctly rounded reciprocal: the final step of the Itanium division sequences).  No overflow / underflow in this range
// (rates and host counts).  vgx_test_div_by_const runs the sequence on the device against the division for the tests.
static __device__ __forceinline__ double div_by_const(double n, double b, double y) {
    double q = n * y;
    double r = __builtin_fma(-q, b, n);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-q, b, n);
    return __builtin_fma(r, y, q);
}

static __device__ __forceinline__ bool any_lane(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
static __device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
static __device__ __forceinline__ int64_t uni_i64(int64_t v) {
    int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
static __device__ __forceinline__ double uni_f64(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int NPR>
static __device__ __forceinline__ double pop_get(const double (&v)[NPR], int pi) {
    if (NPR == 1) return bcast(v[0], pi);
    return pi < 64 ? bcast(v[0], pi) : bcast(v[NPR - 1], pi - 64);
}
template <int NPR>
static __device__ __forceinline__ void pop_set(double (&v)[NPR], int pi, double val, int lane) {
#pragma unroll
    for (int r = 0; r < NPR; ++r) v[r] = (lane + 64 * r == pi) ? val : v[r];
}

template <int NPR, bool CLOCK>
struct Solo {
    // ---- shape, pointers ----
    int P, H, S, sites, nseg, lane;
    bool small;           // P <= 16 and H <= 16: the terms of BirthRate already sit in the row of the haplotype lanes
    bool no_imm;          // every suscepCumulTransition is zero: immuneSourcePopRate stays +0.0
    bool exact_rcp;
    const VgxDevParams *p;
    const VgxDevRep *r;
    int64_t rep;
    Masks M;
    // LDS
    double *rowI, *rowCum, *rowHpr, *rowBirth, *rowTE, *susS, *susSt, *susImm, *ldSigma, *ldTrans, *ldMrate, *ldHmt, *ldCd, *ldAs, *ldMig, *ldRng;
    uint32_t *ldStage;
    const double *gMig;   // migrationRates in global memory (when it does not fit LDS)
    double *gEff;         // effectiveMigration of this replicate [P][P]
    // ---- haplotype lanes ----
    double bh, dh, sh, tmh;      // bRate, dRate, sRate, sum of mRate of haplotype `lane`
    int stype;                   // suscType
    uint32_t path;               // bit s: segment s is on this haplotype's path
    // the current row (population `cur`)
    int cur;
    double I, cum, hpr, birth, tE, e1, e2, sm;
    double mrow[NPR];            // migrationRates[cur][lane + 64 r]
    // ---- susceptibility-group lanes of the current row ----
    double Ssus, Sst, imms, cumul_l;
    // ---- segment lanes ----
    double Sseg, sgsig;
    int sgsn;
    // ---- population lanes ----
    double popRate[NPR], cumPop[NPR], infectP[NPR], immuneP[NPR], migR[NPR], maxEBM[NPR], totS[NPR], totI[NPR], cd[NPR], asz[NPR], rcp[NPR],
        thrOn[NPR], thrOff[NPR], smult[NPR];
    int lock[NPR];
    // ---- wave-uniform ----
    double totalRate, totalMig, Rtot, currentTime, gI, rn;
    bool has_mig, ld_any;
    int64_t cB, cD, cS, cM, cI, cSwap, cMigP, cMigN;
    int64_t ev_ptr, loc_n;
    int error;
    // logs
    int record_events;
    int64_t evcap, ev_base;
    double *ev_rate; int32_t *ev_cols;
    int stage_n; int64_t stage_slot0;
    int32_t *loc_rec; double *loc_time; int64_t *loc_iter; int64_t loc_cap;
    double den; int64_t iter_key; int64_t att_loops;
    double *traj; int64_t traj_points, traj_next; double traj_t0, traj_dt;

    __device__ __forceinline__ double mig_at(int i) const { return ldMig ? ldMig[i] : gMig[i]; }

    // ---- row cache -------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ void row_store() {
        if (cur < 0) return;
        if (lane < H) {
            const int o = cur * H + lane;
            rowI[o] = I; rowCum[o] = cum; rowHpr[o] = hpr; rowBirth[o] = birth; rowTE[o] = tE;
        }
        if (lane < S) {
            const int o = cur * S + lane;
            susS[o] = Ssus; susSt[o] = Sst; susImm[o] = imms;
        }
        WSYNC();
    }
    __device__ __forceinline__ void row_load(int pi) {
        I = 0.0; cum = 0.0; hpr = 0.0; birth = 0.0; tE = 0.0;
        if (lane < H) {
            const int o = pi * H + lane;
            I = rowI[o]; cum = rowCum[o]; hpr = rowHpr[o]; birth = rowBirth[o]; tE = rowTE[o];
        }
        Ssus = 0.0; Sst = 0.0; imms = 0.0;
        if (lane < S) {
            const int o = pi * S + lane;
            Ssus = susS[o]; Sst = susSt[o]; imms = susImm[o];
        }
        Sseg = (lane < nseg) ? susS[pi * S + sgsn] : 0.0;
#pragma unroll
        for (int q = 0; q < NPR; ++q) {
            const int pn = lane + 64 * q;
            mrow[q] = pn < P ? mig_at(pi * P + pn) : 0.0;
        }
        sm = sh * pop_get<NPR>(smult, pi);
        e1 = birth + dh;
        e2 = e1 + sm;
        cur = pi;
    }
    __device__ __forceinline__ void row_switch(int pi) {
        if (pi == cur) return;
        row_store();
        row_load(pi);
    }

    // ---- UpdateRates, infect branch, for the current row (pyx:518-528 with BirthRate pyx:382-392); returns infectPopRate ----
    __device__ __forceinline__ double refresh_row() {
        Sst = Ssus;                                 // BirthRate stores susceptHapPopRate = S * sigma (pyx:385-386)
        const double xseg = Sseg * sgsig;           // segment lanes
        double ps = 0.0;
        for (int s = 0; s < nseg; ++s) {
            const double xs = bcast(xseg, s);
            const double mu = (double)((path >> s) & 1u);
            double T[NPR];
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const double t = xs * mrow[q] * mrow[q] * cd[q];
                T[q] = exact_rcp ? div_by_const(t, asz[q], rcp[q]) : t / asz[q];
            }
            if (small) {
                ps = rows_chain(ps, T[0], mu, P);
            } else {
                for (int b = 0; b < P; b += 16) {
                    const double src = bperm_f64(b < 64 ? T[0] : T[NPR - 1], (b & 63) + (lane & 15));
                    ps = rows_chain(ps, src, mu, min(16, P - b));
                }
            }
        }
        birth = bh * ps;
        e1 = birth + dh;
        e2 = e1 + sm;
        tE = e2 + tmh;                              // ((r0 + r1) + r2) + r3, pyx:522-525
        hpr = tE * I;
        cum = flat_chain<true>(hpr, H, 0.0, M);
        return bcast(cum, H - 1);
    }

    // popRate[pi] changed: its serial prefix sums and totalRate (pyx:536-539)
    __device__ __forceinline__ void rescan_pop() {
        cumPop[0] = flat_chain<true>(popRate[0], min(P, 64), 0.0, M);
        if (NPR > 1) {
            const double c = bcast(cumPop[0], 63);
            cumPop[NPR - 1] = flat_chain<true>(popRate[NPR - 1], P - 64, c, M);
            totalRate = bcast(cumPop[NPR - 1], P - 65);
        } else {
            totalRate = bcast(cumPop[0], P - 1);
        }
    }
    // migPopRate of every population and totalMigrationRate (pyx:541-546)
    __device__ __forceinline__ void remig() {
        if (!has_mig) { totalMig = 0.0; return; }
#pragma unroll
        for (int q = 0; q < NPR; ++q) migR[q] = maxEBM[q] * totS[q] * (gI - totI[q]);
        double acc = flat_chain<false>(migR[0], min(P, 64), 0.0, M);
        if (NPR > 1) {
            const double c = bcast(acc, 63);
            acc = flat_chain<false>(migR[NPR - 1], P - 64, c, M);
            totalMig = bcast(acc, P - 65);
        } else {
            totalMig = bcast(acc, P - 1);
        }
    }
    __device__ __forceinline__ double immune_sum() {   // immunePopRate[pi] = 0 + immuneSourcePopRate[pi, 0] + ... (pyx:530-533)
        if (no_imm) return 0.0;
        const double acc = flat_chain<false>(imms, S, 0.0, M);
        return bcast(acc, S - 1);
    }
    // UpdateRates(pi, infect, immune, migration) for pi == cur; inP / imP: the population's current infect / immune rate where
    // the branch that would refresh it is off
    __device__ __forceinline__ void update(int pi, bool infect, bool immune, bool migration, double inP, double imP) {
        if (infect) { inP = refresh_row(); pop_set<NPR>(infectP, pi, inP, lane); }
        if (immune) { imP = immune_sum(); pop_set<NPR>(immuneP, pi, imP, lane); }
        pop_set<NPR>(popRate, pi, inP + imP, lane);
        rescan_pop();
        if (migration) remig();
        Rtot = totalRate + totalMig;
    }

    // ---- UpdateAllRates (pyx:279-351); the parameter-only parts come from the host (vgx_api.hip) ----
    __device__ void rebuild_all() {
        for (int pn = 0; pn < P; ++pn) {
            row_switch(pn);
            imms = cumul_l * Ssus;                          // pyx:319-321
            const double inP = refresh_row();
            const double imP = immune_sum();
            pop_set<NPR>(infectP, pn, inP, lane);
            pop_set<NPR>(immuneP, pn, imP, lane);
            pop_set<NPR>(popRate, pn, inP + imP, lane);
        }
        rescan_pop();
        // effectiveMigration and its column maxima (pyx:327-338): lane <-> pn2, the sums over pn3 run serially in every lane
        double mx[NPR];
#pragma unroll
        for (int q = 0; q < NPR; ++q) mx[q] = 0.0;
        for (int p1 = 0; p1 < P; ++p1) {
            double e[NPR];
#pragma unroll
            for (int q = 0; q < NPR; ++q) e[q] = 0.0;
            for (int p3 = 0; p3 < P; ++p3) {
                const double m13 = mig_at(p1 * P + p3), c3 = ldCd[p3], a3 = ldAs[p3];
#pragma unroll
                for (int q = 0; q < NPR; ++q) {
                    const int p2 = lane + 64 * q;
                    const double m23 = p2 < P ? mig_at(p2 * P + p3) : 0.0;
                    e[q] += m13 * m23 * c3 / a3;
                }
            }
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const int p2 = lane + 64 * q;
                if (p2 < P && p2 != p1) {
                    gEff[p1 * P + p2] = e[q];
                    if (e[q] > mx[q]) mx[q] = e[q];
                }
            }
        }
        bool any = false;
#pragma unroll
        for (int q = 0; q < NPR; ++q) {
            maxEBM[q] = mx[q] * p->maxEffectiveBirth;
            any = any || maxEBM[q] > 0.0;
        }
        has_mig = any_lane(any);
#pragma unroll
        for (int q = 0; q < NPR; ++q) migR[q] = 0.0;
        remig();
        Rtot = totalRate + totalMig;
        WSYNC();
    }

    // ---- event log ---------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ void stage_flush() {
        if (stage_n > 0) {
            WSYNC();
            if (lane < stage_n) {
                const uint32_t *s = ldStage + lane * 8;
                const int64_t slot = stage_slot0 + lane;
                int32_t *c = ev_cols + slot * VGX_EV_COLS;
                c[0] = (int32_t)s[0]; c[1] = (int32_t)s[1]; c[2] = (int32_t)s[2]; c[3] = (int32_t)s[3]; c[4] = (int32_t)s[4]; c[5] = (int32_t)s[5];
                ev_rate[slot] = __hiloint2double((int)s[7], (int)s[6]);
            }
            WSYNC();
        }
        stage_n = 0;
    }
    __device__ __forceinline__ void add_event(int type, int hap, int pop, int nh, int np) {   // events.pxi:37-44
        if (record_events) {
            const int64_t slot = ev_ptr - ev_base;
            if (slot >= 0 && slot < evcap) {
                if (stage_n == 0) stage_slot0 = slot;
                int v = type;
                v = lane == 1 ? hap : v;
                v = lane == 2 ? pop : v;
                v = lane == 3 ? nh : v;
                v = lane == 4 ? np : v;
                v = lane == 5 ? (int)(uint32_t)att_loops : v;
                v = lane == 6 ? __double2loint(den) : v;
                v = lane == 7 ? __double2hiint(den) : v;
                if (lane < 8) ldStage[stage_n * 8 + lane] = (uint32_t)v;
                stage_n += 1;
                if (stage_n == 64) stage_flush();
            } else {
                error = ERR_CAPACITY;
            }
        }
        ev_ptr += 1;
    }

    // ---- CheckLockdown (pyx:698-710) for populations [lo, hi): applies and logs the switches; returns whether any happened ----
    __device__ bool check_lockdowns(int lo, int hi) {
        bool any = false;
        for (int pi = lo; pi < hi; ++pi)
            for (int pass = 0; pass < 2; ++pass) {
                const double ti = pop_get<NPR>(totI, pi);
                const int lk = NPR == 1 ? __builtin_amdgcn_readlane(lock[0], pi)
                                        : (pi < 64 ? __builtin_amdgcn_readlane(lock[0], pi) : __builtin_amdgcn_readlane(lock[NPR - 1], pi - 64));
                const bool flip = pass == 0 ? (ti > pop_get<NPR>(thrOn, pi) && lk == 0) : (ti < pop_get<NPR>(thrOff, pi) && lk == 1);
                if (!any_lane(flip)) continue;
                const double ncd = pass == 0 ? p->cdAfter[pi] : p->cdBefore[pi];
                pop_set<NPR>(cd, pi, ncd, lane);
#pragma unroll
                for (int q = 0; q < NPR; ++q) lock[q] = (lane + 64 * q == pi) ? (pass == 0 ? 1 : 0) : lock[q];
                if (lane == 0) {
                    ldCd[pi] = ncd;
                    if (loc_n < loc_cap) {
                        loc_rec[loc_n * 2 + 0] = pass == 0 ? 1 : 0;
                        loc_rec[loc_n * 2 + 1] = pi;
                        loc_time[loc_n] = currentTime;
                        loc_iter[loc_n] = iter_key;
                    }
                }
                if (loc_n >= loc_cap) error = ERR_CAPACITY;
                cSwap += 1;
                loc_n += 1;
                any = true;
            }
        if (any) WSYNC();
        return any;
    }

    __device__ void traj_emit(double t_new, bool final_fill) {
        while (traj_next < traj_points) {
            const double tg = traj_t0 + (double)traj_next * traj_dt;
            if (!final_fill && !(tg < t_new)) break;
            double *o = traj + traj_next * (int64_t)P * 2;
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const int pn = lane + 64 * q;
                if (pn < P) { o[pn * 2 + 0] = totI[q]; o[pn * 2 + 1] = totS[q]; }
            }
            traj_next += 1;
        }
    }

    // first lane of `hit` or the clamp at n - 1 (fast_choose.pxi:26)
    static __device__ __forceinline__ int first_or_last(unsigned long long hit, int n) {
        return hit ? (int)__builtin_ctzll(hit) : n - 1;
    }

    // ---- GenerateEvent (pyx:483-512); returns the population whose lockdown state has to be checked ----
    __device__ __forceinline__ int generate_event(double u) {
        double choose = u * Rtot;
        if (any_lane(totalRate > choose)) {
            rn = choose / totalRate;
            // fastChoose(popRate, totalRate, rn), fast_choose.pxi:18-31, on the cached serial prefix sums
            const double r2 = totalRate * rn;
            int pi;
            {
                unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < P && !(cumPop[0] < r2));
                if (NPR == 1) {
                    pi = first_or_last(hit, P);
                } else {
                    if (hit) pi = (int)__builtin_ctzll(hit);
                    else {
                        hit = __builtin_amdgcn_ballot_w64(lane + 64 < P && !(cumPop[NPR - 1] < r2));
                        pi = hit ? 64 + (int)__builtin_ctzll(hit) : P - 1;
                    }
                }
            }
            const double W = pop_get<NPR>(popRate, pi), Cm = pop_get<NPR>(cumPop, pi);
            const double IM = pop_get<NPR>(immuneP, pi), IN = pop_get<NPR>(infectP, pi);
            if (any_lane(W == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
            rn = (r2 - (Cm - W)) / W;
            choose = rn * W;
            row_switch(pi);
            if (any_lane(IM > choose)) {
                // ---- ImmunityTransition (pyx:550-564) ----
                rn = choose / IM;
                int ssi, tsi;
                {
                    const double ci = flat_chain<true>(imms, S, 0.0, M);
                    const double r = IM * rn;
                    ssi = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(ci < r)), S);
                    const double w = bcast(imms, ssi), tot = bcast(ci, ssi);
                    if (any_lane(w == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                    rn = (r - (tot - w)) / w;
                }
                {
                    const double tr = lane < S ? ldTrans[ssi * S + lane] : 0.0;
                    const double ct = flat_chain<true>(tr, S, 0.0, M);
                    const double r = bcast(cumul_l, ssi) * rn;
                    tsi = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(ct < r)), S);
                    const double w = bcast(tr, tsi), tot = bcast(ct, tsi);
                    if (any_lane(w == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                    rn = (r - (tot - w)) / w;
                }
                Ssus += (lane == tsi ? 1.0 : 0.0) - (lane == ssi ? 1.0 : 0.0);
                Sseg += (lane < nseg && sgsn == tsi ? 1.0 : 0.0) - (lane < nseg && sgsn == ssi ? 1.0 : 0.0);
                imms = (lane == ssi || lane == tsi) ? Ssus * cumul_l : imms;
                update(pi, false, true, false, IN, 0.0);
                cI += 1;
                add_event(EV_SUSCCHANGE, ssi, pi, tsi, 0);
            } else {
                rn = (choose - IM) / IN;
                // fastChoose(hapPopRate[pi], infectPopRate[pi], rn) on the row's cached prefix sums
                const double r4 = IN * rn;
                const int hi = first_or_last(__builtin_amdgcn_ballot_w64(lane < H && !(cum < r4)), H);
                if (any_lane(lane == hi && hpr == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                const double rn5 = (r4 - (cum - hpr)) / hpr;           // every candidate lane forms its own rescaled number
                // fastChoose(eventHapPopRate[pi, hi, 0..3], tEventHapPopRate[pi, hi], rn): the running totals are e0, e1, e2
                const double r6 = tE * rn5;
                const int eil = (birth < r6 ? 1 : 0) + (e1 < r6 ? 1 : 0) + (e2 < r6 ? 1 : 0);
                const int ei = __builtin_amdgcn_readlane(eil, hi);
                if (ei == 0) {
                    // ---- Birth (pyx:568-605), no recombination ----
                    if (any_lane(lane == hi && birth == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                    int si = 0;
                    {
                        const double x = lane < S ? Sst * ldSigma[hi * S + lane] : 0.0;   // susceptHapPopRate[pi, hi, :]
                        if (S > 1) {
                            const double rn7 = bcast(r6 / birth, hi);                    // (r - (e0 - e0)) / e0
                            const double cx = flat_chain<true>(x, S, 0.0, M);
                            const double r8 = bcast(cx, S - 1) * rn7;
                            si = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(cx < r8)), S);
                        }
                        if (any_lane(lane == si && x == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                    }
                    // NewInfections(pi, si, hi, 1), pyx:246-251
                    Ssus -= (lane == si ? 1.0 : 0.0);
                    Sseg -= (lane < nseg && sgsn == si ? 1.0 : 0.0);
                    I += (lane == hi ? 1.0 : 0.0);
#pragma unroll
                    for (int q = 0; q < NPR; ++q) {
                        const double d1 = (lane + 64 * q == pi) ? 1.0 : 0.0;
                        totS[q] -= d1; totI[q] += d1;
                    }
                    gI += 1.0;
                    add_event(EV_BIRTH, hi, pi, si, H);
                    imms = lane == si ? cumul_l * Ssus : imms;
                    update(pi, true, true, true, 0.0, 0.0);
                    cB += 1;
                } else if (ei == 1 || ei == 2) {
                    // ---- Death / Sampling (pyx:616-635) ----
                    const int st = __builtin_amdgcn_readlane(stype, hi);
                    Ssus += (lane == st ? 1.0 : 0.0);
                    Sseg += (lane < nseg && sgsn == st ? 1.0 : 0.0);
                    I -= (lane == hi ? 1.0 : 0.0);
#pragma unroll
                    for (int q = 0; q < NPR; ++q) {
                        const double d1 = (lane + 64 * q == pi) ? 1.0 : 0.0;
                        totS[q] += d1; totI[q] -= d1;
                    }
                    gI -= 1.0;
                    imms = lane == st ? Ssus * cumul_l : imms;
                    update(pi, true, true, true, 0.0, 0.0);
                    if (ei == 1) { cD += 1; add_event(EV_DEATH, hi, pi, st, 0); }
                    else { cS += 1; add_event(EV_SAMPLING, hi, pi, st, 0); }
                } else {
                    // ---- Mutation (pyx:640-667) ----
                    const double tEh = bcast(tE, hi), tmv = bcast(tmh, hi), r6h = bcast(r6, hi);
                    if (any_lane(tmv == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                    rn = (r6h - (tEh - tmv)) / tmv;
                    int mi, DS;
                    {   // fastChoose(mRate[hi, :], tmRate[hi], rn)
                        const double *w = ldMrate + hi * sites;
                        const double r = tmv * rn;
                        int i = 0;
                        double total = w[0];
                        while (any_lane(total < r) && i < sites - 1) { i += 1; total += w[i]; }
                        const double wi = w[i];
                        if (any_lane(wi == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                        rn = (r - (total - wi)) / wi;
                        mi = i;
                    }
                    {   // fastChoose(hapMutType[hi, mi, :], their sum, rn)
                        const double *w = ldHmt + (hi * sites + mi) * 3;
                        const double r = (w[0] + w[1] + w[2]) * rn;
                        int i = 0;
                        double total = w[0];
                        while (any_lane(total < r) && i < 2) { i += 1; total += w[i]; }
                        const double wi = w[i];
                        if (any_lane(wi == 0.0)) { error = ERR_ZERO_WEIGHT; return pi; }
                        rn = (r - (total - wi)) / wi;
                        DS = i;
                    }
                    const int digit4 = 1 << (2 * (sites - mi - 1));      // Mutate, pyx:2420-2427
                    const int AS = (hi / digit4) % 4;
                    if (DS >= AS) DS += 1;
                    const int nhi = hi + (DS - AS) * digit4;
                    I += (lane == nhi ? 1.0 : 0.0) - (lane == hi ? 1.0 : 0.0);
                    update(pi, true, false, false, 0.0, IM);
                    cM += 1;
                    add_event(EV_MUTATION, hi, pi, nhi, 0);
                }
            }
            return pi;
        }
        // ---- GenerateMigration (pyx:672-694) ----
        rn = (choose - totalRate) / totalMig;
        int tpi;
        {   // fastChoose(migPopRate, totalMigrationRate, rn)
            double cm[NPR];
            cm[0] = flat_chain<true>(migR[0], min(P, 64), 0.0, M);
            if (NPR > 1) cm[NPR - 1] = flat_chain<true>(migR[NPR - 1], P - 64, bcast(cm[0], 63), M);
            const double r = totalMig * rn;
            unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < P && !(cm[0] < r));
            if (NPR == 1) tpi = first_or_last(hit, P);
            else if (hit) tpi = (int)__builtin_ctzll(hit);
            else { hit = __builtin_amdgcn_ballot_w64(lane + 64 < P && !(cm[NPR - 1] < r)); tpi = hit ? 64 + (int)__builtin_ctzll(hit) : P - 1; }
            const double w = pop_get<NPR>(migR, tpi), tot = pop_get<NPR>(cm, tpi);
            if (any_lane(w == 0.0)) { error = ERR_ZERO_WEIGHT; return tpi; }
            rn = (r - (tot - w)) / w;
        }
        int spi;
        {   // fastChoose_skip(totalInfectious, globalInfectious - totalInfectious[tpi], rn, tpi), fast_choose.pxi:36-52 (integer
            // weights: whole numbers below 2^53 add exactly in any order)
            double wv[NPR], cs[NPR];
#pragma unroll
            for (int q = 0; q < NPR; ++q) wv[q] = (lane + 64 * q == tpi || lane + 64 * q >= P) ? 0.0 : totI[q];
            cs[0] = flat_chain<true>(wv[0], min(P, 64), 0.0, M);
            if (NPR > 1) cs[NPR - 1] = flat_chain<true>(wv[NPR - 1], P - 64, bcast(cs[0], 63), M);
            const double r = (gI - pop_get<NPR>(totI, tpi)) * rn;
            unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < P && lane != tpi && !(cs[0] < r));
            if (hit) spi = (int)__builtin_ctzll(hit);
            else if (NPR == 1) spi = P - 1;
            else { hit = __builtin_amdgcn_ballot_w64(lane + 64 < P && lane + 64 != tpi && !(cs[NPR - 1] < r)); spi = hit ? 64 + (int)__builtin_ctzll(hit) : P - 1; }
            const double w = pop_get<NPR>(totI, spi), tot = pop_get<NPR>(cs, spi);
            if (any_lane(w == 0.0)) { error = ERR_ZERO_WEIGHT; return tpi; }
            // running total at the stop: every weight up to spi except the skipped one (also at a clamp on the skipped index, where
            // upstream's total does not hold the weight it then subtracts)
            rn = (r - (tot - w)) / w;
        }
        int hi;
        double b_hi;
        {   // fastChoose(infectious[spi], totalInfectious[spi], rn)
            row_switch(spi);
            const double ci = flat_chain<true>(I, H, 0.0, M);
            const double r = pop_get<NPR>(totI, spi) * rn;
            hi = first_or_last(__builtin_amdgcn_ballot_w64(lane < H && !(ci < r)), H);
            const double w = bcast(I, hi), tot = bcast(ci, hi);
            if (any_lane(w == 0.0)) { error = ERR_ZERO_WEIGHT; return tpi; }
            rn = (r - (tot - w)) / w;
            b_hi = bcast(bh, hi);
        }
        int si;
        {   // fastChoose(susceptible[tpi], totalSusceptible[tpi], rn)
            row_switch(tpi);
            const double cs = flat_chain<true>(Ssus, S, 0.0, M);
            const double r = pop_get<NPR>(totS, tpi) * rn;
            si = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(cs < r)), S);
            const double w = bcast(Ssus, si), tot = bcast(cs, si);
            if (any_lane(w == 0.0)) { error = ERR_ZERO_WEIGHT; return tpi; }
            rn = (r - (tot - w)) / w;
        }
        const double p_accept = gEff[spi * P + tpi] * b_hi * ldSigma[hi * S + si] / pop_get<NPR>(maxEBM, tpi);
        if (any_lane(rn < p_accept)) {
            Ssus -= (lane == si ? 1.0 : 0.0);
            Sseg -= (lane < nseg && sgsn == si ? 1.0 : 0.0);
            I += (lane == hi ? 1.0 : 0.0);
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const double d1 = (lane + 64 * q == tpi) ? 1.0 : 0.0;
                totS[q] -= d1; totI[q] += d1;
            }
            gI += 1.0;
            update(tpi, true, true, true, 0.0, 0.0);
            cMigP += 1;
            add_event(EV_MIGRATION, hi, spi, si, tpi);
        } else {
            cMigN += 1;
        }
        return tpi;
    }
};

struct SoloRng {
    uint64_t Ah, Al, Gh, Gl;   // per lane: a^(lane+1), sum_{j<=lane} a^j
    uint64_t sh, sl, ih, il;   // stream position before the batch; increment
    int pos;                   // iterations consumed from the current batch (32 = empty)
};
static __device__ void solo_rng_init(SoloRng &g, int lane) {
    const uint64_t MH = 0x2360ED051FC65DA4ull, ML = 0x4385DF649FCCF645ull;
    uint64_t Ah = MH, Al = ML, Gh = 0, Gl = 1;
    for (int j = 1; j < 64; ++j) {
        uint64_t nh, nl, gh, gl;
        vgx_mul128(Ah, Al, MH, ML, nh, nl);
        vgx_mul128(Gh, Gl, MH, ML, gh, gl);
        vgx_add128(gh, gl, 0, 1);
        if (j <= lane) { Ah = nh; Al = nl; Gh = gh; Gl = gl; }
    }
    g.Ah = Ah; g.Al = Al; g.Gh = Gh; g.Gl = Gl;
    g.pos = 32;
}

template <int NPR, bool CLOCK>
static __device__ __forceinline__ void solo_body(const VgxDirectArgs &a, const VgxSoloArgs &sa) {
    const int64_t rep = blockIdx.x;
    if (rep >= a.n_replicates) return;
    const int lane = threadIdx.x;
    const VgxDevParams &p = a.p;
    const VgxDevRep &r = a.r;
    const int P = p.P, H = p.H, S = p.S, sites = p.sites;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgxSoloLayout L = vgx_solo_layout(P, H, S, sites, sa.mig_in_lds);

    Solo<NPR, CLOCK> c;
    c.P = P; c.H = H; c.S = S; c.sites = sites; c.nseg = sa.nseg; c.lane = lane;
    c.small = P <= 16 && H <= 16;
    c.exact_rcp = sa.exact_rcp_div != 0;
    c.p = &a.p; c.r = &a.r; c.rep = rep;
#pragma unroll
    for (int k = 0; k < 16; ++k) c.M.m[k] = (lane & 15) >= k ? 1.0 : 0.0;
    c.ldRng = (double *)(smem + L.rng); c.ldStage = (uint32_t *)(smem + L.stage);
    c.rowI = (double *)(smem + L.rowI); c.rowCum = (double *)(smem + L.rowCum); c.rowHpr = (double *)(smem + L.rowHpr);
    c.rowBirth = (double *)(smem + L.rowBirth); c.rowTE = (double *)(smem + L.rowTE);
    c.susS = (double *)(smem + L.susS); c.susSt = (double *)(smem + L.susSt); c.susImm = (double *)(smem + L.susImm);
    c.ldSigma = (double *)(smem + L.sigma); c.ldTrans = (double *)(smem + L.trans);
    c.ldMrate = (double *)(smem + L.mrate); c.ldHmt = (double *)(smem + L.hmt);
    c.ldCd = (double *)(smem + L.cd); c.ldAs = (double *)(smem + L.as);
    c.ldMig = L.mig >= 0 ? (double *)(smem + L.mig) : nullptr;
    c.gMig = p.mig;
    c.gEff = r.effMig + rep * P * P;

    // ---- parameters ----
    c.bh = 0.0; c.dh = 0.0; c.sh = 0.0; c.tmh = 0.0; c.stype = 0; c.path = 0u;
    if (lane < H) {
        const int cl = p.cls[lane];
        c.bh = p.bRate[lane]; c.dh = p.c_d[cl]; c.sh = p.c_s[cl]; c.tmh = p.c_tm[cl];
        c.stype = (int)p.suscType[lane];
        for (int s = 0; s < sa.nseg; ++s)
            if (p.susc[lane * S + sa.seg_sn[s]] == sa.seg_sig[s]) c.path |= 1u << s;
    }
    c.sgsn = 0; c.sgsig = 0.0;
    if (lane < sa.nseg) { c.sgsn = sa.seg_sn[lane]; c.sgsig = sa.seg_sig[lane]; }
    c.cumul_l = lane < S ? p.suscepCumul[lane] : 0.0;
    c.no_imm = !any_lane(c.cumul_l != 0.0);
    for (int i = lane; i < H * S; i += 64) c.ldSigma[i] = p.susc[i];
    for (int i = lane; i < S * S; i += 64) c.ldTrans[i] = p.suscepTransition[i];
    for (int i = lane; i < H * sites; i += 64) c.ldMrate[i] = p.mRate[i];
    for (int i = lane; i < H * sites * 3; i += 64) c.ldHmt[i] = p.hapMutType[i];
    if (c.ldMig)
        for (int i = lane; i < P * P; i += 64) c.ldMig[i] = p.mig[i];

    // ---- start state from the layout of the other direct kernels (vgx_dev.h) ----
    double *gD = r.popD + rep * PD_COUNT * P;
    int64_t *gI = r.popI + rep * PI_COUNT * P;
    int32_t *gN = r.nocc + rep * P;
    for (int i = lane; i < P * H; i += 64) { c.rowI[i] = 0.0; c.rowCum[i] = 0.0; c.rowHpr[i] = 0.0; c.rowBirth[i] = 0.0; c.rowTE[i] = 0.0; }
    WSYNC();
    for (int pn = 0; pn < P; ++pn) {
        const int n = gN[pn];
        const int32_t *lh = r.lhap + (rep * P + pn) * r.cap;
        const int64_t *ln = r.lcnt + (rep * P + pn) * r.cap;
        for (int k = lane; k < n; k += 64) c.rowI[pn * H + lh[k]] = (double)ln[k];
    }
    for (int i = lane; i < P * S; i += 64) {
        const double v = (double)r.sus[rep * P * S + i];
        c.susS[i] = v; c.susSt[i] = v; c.susImm[i] = 0.0;
    }
    bool ldp = false;
#pragma unroll
    for (int q = 0; q < NPR; ++q) {
        const int pn = lane + 64 * q;
        const bool in = pn < P;
        c.popRate[q] = 0.0; c.cumPop[q] = 0.0; c.infectP[q] = 0.0; c.immuneP[q] = 0.0; c.migR[q] = 0.0; c.maxEBM[q] = 0.0;
        c.cd[q] = in ? gD[PD_CD * P + pn] : 0.0;
        c.asz[q] = in ? p.actualSizes[pn] : 1.0;
        c.rcp[q] = in ? sa.rcpAs[pn] : 1.0;
        c.smult[q] = in ? p.sampMult[pn] : 0.0;
        c.totS[q] = in ? (double)gI[PI_TOTSUS * P + pn] : 0.0;
        c.totI[q] = in ? (double)gI[PI_TOTINF * P + pn] : 0.0;
        c.lock[q] = in ? (int)gI[PI_LOCK * P + pn] : 0;
        const double sz = in ? (double)p.sizes[pn] : 0.0;
        c.thrOn[q] = in ? p.startLD[pn] * sz : 0.0;
        c.thrOff[q] = in ? p.endLD[pn] * sz : 0.0;
        // a population can switch on only if its threshold lies below its size, off only if it is on
        ldp = ldp || (in && (c.thrOn[q] < sz || c.lock[q] != 0));
        if (in) { c.ldCd[pn] = c.cd[q]; c.ldAs[pn] = c.asz[q]; }
    }
    c.ld_any = any_lane(ldp);
    WSYNC();

    VgxRepScalars *sc = r.sc + rep;
    c.currentTime = sc->currentTime; c.totalRate = 0.0; c.totalMig = 0.0; c.Rtot = 0.0; c.rn = 0.0;
    c.gI = (double)uni_i64(sc->globalInfectious);
    c.cB = uni_i64(sc->bCounter); c.cD = uni_i64(sc->dCounter); c.cS = uni_i64(sc->sCounter); c.cM = uni_i64(sc->mCounter);
    c.cI = uni_i64(sc->iCounter); c.cSwap = uni_i64(sc->swapLockdown); c.cMigP = uni_i64(sc->migPlus); c.cMigN = uni_i64(sc->migNonPlus);
    c.ev_ptr = uni_i64(sc->ev_ptr); c.loc_n = 0; c.error = 0;
    c.record_events = a.record_events;
    c.evcap = r.evcap; c.ev_base = r.ev_base;
    c.ev_rate = r.ev_rate + rep * r.evcap;
    c.ev_cols = r.ev_cols + rep * r.evcap * VGX_EV_COLS;
    c.stage_n = 0; c.stage_slot0 = 0;
    c.loc_cap = r.loc_cap;
    c.loc_rec = r.loc_rec + rep * r.loc_cap * 2;
    c.loc_time = r.loc_time + rep * r.loc_cap;
    c.loc_iter = r.loc_iter + rep * r.loc_cap;
    c.den = 0.0; c.iter_key = 0; c.att_loops = 0;
    c.traj_points = r.traj_points; c.traj_t0 = r.traj_t0; c.traj_dt = r.traj_dt; c.traj_next = 0;
    c.traj = r.traj ? r.traj + rep * r.traj_points * P * 2 : nullptr;
    c.cur = -1;
    c.has_mig = true;
    c.I = 0.0; c.cum = 0.0; c.hpr = 0.0; c.birth = 0.0; c.tE = 0.0; c.e1 = 0.0; c.e2 = 0.0; c.sm = 0.0;
    c.Ssus = 0.0; c.Sst = 0.0; c.imms = 0.0; c.Sseg = 0.0;
#pragma unroll
    for (int q = 0; q < NPR; ++q) c.mrow[q] = 0.0;

    const double tlimit = (double)a.time;
    const bool has_tlimit = !(a.time == -1.0f);
    const int64_t seed = r.seeds[rep];
    int64_t loops = 0, restarts = 0, good_attempt = sc->good_attempt, last_att = -1;
    int64_t att_ev0 = c.ev_ptr, att_loc0 = 0, fa_n = 0;   // first log index / lockdown record of the current attempt
    SoloRng g;
    solo_rng_init(g, lane);
    g.sh = g.sl = g.ih = g.il = 0;

    // PrepareParameters tail (pyx:449-451): CheckLockdown for every population, UpdateAllRates
    c.iter_key = 0;
    if (c.ld_any) c.check_lockdowns(0, P);
    c.rebuild_all();

    for (int64_t att = 0; att < a.attempts && !c.error; ++att) {   // pyx:399-418
        {
            VgxPcg64 s;
            vgx_pcg64_seed(s, (uint64_t)seed, (uint32_t)att);
            g.sh = s.sh; g.sl = s.sl; g.ih = s.ih; g.il = s.il;
            g.pos = 32;
        }
        last_att = att; c.att_loops = 0;
        if (any_lane(c.Rtot != 0.0) && any_lane(c.gI != 0.0)) {
            while (c.ev_ptr < a.ev_size && (a.sample_size == -1 || c.cS <= a.sample_size) && (!has_tlimit || any_lane(c.currentTime < tlimit))) {
                if (loops >= a.max_loop) { c.error = ERR_LOOP_GUARD; break; }
                loops += 1;
                c.att_loops += 1;
                if (g.pos == 32) {
                    // 64 PCG64 outputs: lane k jumps the stream k + 1 steps ahead (exact 128-bit arithmetic); even outputs are the
                    // uniforms of SampleTime (pyx:477), odd ones those of GenerateEvent (pyx:488)
                    uint64_t h, l, ch, cl;
                    vgx_mul128(g.Ah, g.Al, g.sh, g.sl, h, l);
                    vgx_mul128(g.Gh, g.Gl, g.ih, g.il, ch, cl);
                    vgx_add128(h, l, ch, cl);
                    const double u = vgx_pcg64_output_double(h, l);
                    WSYNC();
                    c.ldRng[lane] = (CLOCK && !(lane & 1)) ? -vgx_log(u) : u;
                    g.sh = (uint64_t)bcast_i64((int64_t)h, 63);
                    g.sl = (uint64_t)bcast_i64((int64_t)l, 63);
                    g.pos = 0;
                    WSYNC();
                }
                const double u2 = c.ldRng[2 * g.pos + 1];
                c.den = c.Rtot;
                c.iter_key = (att << 40) | c.att_loops;
                if (CLOCK) {
                    const double nlog = c.ldRng[2 * g.pos];
                    const double t_new = c.currentTime + (nlog / c.den);   // SampleTime pyx:476-478
                    if (c.traj) c.traj_emit(t_new, false);
                    c.currentTime = t_new;
                }
                g.pos += 1;
                const int pi = c.generate_event(u2);
                if (c.error) break;
                if (any_lane(c.totalRate == 0.0) || any_lane(c.gI == 0.0)) break;   // pyx:410-411
                if (c.ld_any && c.check_lockdowns(pi, pi + 1)) c.rebuild_all();    // pyx:412
                if (c.error) break;
            }
        }
        if (c.error) break;
        c.stage_flush();
        if (c.ev_ptr <= 100 && a.iterations > 100) {
            // Restart (pyx:714-738); swapLockdown survives.  Lockdown records of the failed attempt stay in the log: keep the
            // (rate, iteration) pairs the host clock needs for them
            if (c.loc_n > att_loc0 && c.record_events && r.fa_cap > 0) {
                const int64_t n = c.ev_ptr - att_ev0;
                for (int64_t k = lane; k < n; k += 64) {
                    const int64_t slot = att_ev0 + k - c.ev_base;
                    if (fa_n + k < r.fa_cap && slot >= 0 && slot < c.evcap) {
                        r.fa_rate[rep * r.fa_cap + fa_n + k] = c.ev_rate[slot];
                        r.fa_key[rep * r.fa_cap + fa_n + k] = (att << 40) | (int64_t)(uint32_t)c.ev_cols[slot * VGX_EV_COLS + 5];
                    }
                }
                fa_n += n;
                WSYNC();
            }
            att_ev0 = 0;
            c.iter_key = (att + 1) << 40;   // the CheckLockdown below belongs to the next attempt, before its first iteration
            c.ev_ptr = 0;
            c.cB = c.cD = c.cS = c.cM = c.cI = 0; c.cMigP = c.cMigN = 0;
            c.currentTime = 0.0;
            c.traj_next = 0;
            // compartments back to the initial snapshot
            c.cur = -1;
            WSYNC();
            for (int i = lane; i < P * H; i += 64) c.rowI[i] = 0.0;
            WSYNC();
            double g_all = 0.0;
            for (int pn = 0; pn < P; ++pn) {
                const int n = r.i_nocc[pn];
                double ti = 0.0, ts = 0.0;
                for (int k = 0; k < n; ++k) {   // (lists of the initial state are short: one index case upstream)
                    const double v = (double)r.i_cnt[(int64_t)pn * r.i_cap + k];
                    if (lane == 0) c.rowI[pn * H + r.i_hap[(int64_t)pn * r.i_cap + k]] = v;
                    ti += v;
                }
                for (int sn = 0; sn < S; ++sn) {
                    const double v = (double)r.i_sus[pn * S + sn];
                    if (lane == 0) c.susS[pn * S + sn] = v;
                    ts += v;
                }
                pop_set<NPR>(c.totI, pn, ti, lane);
                pop_set<NPR>(c.totS, pn, ts, lane);
                g_all += ti;
            }
            c.gI = g_all;
            WSYNC();
            restarts += 1;
            att_loc0 = c.loc_n;
            if (c.ld_any) c.check_lockdowns(0, P);
            c.rebuild_all();
        } else {
            good_attempt = att + 1;
            break;
        }
    }
    c.stage_flush();
    if (c.traj) c.traj_emit(0.0, true);

    // ---- end state back in the layout of the other direct kernels ----
    c.row_store();
    for (int pn = 0; pn < P; ++pn) {
        int32_t *lh = r.lhap + (rep * P + pn) * r.cap, *lc = r.lcls + (rep * P + pn) * r.cap;
        int64_t *ln = r.lcnt + (rep * P + pn) * r.cap, *lt = r.ltsum + (rep * P + pn) * r.capT;
        const double v = lane < H ? c.rowI[pn * H + lane] : 0.0;
        const unsigned long long nz = __builtin_amdgcn_ballot_w64(v != 0.0);
        const int n = __builtin_popcountll(nz);
        const int pos = __builtin_popcountll(nz & ((1ull << lane) - 1ull));
        if (n > r.cap) c.error = ERR_CAPACITY;
        if (v != 0.0 && pos < r.cap) { lh[pos] = lane; lc[pos] = p.cls[lane]; ln[pos] = (int64_t)v; }
        for (int j = lane; j < r.capT; j += 64) lt[j] = j == 0 ? (int64_t)pop_get<NPR>(c.totI, pn) : 0;
        if (lane == 0) gN[pn] = n < r.cap ? n : (int)r.cap;
    }
#pragma unroll
    for (int q = 0; q < NPR; ++q) {
        const int pn = lane + 64 * q;
        if (pn < P) {
            gD[PD_POPRATE * P + pn] = c.popRate[q];
            gD[PD_INFECT * P + pn] = c.infectP[q];
            gD[PD_IMMUNE * P + pn] = c.immuneP[q];
            gD[PD_MIG * P + pn] = c.migR[q];
            gD[PD_MAXEBM * P + pn] = c.maxEBM[q];
            gD[PD_CD * P + pn] = c.cd[q];
            gI[PI_TOTSUS * P + pn] = (int64_t)c.totS[q];
            gI[PI_TOTINF * P + pn] = (int64_t)c.totI[q];
            gI[PI_LOCK * P + pn] = c.lock[q];
        }
    }
    for (int i = lane; i < P * S; i += 64) {
        r.sus[rep * P * S + i] = (int64_t)c.susS[i];
        r.immSrc[rep * P * S + i] = c.susImm[i];
    }
    if (lane == 0) {
        sc->currentTime = c.currentTime; sc->totalRate = c.totalRate; sc->totalMig = c.totalMig;
        sc->globalInfectious = (int64_t)c.gI;
        sc->bCounter = c.cB; sc->dCounter = c.cD; sc->sCounter = c.cS; sc->mCounter = c.cM; sc->iCounter = c.cI;
        sc->swapLockdown = c.cSwap; sc->migPlus = c.cMigP; sc->migNonPlus = c.cMigN;
        sc->good_attempt = good_attempt;
        sc->ev_ptr = c.ev_ptr; sc->loop_iterations = loops; sc->restarts = restarts;
        sc->loc_n = c.loc_n; sc->error = c.error; sc->traj_next = c.traj_next;
        sc->last_attempt = last_att; sc->last_attempt_loops = c.att_loops;
        sc->rec_n = 0;
        sc->fa_n = fa_n;
    }
}

}  // namespace

extern "C" __global__ void __launch_bounds__(64) vgx_solo_kernel_p64(VgxDirectArgs a, VgxSoloArgs sa) { solo_body<1, false>(a, sa); }
extern "C" __global__ void __launch_bounds__(64) vgx_solo_kernel_p64_clock(VgxDirectArgs a, VgxSoloArgs sa) { solo_body<1, true>(a, sa); }
extern "C" __global__ void __launch_bounds__(64) vgx_solo_kernel_p128(VgxDirectArgs a, VgxSoloArgs sa) { solo_body<2, false>(a, sa); }
extern "C" __global__ void __launch_bounds__(64) vgx_solo_kernel_p128_clock(VgxDirectArgs a, VgxSoloArgs sa) { solo_body<2, true>(a, sa); }

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_solo(const VgxDirectArgs *a, const VgxSoloArgs *sa, int clock,
                                                                            hipStream_t stream) {
    const VgxSoloLayout L = vgx_solo_layout(a->p.P, a->p.H, a->p.S, a->p.sites, sa->mig_in_lds);
    void (*k)(VgxDirectArgs, VgxSoloArgs) =
        a->p.P <= 64 ? (clock ? vgx_solo_kernel_p64_clock : vgx_solo_kernel_p64) : (clock ? vgx_solo_kernel_p128_clock : vgx_solo_kernel_p128);
    hipError_t err = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k, dim3((unsigned)a->n_replicates), dim3(64), (size_t)L.total, stream, *a, *sa);
    return hipGetLastError();
}

// ---- test hook: the reciprocal division of BirthRate's terms against the division, on the device ----
extern "C" __global__ void vgx_solo_divtest_kernel(const double *n, const double *b, double *q_seq, double *q_div, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double y = 1.0 / b[i];
    q_seq[i] = div_by_const(n[i], b[i], y);
    q_div[i] = n[i] / b[i];
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_solo_divtest(const double *n, const double *b, double *q_seq, double *q_div,
                                                                                    int64_t count, hipStream_t stream) {
    hipLaunchKernelGGL(vgx_solo_divtest_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, n, b, q_seq, q_div, count);
    return hipGetLastError();
}
extern "C" int vgx_test_div_by_const(const double *n, const double *b, int64_t count, double *q_seq, double *q_div) {
    if (!n || !b || !q_seq || !q_div || count < 0) return 1;
    double *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)(count > 0 ? count : 1) * 32) != hipSuccess) return 2;
    int rc = 0;
    if (hipMemcpy(d, n, (size_t)count * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + count, b, (size_t)count * 8, hipMemcpyHostToDevice) != hipSuccess ||
        vgxi_launch_solo_divtest(d, d + count, d + 2 * count, d + 3 * count, count, nullptr) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(q_seq, d + 2 * count, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(q_div, d + 3 * count, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess)
        rc = 2;
    (void)hipFree(d);
    return rc;
}
