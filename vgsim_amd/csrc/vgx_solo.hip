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
//     when the model has no recombination);
//   * uniforms: 64 PCG64 outputs per refill by lane-parallel jump-ahead into LDS, two per iteration as upstream (pyx:477, 488);
//     the logarithm of SampleTime only in calls that need the device clock (a time limit, trajectories, or no event log: event
//     times are rebuilt on the host with libm from the logged rates, vgx_api.hip host_clock);
//   * event records are staged in LDS and written out 64 at a time (2 KB bursts).
// Everything else (Restart, lockdown switches with UpdateAllRates, migration with its rejection step, mutations, immunity
// transitions, trajectories, the logs the host clock needs) follows vgx_direct.hip / vgx_lanes.hip; start and end state are exchanged
// in their layout (occupancy lists, population blocks), so the host side and the other kernels see no difference.
// Exact mode.  Recombinant births (pyx:575-596) take the general path of event().  One wavefront per replicate: ensembles run as
// independent wavefronts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_solo.h"
#include "vgx_flat.h"

// In-kernel stamps (diagnostic build only, -DVGX_PROFILE: tools/profile_solo.py): shader cycles per phase of the event loop, lane i of
// a counter vector <-> phase i, written to the debug buffer r.prof at the end.  No stamp executes in the product build.
#ifdef VGX_PROFILE
#define PROF(i)                                                                   \
    do {                                                                          \
        const unsigned long long prof_t1 = __builtin_readcyclecounter();          \
        prof_acc += (lane == (i)) ? prof_t1 - prof_t0 : 0ull;                      \
        prof_t0 = prof_t1;                                                        \
    } while (0)
#else
#define PROF(i)
#endif
#ifdef VGX_SOLO_MARKS
#define MARK(name) asm volatile("; MARK " name)
#else
#define MARK(name)
#endif

namespace {

enum { ERR_ZERO_WEIGHT = 3, ERR_CAPACITY = 4, ERR_LOOP_GUARD = 5 };
enum { EV_BIRTH = 0, EV_DEATH, EV_SAMPLING, EV_MUTATION, EV_SUSCCHANGE, EV_MIGRATION };


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

// slots of the cold block (LDS): bookkeeping of the call that the event loop itself never reads
enum { C_EV_PTR = 0, C_LOOPS, C_ATT_LOOPS, C_LOC_N, C_TRAJ_NEXT, C_FA_N, C_ATT_EV0, C_ATT_LOC0, C_RESTARTS, C_ATT, C_GOOD, C_LAST_ATT, C_REC_N };
// lanes of the counter vector
enum { CNT_MIGN = 6, CNT_SWAP = 7 };
#ifndef VGX_SOLO_SEG
#define VGX_SOLO_SEG (1 << 30)   // iterations per segment of the event loop (a test build shortens it: the segment ends are then exercised)
#endif
#define SOLO_BIG VGX_SOLO_SEG

// The kernel's arguments are read through the kernarg segment pointer (constant address space: scalar loads, wave-uniform values;
// taking the address of a by-value kernel parameter would make a private copy whose loads count as divergent).  The cold paths see
// that pointer as an opaque value, so that nothing they read from it is hoisted into the event loop and held in registers there.
struct VgxSoloKArgs { VgxDirectArgs a; VgxSoloArgs sa; };
typedef const VgxSoloKArgs __attribute__((address_space(4))) *SoloKA;
static __device__ __forceinline__ SoloKA cold_args(SoloKA k) {
    asm volatile("" : "+s"(k));
    return k;
}

// UNIT: one haplotype, one population, one susceptibility group (BASELINE config 2; upstream's default model): every chain is one addition,
// every fastChoose over them stops at index 0.
// NPR: registers of population lanes (1: popNum <= 64, 2: <= 128).  CLOCK: the device clock runs (SampleTime's logarithm).
// RCPDIV: BirthRate's x / actualSizes through the reciprocal.  NT: 0 = general BirthRate (one pass per segment), 1 / 2 = compact layout
// with that many registers of terms (popNum <= 64 only).  TINY: popNum, susNum and the compact layout's terms are all <= 4: the
// chains over them are four steps without a way out.
template <int NPR, bool CLOCK, bool RCPDIV, int NT, bool TINY, bool UNIT>
struct Solo {
    static constexpr int NTT = NT > 0 ? NT : 1;
    // ---- shape (wave-uniform: every one of these went through v_readfirstlane, so the compiler keeps them scalar) ----
    int P, H, S, sites, nseg, lane, l15;
    int small;            // P <= 16 and H <= 16: the terms of BirthRate already sit in the row of the haplotype lanes (general layout)
    int no_imm;           // every suscepCumulTransition is zero: immuneSourcePopRate stays +0.0
    int maxterms;         // compact layout: terms of the longest class sum
    int recomb;           // recombination_probability > 0: births go through the general form (the recombination branch of Birth, pyx:575-596)
    int mig_lds;          // migrationRates lies in LDS
    int one_cls;          // compact layout with ONE susceptibility class: all four rows hold its terms (no cross-row read of the sum)
    Masks M;
    // LDS
    double *rowI, *rowCum, *rowHpr, *rowBirth, *rowTE, *susS, *susSt, *susImm, *ldSigma, *ldTrans, *ldMrate, *ldHmt, *ldCd, *ldAs, *ldSmult, *ldMig, *ldRng;
    uint32_t *ldStage;
    uint64_t *ldRngK, *ldRngS;
    int64_t *ldCold;
    const double *gMig;   // migrationRates in global memory (when it does not fit LDS)
    double *gEff;         // effectiveMigration of this replicate [P][P]
    // ---- haplotype lanes ----
    double bh, dh, sh, tmh;      // bRate, dRate, sRate, sum of mRate of haplotype `lane`
    int stype;                   // suscType
    uint32_t path;               // general layout, bit s: segment s is on this haplotype's path
    int hapClsLane;              // compact layout: 16 * (susceptibility class of the haplotype) = first lane of its class's row
    // the current row (population `cur`)
    int cur;
    double I, cum, hpr, birth, tE, e1, e2, sm;
    double mrow[NPR];            // general layout: migrationRates[cur][lane + 64 r]
    // ---- susceptibility-group lanes of the current row: lane (row, s) <-> group s, the same values in all four rows ----
    double Ssus, Sst, imms, cumul_l;
    double sigcs;                // compact layout: susceptibility of class `row` in group s
    // ---- segment lanes (general layout) ----
    double Sseg, sgsig;
    int sgsn;
    // general layout beyond 16 populations / haplotypes (vgx_solo.h): the segment lanes hold the program's segments; lane k the chains of pass k
    int sgpar, hseg;             // parent segment of segment `lane`; haplotype `lane`'s last segment (-1: BirthRate's sum is +0.0)
    int passA0, passB0, passA1, passB1, npass0, npass1;
    // ---- term lanes (compact layout): lane (c, j), register t <-> term 16 t + j of class c's BirthRate sum ----
    double tlS[NTT], tlSig[NTT], tlM[NTT], tlCd[NTT], tlAs[NTT], tlRcp[NTT];   // susceptible count of the term's group, its susceptibility,
                                                                            // migrationRates[cur][pn], contact density, actualSizes and 1 / it
    int tlSn[NTT], tlPn[NTT];    // the term's group and population
    // ---- population lanes ----
    double popRate[NPR], cumPop[NPR], infectP[NPR], immuneP[NPR], migR[NPR], maxEBM[NPR], totS[NPR], totI[NPR], cd[NPR], asz[NPR], rcp[NPR];
    double thrCur[NPR], sgnLD[NPR];   // the threshold whose crossing switches the population's lockdown state; +1: switch on above it, -1: off below it
    // ---- counters: lane t < 6 <-> events of type t, lane 6 rejected migrations, lane 7 lockdown switches ----
    uint64_t cnt;
    // ---- wave-uniform ----
    double totalRate, totalMig, Rtot, currentTime, gI;
    int has_mig, ld_any;
    unsigned long long zero_w;    // some fastChoose of this call stopped on a zero weight
    int stage_n;
    int pos;                      // iterations consumed from the current batch of 64 uniforms (32 = empty)
    double u_pre, n_pre;          // the uniform of GenerateEvent / -log of the uniform of SampleTime for iteration `pos`, read ahead
    double tlimit, next_tg;       // CLOCK: the call's time limit (+inf without one); the next trajectory grid time (+inf without trajectories)
    int ev_left, loop_left, s_left, ev_left0, loop_left0;
    uint32_t iter_base;           // low word of the attempt's loop-iteration count at the segment's start + loop_left0
#ifdef VGX_PROFILE
    unsigned long long prof_t0, prof_acc;
#endif

    __device__ __forceinline__ double mig_at(int i) const {   // (a branch, not a select of pointers: that would be a flat access)
        if (mig_lds) return ldMig[i];
        return gMig[i];
    }
    __device__ __forceinline__ int64_t cold_get(int i) const { return ldCold[i]; }
    __device__ __forceinline__ void cold_set(int i, int64_t v) {
        if (lane == 0) ldCold[i] = v;
        WSYNC();
    }
    // the uniforms of iteration `pos` (of the batch in LDS) on their way into registers
    __device__ __forceinline__ void prefetch_uniforms() {
        const int q = min(pos, 31);
        u_pre = ldRng[2 * q + 1];
        if (CLOCK) n_pre = ldRng[2 * q];
    }

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
        if (l15 < S) {
            const int o = pi * S + l15;
            Ssus = susS[o]; Sst = susSt[o]; imms = susImm[o];
        }
        if (NT > 0) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                tlS[t] = susS[pi * S + tlSn[t]];
                tlM[t] = mig_at(pi * P + tlPn[t]);
            }
        } else {
            Sseg = (lane < nseg) ? susS[pi * S + sgsn] : 0.0;
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const int pn = lane + 64 * q;
                mrow[q] = pn < P ? mig_at(pi * P + pn) : 0.0;
            }
        }
        sm = sh * ldSmult[pi];
        e1 = birth + dh;
        e2 = e1 + sm;
        cur = uni_i32(pi);
    }
    __device__ __forceinline__ void row_switch(int pi) {
        if (pi == cur) return;
        row_store();
        row_load(pi);
    }
    // the current row's susceptible group `sidx` changes by -d (NewInfections +1 / NewRecoveries -1, pyx:246-260)
    __device__ __forceinline__ void sus_add(int sidx, double d) {
        Ssus -= (l15 == sidx ? d : 0.0);
        if (NT > 0) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) tlS[t] -= (tlSn[t] == sidx ? d : 0.0);
        } else {
            Sseg -= (lane < nseg && sgsn == sidx ? d : 0.0);
        }
    }

    // ---- UpdateRates, infect branch, for the current row (pyx:518-528) in two halves: BirthRate's sum (pyx:382-392) of every
    // haplotype lane, then the rates, hapPopRate and its prefix sums; returns infectPopRate.  (Two halves so that the event loop can
    // put work that does not depend on the sum between them: the compact layout reads it across rows, an LDS round trip.) ----
    __device__ __forceinline__ double refresh_row() { return refresh_rates(birth_sums()); }
    // the terms of one chain of a pass: segment j >= 0 (xs = S[cur, sn_j] * sigma_j), the migration rates (j = -2), nothing (j = -1)
    __device__ __forceinline__ void pass_terms(int j, double xseg, double segv, double (&T)[NPR], double &carry) const {
        carry = 0.0;
        if (j >= 0) {
            const double xs = bcast(xseg, j);
            const int par = uni_i32(__builtin_amdgcn_readlane(sgpar, j));
            if (par >= 0) carry = bcast(segv, par);
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const double t = xs * mrow[q] * mrow[q] * cd[q];
                const double d = RCPDIV ? div_by_const(t, asz[q], rcp[q]) : t / asz[q];
                T[q] = lane + 64 * q < P ? d : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NPR; ++q) T[q] = j == -2 ? migR[q] : 0.0;
        }
    }
    // BirthRate's sum for every haplotype lane by the program of chain segments, two chains per pass; MIG: the sum of the migration
    // rates (remig) rides along as one of the chains
    template <bool MIG>
    __device__ __forceinline__ double birth_sums_program() {
        Sst = Ssus;
        const bool mig = MIG && has_mig;
        if (MIG) {
            if (mig) {
#pragma unroll
                for (int q = 0; q < NPR; ++q) migR[q] = maxEBM[q] * totS[q] * (gI - totI[q]);
            } else {
                totalMig = 0.0;
            }
        }
        const double xseg = Sseg * sgsig;
        double segv = 0.0;                          // lane sg: the sum at the end of segment sg
        const int np = mig ? npass1 : npass0;
        const int rows0 = (min(P, 64) + 15) >> 4, rows1 = (P - 64 + 15) >> 4, last = (P - 1) & 63;
        for (int k = 0; k < np; ++k) {
            const int ja = uni_i32(__builtin_amdgcn_readlane(mig ? passA1 : passA0, k)), jb = uni_i32(__builtin_amdgcn_readlane(mig ? passB1 : passB0, k));
            double TA[NPR], TB[NPR], ca, cb, oa, ob;
            pass_terms(ja, xseg, segv, TA, ca);
            if (jb == -1) {         // (a chain without a partner: alone — a lone wavefront pays per instruction, an idle second chain is not free)
                oa = flat_chain<false>(TA[0], min(P, 64), ca, M);
                if (NPR > 1 && P > 64) oa = flat_chain<false>(TA[NPR - 1], P - 64, bcast(oa, 63), M);
                ob = 0.0;
            } else {
                pass_terms(jb, xseg, segv, TB, cb);
                flat_two_sums(TA[0], TB[0], rows0, ca, cb, oa, ob);
                if (NPR > 1 && P > 64) flat_two_sums(TA[NPR - 1], TB[NPR - 1], rows1, bcast(oa, 63), bcast(ob, 63), oa, ob);
            }
            const double ta = bcast(oa, last), tb = bcast(ob, last);
            if (ja >= 0) segv = lane == ja ? ta : segv; else if (ja == -2) totalMig = ta;
            if (jb >= 0) segv = lane == jb ? tb : segv; else if (jb == -2) totalMig = tb;
        }
        const double ps = bperm_f64(segv, max(hseg, 0));
        return hseg >= 0 ? ps : 0.0;
    }
    __device__ __forceinline__ double birth_sums() {
        if (NT == 0 && !small) return birth_sums_program<false>();
        Sst = Ssus;                                 // BirthRate stores susceptHapPopRate = S * sigma (pyx:385-386)
        double ps;
        if (NT > 0) {
            // compact layout: every term of every class at once, one class per row; a row's chain is the class's sum in the
            // reference's (sn, pn) order with the zero-susceptibility terms left out (+0.0 upstream)
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                const double tt = tlS[t] * tlSig[t] * tlM[t] * tlM[t] * tlCd[t];
                const double T = RCPDIV ? div_by_const(tt, tlAs[t], tlRcp[t]) : tt / tlAs[t];
                acc = t == 0 ? rows_chain<TINY, UNIT>(acc, T, 1.0, min(16, maxterms)) : rows_chain<false>(acc, T, 1.0, maxterms - 16);
            }
            ps = one_cls ? acc : bperm_f64(acc, hapClsLane);
        } else {
            const double xseg = Sseg * sgsig;       // segment lanes
            ps = 0.0;
            for (int s = 0; s < nseg; ++s) {
                const double xs = bcast(xseg, s);
                const double mu = (double)((path >> s) & 1u);
                double T[NPR];
#pragma unroll
                for (int q = 0; q < NPR; ++q) {
                    const double t = xs * mrow[q] * mrow[q] * cd[q];
                    T[q] = RCPDIV ? div_by_const(t, asz[q], rcp[q]) : t / asz[q];
                }
                if (small) {
                    ps = rows_chain<false>(ps, T[0], mu, P);
                } else {
                    for (int b = 0; b < P; b += 16) {
                        const double src = bperm_f64(b < 64 ? T[0] : T[NPR - 1], (b & 63) + l15);
                        ps = rows_chain<false>(ps, src, mu, min(16, P - b));
                    }
                }
            }
        }
        return ps;
    }
    __device__ __forceinline__ double refresh_rates(double ps) {
        PROF(8);
        MARK("birthrate_done");
        birth = bh * ps;
        e1 = birth + dh;
        e2 = e1 + sm;
        tE = e2 + tmh;                              // ((r0 + r1) + r2) + r3, pyx:522-525
        hpr = tE * I;
        cum = flat_chain<true, false, UNIT>(hpr, H, 0.0, M);
        MARK("row_scanned");
        return bcast(cum, H - 1);
    }

    // popRate[pi] changed: its serial prefix sums and totalRate (pyx:536-539)
    __device__ __forceinline__ void rescan_pop() {
        cumPop[0] = flat_chain<true, TINY, UNIT>(popRate[0], min(P, 64), 0.0, M);
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
        double acc = flat_chain<false, TINY, UNIT>(migR[0], min(P, 64), 0.0, M);
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
        const double acc = flat_chain<false, TINY, UNIT>(imms, S, 0.0, M);
        return bcast(acc, S - 1);
    }

    // ---- UpdateAllRates (pyx:279-351); the parameter-only parts come from the host (vgx_api.hip) ----
    __device__ __forceinline__ void rebuild_all(SoloKA ka) {
        if (NT > 0) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) tlCd[t] = ldCd[tlPn[t]];
        }
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
        const double meb = cold_args(ka)->a.p.maxEffectiveBirth;
#pragma unroll
        for (int q = 0; q < NPR; ++q) {
            maxEBM[q] = mx[q] * meb;
            any = any || maxEBM[q] > 0.0;
        }
        has_mig = any_lane(any) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < NPR; ++q) migR[q] = 0.0;
        remig();
        Rtot = totalRate + totalMig;
        WSYNC();
    }

    // ---- event log: records are staged in LDS (32 bytes each) and written out 64 at a time ----
    __device__ __forceinline__ void stage_flush(SoloKA ka_, int64_t rep) {
        if (stage_n > 0) {
            const auto *a = &cold_args(ka_)->a;
            WSYNC();
            // slot of the first staged record: events.ptr now, minus the staged ones, relative to the log's base
            const int64_t ev_now = cold_get(C_EV_PTR) + (int64_t)(ev_left0 - ev_left);
            const int64_t slot0 = ev_now - stage_n - a->r.ev_base;
            if (slot0 < 0 || slot0 + stage_n > a->r.evcap) {
                zero_w |= 2ull << 32;   // (reported as a capacity error)
            } else if (lane < stage_n) {
                const uint32_t *s = ldStage + lane * 8;
                const int64_t slot = slot0 + lane;
                int32_t *c = a->r.ev_cols + (rep * a->r.evcap + slot) * VGX_EV_COLS;
                c[0] = (int32_t)s[0]; c[1] = (int32_t)s[1]; c[2] = (int32_t)s[2]; c[3] = (int32_t)s[3]; c[4] = (int32_t)s[4]; c[5] = (int32_t)s[5];
                a->r.ev_rate[rep * a->r.evcap + slot] = __hiloint2double((int)s[7], (int)s[6]);
            }
            WSYNC();
        }
        stage_n = 0;
    }

    // ---- CheckLockdown (pyx:698-710) for populations [lo, hi): applies and logs the switches; returns whether any happened ----
    __device__ __forceinline__ bool check_lockdowns(SoloKA ka_, int64_t rep, int lo, int hi) {
        const auto *a = &cold_args(ka_)->a;
        const auto &p = a->p;
        const auto &r = a->r;
        bool any = false;
        int64_t loc_n = cold_get(C_LOC_N);
        const int64_t iter_key = (cold_get(C_ATT) << 40) | (cold_get(C_ATT_LOOPS) + (int64_t)(loop_left0 - loop_left));
        for (int pi = lo; pi < hi; ++pi)
            for (int pass = 0; pass < 2; ++pass) {
                const double ti = pop_get<NPR>(totI, pi), sg = pop_get<NPR>(sgnLD, pi);
                const double sz = (double)p.sizes[pi];
                const bool flip = pass == 0 ? (ti > p.startLD[pi] * sz && sg > 0.0) : (ti < p.endLD[pi] * sz && sg < 0.0);
                if (!any_lane(flip)) continue;
                const double ncd = pass == 0 ? p.cdAfter[pi] : p.cdBefore[pi];
                pop_set<NPR>(cd, pi, ncd, lane);
                pop_set<NPR>(sgnLD, pi, pass == 0 ? -1.0 : 1.0, lane);
                pop_set<NPR>(thrCur, pi, pass == 0 ? p.endLD[pi] * sz : p.startLD[pi] * sz, lane);
                if (lane == 0) {
                    ldCd[pi] = ncd;
                    if (loc_n < r.loc_cap) {
                        r.loc_rec[(rep * r.loc_cap + loc_n) * 2 + 0] = pass == 0 ? 1 : 0;
                        r.loc_rec[(rep * r.loc_cap + loc_n) * 2 + 1] = pi;
                        r.loc_time[rep * r.loc_cap + loc_n] = currentTime;
                        r.loc_iter[rep * r.loc_cap + loc_n] = iter_key;
                    }
                }
                if (loc_n >= r.loc_cap) zero_w |= 2ull << 32;
                cnt += (lane == CNT_SWAP) ? 1u : 0u;
                loc_n += 1;
                any = true;
            }
        if (any) cold_set(C_LOC_N, loc_n);
        return any;
    }

    __device__ __forceinline__ void traj_emit(SoloKA ka_, int64_t rep, double t_new, bool final_fill) {
        const auto &r = cold_args(ka_)->a.r;
        int64_t traj_next = cold_get(C_TRAJ_NEXT);
        const int64_t n0 = traj_next;
        while (traj_next < r.traj_points) {
            const double tg = r.traj_t0 + (double)traj_next * r.traj_dt;
            if (!final_fill && !(tg < t_new)) break;
            double *o = r.traj + (rep * r.traj_points + traj_next) * (int64_t)P * 2;
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const int pn = lane + 64 * q;
                if (pn < P) { o[pn * 2 + 0] = totI[q]; o[pn * 2 + 1] = totS[q]; }
            }
            traj_next += 1;
        }
        if (traj_next != n0) cold_set(C_TRAJ_NEXT, traj_next);
    }

    // first lane of `hit` or the clamp at n - 1 (fast_choose.pxi:26)
    static __device__ __forceinline__ int first_or_last(unsigned long long hit, int n) {
        return uni_i32(hit ? (int)__builtin_ctzll(hit) : n - 1);   // (readfirstlane: the index is wave-uniform, and the compiler should know)
    }
    // a fastChoose that stops on a zero weight makes the reference exit (fast_choose.pxi:5-13); here the call ends with an error
    // after the event (indices stay valid, so the rest of the event runs on harmlessly)
    __device__ __forceinline__ void zero_weight(bool c) { zero_w |= __builtin_amdgcn_ballot_w64(c); }

    // fastChoose(popRate, totalRate, rn) on the cached serial prefix sums (fast_choose.pxi:18-31): r2 = totalRate * rn
    __device__ __forceinline__ int choose_pop(double r2) {
        if (UNIT) return 0;
        unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < P && !(cumPop[0] < r2));
        if (NPR == 1) return first_or_last(hit, P);
        int pi;
        if (hit) pi = (int)__builtin_ctzll(hit);
        else {
            hit = __builtin_amdgcn_ballot_w64(lane + 64 < P && !(cumPop[NPR - 1] < r2));
            pi = hit ? 64 + (int)__builtin_ctzll(hit) : P - 1;
        }
        return uni_i32(pi);
    }
    // Birth's fastChoose(susceptHapPopRate[pi, hi, :], their sum, rn7) (pyx:569-572): the group that loses a host
    // (rn_out: the random number rescaled once more, fast_choose.pxi:31 — read by the recombination branch only)
    __device__ __forceinline__ int choose_group(int hi, double rn7, double *rn_out = nullptr) {
        if (NT > 0) {
            // compact layout: lane (c, s) holds class c's susceptHapPopRate for group s; hi's class decides which row is read
            const double x = Sst * sigcs;
            const int c16 = one_cls ? 0 : uni_i32(__builtin_amdgcn_readlane(hapClsLane, hi));
            int sidx = 0;
            if (S > 1) {
                const double cx = rows_scan<TINY>(x, S, M);
                const double r8 = bcast(cx, c16 + S - 1) * rn7;
                const unsigned long long hit = (__builtin_amdgcn_ballot_w64(l15 < S && !(cx < r8)) >> c16) & 0xffffull;
                sidx = first_or_last(hit, S);
            }
            zero_w |= (__builtin_amdgcn_ballot_w64(x == 0.0) >> (c16 + sidx)) & 1ull;
            if (rn_out) {
                const double w = bcast(x, c16 + sidx), tot = S > 1 ? bcast(rows_scan<TINY>(x, S, M), c16 + sidx) : w;
                const double r8 = (S > 1 ? bcast(rows_scan<TINY>(x, S, M), c16 + S - 1) : w) * rn7;
                *rn_out = (r8 - (tot - w)) / w;
            }
            return sidx;
        }
        const double x = lane < S ? Sst * ldSigma[hi * S + lane] : 0.0;   // susceptHapPopRate[pi, hi, :]
        int sidx = 0;
        if (S > 1) {
            const double cx = flat_chain<true>(x, S, 0.0, M);
            const double r8 = bcast(cx, S - 1) * rn7;
            sidx = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(cx < r8)), S);
        }
        zero_weight(lane == sidx && x == 0.0);
        if (rn_out) {
            const double cx = flat_chain<true>(x, S, 0.0, M);
            const double w = bcast(x, sidx), tot = bcast(cx, sidx);
            *rn_out = (bcast(cx, S - 1) * rn7 - (tot - w)) / w;
        }
        return sidx;
    }

    // ---- AddEvent (events.pxi:37-44) into the LDS stage + the counters; type < 0: a rejected migration (counter only) ----
    template <bool KNOWN = false>   // KNOWN: type >= 0
    __device__ __forceinline__ void log_event(int type, int hap, int pop, int nh, int np, double den) {
        const int ctr = (KNOWN || type >= 0) ? type : CNT_MIGN;
        cnt += (lane == ctr) ? 1u : 0u;
        if (KNOWN || type >= 0) {
            int v = __double2hiint(den);                                  // lane 7 (and the unused lanes)
            const int it = uni_i32((int)(iter_base - (uint32_t)loop_left));
            type = uni_i32(type); hap = uni_i32(hap); pop = uni_i32(pop); nh = uni_i32(nh); np = uni_i32(np);
            SOLO_WRITELANE(v, type, 0); SOLO_WRITELANE(v, hap, 1); SOLO_WRITELANE(v, pop, 2); SOLO_WRITELANE(v, nh, 3);
            SOLO_WRITELANE(v, np, 4); SOLO_WRITELANE(v, it, 5);
            v = lane == 6 ? __double2loint(den) : v;
            if (lane < 8) ldStage[stage_n * 8 + lane] = (uint32_t)v;   // (staged also without an event log: never flushed then)
            stage_n += 1;
            ev_left -= 1;
            s_left -= (type == EV_SAMPLING) ? 1 : 0;
        }
    }

    // ---- the event loop proper: iterations whose event is a Birth, Death or Sampling (pyx:568-635) run here back to back as one
    // straight path; it returns when something else has to happen — why, and for FAST_SLOW the uniform of GenerateEvent (nothing of
    // the iteration is consumed yet: the general form below repeats its choice), for FAST_POST the event's population ----
    enum { FAST_END = 0, FAST_REFILL, FAST_SLOW, FAST_POST };
    __device__ __forceinline__ int fast_loop(double &u_slow, int &pi_post) {
        for (;;) {
            PROF(14);
            MARK("loop_top");
            if (min(min(ev_left, loop_left), s_left) <= 0) return FAST_END;
            if (CLOCK && !any_lane(currentTime < tlimit)) return FAST_END;
            if (pos == 32) return FAST_REFILL;
            const double u = u_pre;
            double t_new = 0.0;
            if (CLOCK) {
                t_new = currentTime + (n_pre / Rtot);                      // SampleTime pyx:476-478
                if (any_lane(next_tg < t_new)) { u_slow = u; return FAST_SLOW; }
            }
            PROF(0);
            MARK("event_begin");
            // GenerateEvent (pyx:483-512)
            double choose = u * Rtot;
            if (!any_lane(totalRate > choose)) { u_slow = u; return FAST_SLOW; }
            double rn = fdiv(choose, totalRate);
            const int pi = choose_pop(totalRate * rn);
            PROF(2);
            const double W = pop_get<NPR>(popRate, pi), Cm = pop_get<NPR>(cumPop, pi);
            const double IM = pop_get<NPR>(immuneP, pi), IN = pop_get<NPR>(infectP, pi);
            zero_weight(W == 0.0);
            rn = fdiv(totalRate * rn - (Cm - W), W);
            choose = rn * W;
            if (any_lane(IM > choose)) { u_slow = u; return FAST_SLOW; }
            PROF(3);
            row_switch(pi);
            PROF(4);
            rn = fdiv(choose - IM, IN);
            const double r4 = IN * rn;                                     // fastChoose(hapPopRate[pi], infectPopRate[pi], rn)
            const int hi = UNIT ? 0 : first_or_last(__builtin_amdgcn_ballot_w64(lane < H && !(cum < r4)), H);
            const double rn5 = fdiv(r4 - (cum - hpr), hpr);                // every candidate lane forms its own rescaled number
            const double r6 = tE * rn5;                                    // fastChoose(eventHapPopRate[pi, hi, 0..3], tEventHapPopRate[pi, hi], rn)
            const int eil = (birth < r6 ? 1 : 0) + (e1 < r6 ? 1 : 0) + (e2 < r6 ? 1 : 0);
            const int ei = __builtin_amdgcn_readlane(eil, hi);
            if (ei == 3 || (ei == 0 && recomb)) { u_slow = u; return FAST_SLOW; }
            PROF(5);
            // ---- the iteration is this path's: Birth (pyx:568-605, no recombination) / Death / Sampling (pyx:616-635) ----
            loop_left -= 1;
            pos += 1;
            prefetch_uniforms();
            if (CLOCK) currentTime = t_new;
            zero_weight(lane == hi && hpr == 0.0);
            int sidx;
            if (ei == 0) {
                zero_weight(lane == hi && birth == 0.0);
                sidx = choose_group(hi, S > 1 ? bcast(fdiv(r6, birth), hi) : 0.0);   // (r - (e0 - e0)) / e0
            } else {
                sidx = uni_i32(__builtin_amdgcn_readlane(stype, hi));
            }
            // NewInfections / NewRecoveries (pyx:246-260)
            const double sgn = ei == 0 ? 1.0 : -1.0;
            sus_add(sidx, sgn);
            I += (lane == hi ? sgn : 0.0);
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                const double d1 = (lane + 64 * q == pi) ? sgn : 0.0;
                totS[q] -= d1; totI[q] += d1;
            }
            gI += sgn;
            imms = l15 == sidx ? cumul_l * Ssus : imms;
            PROF(6);
            // UpdateRates(pi, True, True, True), pyx:516-546
            double ps;
            if (NT == 0 && !small) {
                ps = birth_sums_program<true>();    // (with the migration rates' sum)
            } else {
                ps = birth_sums();
            }
            const double imP = immune_sum();
            PROF(10);
            if (!(NT == 0 && !small)) remig();
            PROF(12);
            const double inP = refresh_rates(ps);
            PROF(9);
            pop_set<NPR>(infectP, pi, inP, lane);
            pop_set<NPR>(immuneP, pi, imP, lane);
            pop_set<NPR>(popRate, pi, inP + imP, lane);
            rescan_pop();
            PROF(11);
            const double den = Rtot;
            Rtot = totalRate + totalMig;
            log_event<true>(ei, hi, pi, sidx, ei == 0 ? H : 0, den);
            PROF(13);
            // what ends the run of fast iterations: a full stage, a zero weight, extinction (pyx:410-411), a lockdown threshold crossed
            // (pyx:412; never true where no lockdown can switch: the threshold then lies at or above the population's size) — one test,
            // no short circuits
            bool stop = (totalRate == 0.0) | (gI == 0.0);
#pragma unroll
            for (int q = 0; q < NPR; ++q) stop = stop | ((lane + 64 * q == pi) & ((totI[q] - thrCur[q]) * sgnLD[q] > 0.0));
            const unsigned long long post = __builtin_amdgcn_ballot_w64(stop) | zero_w | (unsigned long long)(stage_n >> 6);
            if (post) { pi_post = pi; return FAST_POST; }
        }
    }

    // ---- the same loop for ONE haplotype, ONE population, ONE group without immunity loss (BASELINE config 2; upstream's default model).
    // There totalRate, popRate[0], infectPopRate[0] and hapPopRate[0, 0] are ONE number — popRate = infectPopRate + (+0.0), the prefix
    // sums of one term are 0.0 + the term — so GenerateEvent's four rescalings (pyx:483-511, fc:18-31) divide by the same divisor: one
    // refined reciprocal serves all four (fdiv_y: the quotients are the same correctly rounded ones), every index is 0 without a
    // search, `x - (cum - w)` is `x - 0.0`, and no row is switched.  The apply / UpdateRates half is fast_loop's. ----
    __device__ __forceinline__ int unit_fast_loop(double &u_slow, int &pi_post) {
        for (;;) {
            PROF(14);
            if (min(min(ev_left, loop_left), s_left) <= 0) return FAST_END;
            if (CLOCK && !any_lane(currentTime < tlimit)) return FAST_END;
            if (pos == 32) return FAST_REFILL;
            const double u = u_pre;
            double t_new = 0.0;
            if (CLOCK) {
                t_new = currentTime + (n_pre / Rtot);
                if (any_lane(next_tg < t_new)) { u_slow = u; return FAST_SLOW; }
            }
            PROF(0);
            const double T = totalRate;
            double choose = u * Rtot;
            if (!any_lane(T > choose)) { u_slow = u; return FAST_SLOW; }
            const double y = refined_rcp(T);
            double rn = fdiv_y(choose, T, y);                              // choose / totalRate
            zero_weight(T == 0.0);
            rn = fdiv_y(T * rn - 0.0, T, y);                               // (r - (cumPop - popRate)) / popRate
            choose = rn * T;                                               // (immunePopRate = +0.0 > choose: never)
            rn = fdiv_y(choose - 0.0, T, y);                               // (choose - immunePopRate) / infectPopRate
            const double rn5 = fdiv_y(T * rn - 0.0, T, y);                 // (r - (cum - hapPopRate)) / hapPopRate
            const double r6 = tE * rn5;
            const int eil = (birth < r6 ? 1 : 0) + (e1 < r6 ? 1 : 0) + (e2 < r6 ? 1 : 0);
            const int ei = __builtin_amdgcn_readlane(eil, 0);
            if (ei == 3 || (ei == 0 && recomb)) { u_slow = u; return FAST_SLOW; }
            PROF(5);
            loop_left -= 1;
            pos += 1;
            prefetch_uniforms();
            if (CLOCK) currentTime = t_new;
            zero_weight(lane == 0 && hpr == 0.0);
            if (ei == 0) {
                zero_weight(lane == 0 && birth == 0.0);
                zero_w |= __builtin_amdgcn_ballot_w64(Sst * sigcs == 0.0) & 1ull;      // choose_group: susceptHapPopRate[0, 0, 0] == 0
            }
            const double sgn = ei == 0 ? 1.0 : -1.0;
            sus_add(0, sgn);
            const double d1 = lane == 0 ? sgn : 0.0;
            I += d1;
            totS[0] -= d1; totI[0] += d1;
            gI += sgn;
            imms = l15 == 0 ? cumul_l * Ssus : imms;
            PROF(6);
            const double ps = birth_sums();
            PROF(10);
            totalMig = 0.0;                                                // (one population: no migration)
            PROF(12);
            const double inP = refresh_rates(ps);
            PROF(9);
            infectP[0] = lane == 0 ? inP : infectP[0];
            immuneP[0] = lane == 0 ? 0.0 : immuneP[0];
            popRate[0] = lane == 0 ? inP + 0.0 : popRate[0];
            cumPop[0] = 0.0 + popRate[0];
            totalRate = bcast(cumPop[0], 0);
            PROF(11);
            const double den = Rtot;
            Rtot = totalRate + totalMig;
            log_event<true>(ei, 0, 0, 0, ei == 0 ? H : 0, den);
            PROF(13);
            const bool stop = (totalRate == 0.0) | (gI == 0.0) | ((lane == 0) & ((totI[0] - thrCur[0]) * sgnLD[0] > 0.0));
            const unsigned long long post = __builtin_amdgcn_ballot_w64(stop) | zero_w | (unsigned long long)(stage_n >> 6);
            if (post) { pi_post = 0; return FAST_POST; }
        }
    }

    // ---- one iteration of the event loop after SampleTime in its general form: GenerateEvent (pyx:483-512) with UpdateRates and
    // AddEvent; returns the population whose lockdown state has to be checked ----
    __device__ __forceinline__ int event(double u) {
        // what the event leaves for the common tail
        int u_pi;                      // population whose rates change
        bool f_infect, f_immune, f_mig;
        double inP = 0.0, imP = 0.0;   // its infect / immune rate where the tail does not refresh them
        int ev_type, ev_hap, ev_pop, ev_nh, ev_np;
        double choose = u * Rtot;
        if (any_lane(totalRate > choose)) {
            double rn = choose / totalRate;
            const double r2 = totalRate * rn;
            const int pi = choose_pop(r2);
            const double W = pop_get<NPR>(popRate, pi), Cm = pop_get<NPR>(cumPop, pi);
            const double IM = pop_get<NPR>(immuneP, pi), IN = pop_get<NPR>(infectP, pi);
            zero_weight(W == 0.0);
            rn = (r2 - (Cm - W)) / W;
            choose = rn * W;
            row_switch(pi);
            u_pi = pi; ev_pop = pi;
            if (any_lane(IM > choose)) {
                // ---- ImmunityTransition (pyx:550-564) ----
                rn = choose / IM;
                int ssi, tsi;
                {
                    const double ci = flat_chain<true>(imms, S, 0.0, M);
                    const double r = IM * rn;
                    ssi = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(ci < r)), S);
                    const double w = bcast(imms, ssi), tot = bcast(ci, ssi);
                    zero_weight(w == 0.0);
                    rn = (r - (tot - w)) / w;
                }
                {
                    const double tr = lane < S ? ldTrans[ssi * S + lane] : 0.0;
                    const double ct = flat_chain<true>(tr, S, 0.0, M);
                    const double r = bcast(cumul_l, ssi) * rn;
                    tsi = first_or_last(__builtin_amdgcn_ballot_w64(lane < S && !(ct < r)), S);
                    zero_weight(lane == tsi && tr == 0.0);
                }
                sus_add(ssi, 1.0);
                sus_add(tsi, -1.0);
                imms = (l15 == ssi || l15 == tsi) ? Ssus * cumul_l : imms;
                f_infect = false; f_immune = true; f_mig = false; inP = IN;
                ev_type = EV_SUSCCHANGE; ev_hap = ssi; ev_nh = tsi; ev_np = 0;
            } else {
                rn = (choose - IM) / IN;
                // fastChoose(hapPopRate[pi], infectPopRate[pi], rn) on the row's cached prefix sums
                const double r4 = IN * rn;
                const int hi = first_or_last(__builtin_amdgcn_ballot_w64(lane < H && !(cum < r4)), H);
                zero_weight(lane == hi && hpr == 0.0);
                const double rn5 = (r4 - (cum - hpr)) / hpr;           // every candidate lane forms its own rescaled number
                // fastChoose(eventHapPopRate[pi, hi, 0..3], tEventHapPopRate[pi, hi], rn): the running totals are e0, e1, e2
                const double r6 = tE * rn5;
                const int eil = (birth < r6 ? 1 : 0) + (e1 < r6 ? 1 : 0) + (e2 < r6 ? 1 : 0);
                const int ei = __builtin_amdgcn_readlane(eil, hi);
                ev_hap = hi;
                if (ei < 3) {
                    // ---- Birth (pyx:568-605, no recombination) / Death / Sampling (pyx:616-635) ----
                    int sidx;
                    int hnew = hi;                 // the haplotype whose count changes
                    ev_np = ei == 0 ? H : 0;
                    if (ei == 0) {
                        zero_weight(lane == hi && birth == 0.0);
                        if (!recomb) {
                            sidx = choose_group(hi, S > 1 ? bcast(r6 / birth, hi) : 0.0);   // (r - (e0 - e0)) / e0
                        } else {
                            double rnr;
                            sidx = choose_group(hi, bcast(r6 / birth, hi), &rnr);
                            const auto &p = cold_args((SoloKA)__builtin_amdgcn_kernarg_segment_ptr())->a.p;
                            if (any_lane(rnr < p.recombination) && any_lane(pop_get<NPR>(totI, pi) > 1.0)) {
                                // ---- recombinant birth (pyx:575-596): the second parent by fastChoose over
                                // birthInf[hn] = eventHapPopRate[pi, hn, 0] * infectious[pi, hn] with one host of hi set aside ----
                                rnr = rnr / p.recombination;
                                const double w = birth * (I - (lane == hi ? 1.0 : 0.0));
                                const double cw = flat_chain<true>(w, H, 0.0, M);
                                const double r = bcast(cw, H - 1) * rnr;
                                const int hi2 = first_or_last(__builtin_amdgcn_ballot_w64(lane < H && !(cw < r)), H);
                                const double w2 = bcast(w, hi2), tot2 = bcast(cw, hi2);
                                zero_weight(w2 == 0.0);
                                rnr = (r - (tot2 - w2)) / w2;
                                const int64_t posRecomb = (int64_t)((double)p.genome_length * uni_f64(rnr));
                                // pyx:586-591 as written: `4**k * floor(h / 4**k) % 4` is 0 for every site but the last, where it is
                                // h % 4 — the recombinant carries only the last site of one parent
                                hnew = sites > 0 ? ((p.sitesPosition[sites - 1] < posRecomb ? hi : hi2) % 4) : 0;
                                hnew = uni_i32(hnew);
                                const auto &rr = cold_args((SoloKA)__builtin_amdgcn_kernarg_segment_ptr())->a.r;
                                const int64_t rec_n = cold_get(C_REC_N), rep = blockIdx.x;
                                if (rr.rec) {
                                    if (rec_n < rr.rec_cap) {
                                        if (lane == 0) {
                                            int64_t *o = rr.rec + (rep * rr.rec_cap + rec_n) * 5;
                                            o[0] = cold_get(C_EV_PTR) + (int64_t)(ev_left0 - ev_left); o[1] = hi; o[2] = hi2; o[3] = hnew; o[4] = posRecomb;
                                        }
                                    } else {
                                        zero_w |= 2ull << 32;
                                    }
                                }
                                cold_set(C_REC_N, rec_n + 1);
                                ev_np = hi2;
                            }
                        }
                    } else {
                        sidx = __builtin_amdgcn_readlane(stype, hi);
                    }
                    sidx = uni_i32(sidx);
                    // NewInfections / NewRecoveries (pyx:246-260)
                    const double sgn = ei == 0 ? 1.0 : -1.0;
                    sus_add(sidx, sgn);
                    I += (lane == hnew ? sgn : 0.0);
#pragma unroll
                    for (int q = 0; q < NPR; ++q) {
                        const double d1 = (lane + 64 * q == pi) ? sgn : 0.0;
                        totS[q] -= d1; totI[q] += d1;
                    }
                    gI += sgn;
                    imms = l15 == sidx ? cumul_l * Ssus : imms;
                    f_infect = true; f_immune = true; f_mig = true;
                    ev_type = ei; ev_nh = sidx;
                } else {
                    // ---- Mutation (pyx:640-667) ----
                    const double tEh = bcast(tE, hi), tmv = bcast(tmh, hi), r6h = bcast(r6, hi);
                    zero_weight(tmv == 0.0);
                    rn = (r6h - (tEh - tmv)) / tmv;
                    int mi, DS;
                    {   // fastChoose(mRate[hi, :], tmRate[hi], rn)
                        const double *w = ldMrate + hi * sites;
                        const double r = tmv * rn;
                        int i = 0;
                        double total = w[0];
                        while (any_lane(total < r) && i < sites - 1) { i += 1; total += w[i]; }
                        const double wi = w[i];
                        zero_weight(wi == 0.0);
                        rn = (r - (total - wi)) / wi;
                        mi = uni_i32(i);
                    }
                    {   // fastChoose(hapMutType[hi, mi, :], their sum, rn)
                        const double *w = ldHmt + (hi * sites + mi) * 3;
                        const double r = (w[0] + w[1] + w[2]) * rn;
                        int i = 0;
                        double total = w[0];
                        while (any_lane(total < r) && i < 2) { i += 1; total += w[i]; }
                        zero_weight(w[i] == 0.0);
                        DS = uni_i32(i);
                    }
                    const int digit4 = 1 << (2 * (sites - mi - 1));      // Mutate, pyx:2420-2427
                    const int AS = (hi / digit4) % 4;
                    if (DS >= AS) DS += 1;
                    const int nhi = hi + (DS - AS) * digit4;
                    I += (lane == nhi ? 1.0 : 0.0) - (lane == hi ? 1.0 : 0.0);
                    f_infect = true; f_immune = false; f_mig = false; imP = IM;
                    ev_type = EV_MUTATION; ev_nh = nhi; ev_np = 0;
                }
            }
        } else {
            // ---- GenerateMigration (pyx:672-694) ----
            double rn = (choose - totalRate) / totalMig;
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
                tpi = uni_i32(tpi);
                const double w = pop_get<NPR>(migR, tpi), tot = pop_get<NPR>(cm, tpi);
                zero_weight(w == 0.0);
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
                spi = uni_i32(spi);
                const double w = pop_get<NPR>(totI, spi), tot = pop_get<NPR>(cs, spi);
                zero_weight(w == 0.0);
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
                zero_weight(w == 0.0);
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
                zero_weight(w == 0.0);
                rn = (r - (tot - w)) / w;
            }
            const double p_accept = gEff[spi * P + tpi] * b_hi * ldSigma[hi * S + si] / pop_get<NPR>(maxEBM, tpi);
            u_pi = tpi;
            ev_type = -1; ev_hap = hi; ev_pop = spi; ev_nh = si; ev_np = tpi;
            f_infect = false; f_immune = false; f_mig = false;
            if (any_lane(rn < p_accept)) {
                sus_add(si, 1.0);
                I += (lane == hi ? 1.0 : 0.0);
#pragma unroll
                for (int q = 0; q < NPR; ++q) {
                    const double d1 = (lane + 64 * q == tpi) ? 1.0 : 0.0;
                    totS[q] -= d1; totI[q] += d1;
                }
                gI += 1.0;
                f_infect = true; f_immune = true; f_mig = true;
                ev_type = EV_MIGRATION;
            }
        }
        // ---- UpdateRates(u_pi, f_infect, f_immune, f_mig) (pyx:516-546): u_pi is the current row ----
        if (f_infect) { inP = refresh_row(); pop_set<NPR>(infectP, u_pi, inP, lane); }
        if (f_immune) { imP = immune_sum(); pop_set<NPR>(immuneP, u_pi, imP, lane); }
        if (f_infect || f_immune) {
            pop_set<NPR>(popRate, u_pi, inP + imP, lane);
            rescan_pop();
        }
        if (f_mig) remig();
        // ---- AddEvent (events.pxi:37-44) and the counters ----
        const double den = Rtot;       // the denominator of this iteration's time step
        Rtot = totalRate + totalMig;
        log_event(ev_type, ev_hap, ev_pop, ev_nh, ev_np, den);
        return u_pi;
    }
};

template <int NPR, bool CLOCK, bool RCPDIV, int NT, bool TINY, bool UNIT>
static __device__ __forceinline__ void solo_body() {
    const SoloKA ka = (SoloKA)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &a = ka->a;
    const auto &sa = ka->sa;
    const int64_t rep = blockIdx.x;
    if (rep >= a.n_replicates) return;
    const int lane = threadIdx.x;
    const auto &p = a.p;
    const auto &r = a.r;
    const int P = uni_i32(p.P), H = uni_i32(p.H), S = uni_i32(p.S), sites = uni_i32(p.sites);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgxSoloLayout L = vgx_solo_layout(P, H, S, sites, sa.mig_in_lds);

    Solo<NPR, CLOCK, RCPDIV, NT, TINY, UNIT> c;
    constexpr int NTT = NT > 0 ? NT : 1;
    c.P = P; c.H = H; c.S = S; c.sites = sites; c.nseg = uni_i32(sa.nseg); c.lane = lane; c.l15 = lane & 15;
    c.small = uni_i32((P <= 16 && H <= 16) ? 1 : 0);
    c.maxterms = uni_i32(sa.maxterms);
    c.one_cls = uni_i32((NT > 0 && sa.n_cls == 1) ? 1 : 0);
    c.recomb = uni_i32(p.recombination != 0.0 ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) c.M.m[k] = (lane & 15) >= k ? 1.0 : 0.0;
    c.ldRng = (double *)(smem + L.rng); c.ldStage = (uint32_t *)(smem + L.stage);
    c.ldRngK = (uint64_t *)(smem + L.rngk); c.ldRngS = (uint64_t *)(smem + L.rngs); c.ldCold = (int64_t *)(smem + L.cold);
    c.rowI = (double *)(smem + L.rowI); c.rowCum = (double *)(smem + L.rowCum); c.rowHpr = (double *)(smem + L.rowHpr);
    c.rowBirth = (double *)(smem + L.rowBirth); c.rowTE = (double *)(smem + L.rowTE);
    c.susS = (double *)(smem + L.susS); c.susSt = (double *)(smem + L.susSt); c.susImm = (double *)(smem + L.susImm);
    c.ldSigma = (double *)(smem + L.sigma); c.ldTrans = (double *)(smem + L.trans);
    c.ldMrate = (double *)(smem + L.mrate); c.ldHmt = (double *)(smem + L.hmt);
    c.ldCd = (double *)(smem + L.cd); c.ldAs = (double *)(smem + L.as); c.ldSmult = (double *)(smem + L.smult);
    c.ldMig = (double *)(smem + (L.mig >= 0 ? L.mig : 0));
    c.mig_lds = uni_i32(L.mig >= 0 ? 1 : 0);
    c.gMig = p.mig;
    c.gEff = r.effMig + rep * P * P;

    // ---- parameters ----
    c.bh = 0.0; c.dh = 0.0; c.sh = 0.0; c.tmh = 0.0; c.stype = 0; c.path = 0u; c.hapClsLane = 0;
    if (lane < H) {
        const int cl = p.cls[lane];
        c.bh = p.bRate[lane]; c.dh = p.c_d[cl]; c.sh = p.c_s[cl]; c.tmh = p.c_tm[cl];
        c.stype = (int)p.suscType[lane];
        if (NT > 0) {
            c.hapClsLane = 16 * sa.hap_cls[lane];
        } else {
            for (int s = 0; s < sa.nseg; ++s)
                if (p.susc[lane * S + sa.seg_sn[s]] == sa.seg_sig[s]) c.path |= 1u << s;
        }
    }
    c.sgsn = 0; c.sgsig = 0.0; c.sgpar = -1; c.hseg = -1;
    c.passA0 = -1; c.passB0 = -1; c.passA1 = -1; c.passB1 = -1; c.npass0 = 0; c.npass1 = 0;
    if (NT == 0 && c.small) {
        if (lane < sa.nseg) { c.sgsn = sa.seg_sn[lane]; c.sgsig = sa.seg_sig[lane]; }
    } else if (NT == 0) {
        c.nseg = uni_i32(sa.tnseg);
        if (lane < sa.tnseg) { c.sgsn = sa.tseg_sn[lane]; c.sgsig = sa.tseg_sig[lane]; c.sgpar = sa.tseg_par[lane]; }
        if (lane < H) c.hseg = sa.cb_seg[p.c_bidx[p.cls[lane]]];
        c.passA0 = sa.pass[0 * VGX_SOLO_MAX_PASS + lane]; c.passB0 = sa.pass[1 * VGX_SOLO_MAX_PASS + lane];
        c.passA1 = sa.pass[2 * VGX_SOLO_MAX_PASS + lane]; c.passB1 = sa.pass[3 * VGX_SOLO_MAX_PASS + lane];
        c.npass0 = uni_i32(sa.npass0); c.npass1 = uni_i32(sa.npass1);
    }
    c.cumul_l = c.l15 < S ? p.suscepCumul[c.l15] : 0.0;
    c.no_imm = any_lane(c.cumul_l != 0.0) ? 0 : 1;
    c.sigcs = 0.0;
#pragma unroll
    for (int t = 0; t < NTT; ++t) { c.tlS[t] = 0.0; c.tlSig[t] = 0.0; c.tlM[t] = 0.0; c.tlCd[t] = 0.0; c.tlAs[t] = 1.0; c.tlRcp[t] = 1.0; c.tlSn[t] = 0; c.tlPn[t] = 0; }
    if (NT > 0) {
        const int cl = c.one_cls ? 0 : lane >> 4;               // the class of this lane's row
        if (cl < sa.n_cls && c.l15 < S) c.sigcs = sa.cls_sigma[cl * VGX_SOLO_MAX_S + c.l15];
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int k = 16 * t + c.l15, o = k / P, pn = k - o * P;
            if (cl < sa.n_cls && o < sa.cls_nnz[cl]) {
                c.tlSn[t] = sa.cls_tsn[cl * VGX_SOLO_MAX_S + o];
                c.tlSig[t] = sa.cls_tsig[cl * VGX_SOLO_MAX_S + o];
                c.tlPn[t] = pn;
                c.tlAs[t] = p.actualSizes[pn];
                c.tlRcp[t] = sa.rcpAs[pn];
            }
        }
    }
    for (int i = lane; i < H * S; i += 64) c.ldSigma[i] = p.susc[i];
    for (int i = lane; i < S * S; i += 64) c.ldTrans[i] = p.suscepTransition[i];
    for (int i = lane; i < H * sites; i += 64) c.ldMrate[i] = p.mRate[i];
    for (int i = lane; i < H * sites * 3; i += 64) c.ldHmt[i] = p.hapMutType[i];
    if (c.mig_lds)
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
        c.totS[q] = in ? (double)gI[PI_TOTSUS * P + pn] : 0.0;
        c.totI[q] = in ? (double)gI[PI_TOTINF * P + pn] : 0.0;
        const bool on = in && gI[PI_LOCK * P + pn] != 0;
        const double sz = in ? (double)p.sizes[pn] : 0.0;
        const double thrOn = in ? p.startLD[pn] * sz : 0.0, thrOff = in ? p.endLD[pn] * sz : 0.0;
        c.sgnLD[q] = on ? -1.0 : 1.0;
        c.thrCur[q] = on ? thrOff : thrOn;
        // a population can switch on only if its threshold lies below its size, off only if it is on
        ldp = ldp || (in && (thrOn < sz || on));
        if (in) { c.ldCd[pn] = c.cd[q]; c.ldAs[pn] = c.asz[q]; c.ldSmult[pn] = p.sampMult[pn]; }
    }
    c.ld_any = any_lane(ldp) ? 1 : 0;

    VgxRepScalars *sc = r.sc + rep;
    c.currentTime = sc->currentTime; c.totalRate = 0.0; c.totalMig = 0.0; c.Rtot = 0.0;
    c.gI = (double)uni_i64(sc->globalInfectious);
    {
        const int64_t v = lane == 0 ? sc->bCounter : lane == 1 ? sc->dCounter : lane == 2 ? sc->sCounter : lane == 3 ? sc->mCounter
                        : lane == 4 ? sc->iCounter : lane == 5 ? sc->migPlus : lane == 6 ? sc->migNonPlus : lane == 7 ? sc->swapLockdown : 0;
        c.cnt = (uint64_t)v;
    }
    c.zero_w = 0ull;
#ifdef VGX_PROFILE
    c.prof_acc = 0ull; c.prof_t0 = __builtin_readcyclecounter();
#endif
    c.stage_n = 0;
    c.pos = 32; c.u_pre = 0.0; c.n_pre = 0.0;
    c.ev_left = 0; c.loop_left = 0; c.s_left = 0; c.ev_left0 = 0; c.loop_left0 = 0; c.iter_base = 0u;
    c.cur = -1;
    c.has_mig = 1;
    c.I = 0.0; c.cum = 0.0; c.hpr = 0.0; c.birth = 0.0; c.tE = 0.0; c.e1 = 0.0; c.e2 = 0.0; c.sm = 0.0;
    c.Ssus = 0.0; c.Sst = 0.0; c.imms = 0.0; c.Sseg = 0.0;
#pragma unroll
    for (int q = 0; q < NPR; ++q) c.mrow[q] = 0.0;
    if (lane == 0) {
        c.ldCold[C_EV_PTR] = sc->ev_ptr; c.ldCold[C_LOOPS] = 0; c.ldCold[C_ATT_LOOPS] = 0; c.ldCold[C_LOC_N] = 0; c.ldCold[C_TRAJ_NEXT] = 0;
        c.ldCold[C_FA_N] = 0; c.ldCold[C_ATT_EV0] = sc->ev_ptr; c.ldCold[C_ATT_LOC0] = 0; c.ldCold[C_RESTARTS] = 0; c.ldCold[C_ATT] = 0;
        c.ldCold[C_GOOD] = sc->good_attempt; c.ldCold[C_LAST_ATT] = -1; c.ldCold[C_REC_N] = 0;
    }
    {   // PCG64 jump constants of this lane: a^(lane+1), sum_{j<=lane} a^j
        const uint64_t MH = 0x2360ED051FC65DA4ull, ML = 0x4385DF649FCCF645ull;
        uint64_t Ah = MH, Al = ML, Gh = 0, Gl = 1;
        for (int j = 1; j < 64; ++j) {
            uint64_t nh, nl, gh, gl;
            vgx_mul128(Ah, Al, MH, ML, nh, nl);
            vgx_mul128(Gh, Gl, MH, ML, gh, gl);
            vgx_add128(gh, gl, 0, 1);
            if (j <= lane) { Ah = nh; Al = nl; Gh = gh; Gl = gl; }
        }
        c.ldRngK[lane * 4 + 0] = Ah; c.ldRngK[lane * 4 + 1] = Al; c.ldRngK[lane * 4 + 2] = Gh; c.ldRngK[lane * 4 + 3] = Gl;
    }
    WSYNC();

    const bool has_tlimit = !(a.time == -1.0f);
    const double tlimit = has_tlimit ? (double)a.time : __builtin_inf();
    c.tlimit = tlimit;
    c.next_tg = (CLOCK && r.traj != nullptr && r.traj_points > 0) ? r.traj_t0 : __builtin_inf();
    const bool has_traj = r.traj != nullptr;
    const int record_events = a.record_events;
    int error = 0;

    // PrepareParameters tail (pyx:449-451): CheckLockdown for every population, UpdateAllRates
    if (c.ld_any) c.check_lockdowns(ka, rep, 0, P);
    c.rebuild_all(ka);

    for (int64_t att = 0; att < a.attempts && !error; ++att) {   // pyx:399-418
        {
            VgxPcg64 s;
            vgx_pcg64_seed(s, (uint64_t)r.seeds[rep], (uint32_t)att);
            if (lane == 0) {
                c.ldRngS[0] = s.sh; c.ldRngS[1] = s.sl; c.ldRngS[2] = s.ih; c.ldRngS[3] = s.il;
                c.ldCold[C_ATT] = att; c.ldCold[C_LAST_ATT] = att; c.ldCold[C_ATT_LOOPS] = 0;
            }
            WSYNC();
        }
        c.pos = 32;
        if (any_lane(c.Rtot != 0.0) && any_lane(c.gI != 0.0)) {
            bool done = false;
            while (!done) {
                // ---- a segment of the event loop: at most 2^30 iterations on 32-bit countdowns ----
                int64_t ev_ptr = c.cold_get(C_EV_PTR), loops = c.cold_get(C_LOOPS), att_loops = c.cold_get(C_ATT_LOOPS);
                {
                    const int64_t cS = (int64_t)bcast_i64((int64_t)c.cnt, EV_SAMPLING);
                    const bool go = ev_ptr < a.ev_size && (a.sample_size == -1 || cS <= a.sample_size) && (!has_tlimit || any_lane(c.currentTime < tlimit));
                    if (!go) break;
                    if (loops >= a.max_loop) { error = ERR_LOOP_GUARD; break; }
                    c.ev_left0 = (int)min((int64_t)SOLO_BIG, a.ev_size - ev_ptr);
                    c.loop_left0 = (int)min((int64_t)SOLO_BIG, a.max_loop - loops);
                    c.s_left = a.sample_size == -1 ? SOLO_BIG : (int)min((int64_t)SOLO_BIG, a.sample_size - cS + 1);
                    c.ev_left = c.ev_left0; c.loop_left = c.loop_left0;
                    c.iter_base = (uint32_t)att_loops + (uint32_t)c.loop_left0;
                }
                for (;;) {
                    double u_slow = 0.0;
                    int pi = 0;
                    const int why = (UNIT && c.no_imm && !c.has_mig && c.cur == 0) ? c.unit_fast_loop(u_slow, pi) : c.fast_loop(u_slow, pi);
                    if (why == c.FAST_END) break;
                    if (why == c.FAST_REFILL) {
                        // 64 PCG64 outputs: lane k jumps the stream k + 1 steps ahead (exact 128-bit arithmetic); even outputs are the
                        // uniforms of SampleTime (pyx:477), odd ones those of GenerateEvent (pyx:488)
                        WSYNC();
                        const uint64_t Ah = c.ldRngK[lane * 4 + 0], Al = c.ldRngK[lane * 4 + 1], Gh = c.ldRngK[lane * 4 + 2], Gl = c.ldRngK[lane * 4 + 3];
                        const uint64_t sh = c.ldRngS[0], sl = c.ldRngS[1], ih = c.ldRngS[2], il = c.ldRngS[3];
                        uint64_t h, l, ch, cl;
                        vgx_mul128(Ah, Al, sh, sl, h, l);
                        vgx_mul128(Gh, Gl, ih, il, ch, cl);
                        vgx_add128(h, l, ch, cl);
                        double u = vgx_pcg64_output_double(h, l);
                        // the counter-based stream (vgx_run_opts.mode = 2: this kernel's exact arithmetic on other random numbers): iteration
                        // i of the attempt takes outputs 2 i and 2 i + 1 of the stream of (seed, attempt), as the host clock reads them
                        if (a.rng_philox)
                            u = vgx_philox_stream_double((uint64_t)r.seeds[rep], (uint32_t)att,
                                                         2 * (uint64_t)(att_loops + (int64_t)(c.loop_left0 - c.loop_left)) + (uint64_t)lane);
                        WSYNC();
                        c.ldRng[lane] = (CLOCK && !(lane & 1)) ? -vgx_log(u) : u;
                        if (lane == 63) { c.ldRngS[0] = h; c.ldRngS[1] = l; }
                        c.pos = 0;
                        WSYNC();
                        c.prefetch_uniforms();
                        continue;
                    }
                    if (why == c.FAST_SLOW) {
                        // any other event (or a trajectory grid point first): the general form of one iteration
                        c.loop_left -= 1;
                        if (CLOCK) {
                            const double t_new = c.currentTime + (c.n_pre / c.Rtot);   // SampleTime pyx:476-478
                            if (has_traj) {
                                c.traj_emit(ka, rep, t_new, false);
                                const int64_t tn = c.cold_get(C_TRAJ_NEXT);
                                c.next_tg = tn < r.traj_points ? r.traj_t0 + (double)tn * r.traj_dt : __builtin_inf();
                            }
                            c.currentTime = t_new;
                        }
                        c.pos = uni_i32(c.pos + 1);
                        c.prefetch_uniforms();
                        pi = c.event(u_slow);
                        MARK("event_end");
                    }
                    // after an iteration: a full stage, a zero weight, extinction (pyx:410-411), CheckLockdown (pyx:412)
                    if (c.stage_n == 64) {
                        if (record_events) c.stage_flush(ka, rep);
                        c.stage_n = 0;
                    }
                    if (c.zero_w) { done = true; break; }
                    if (any_lane(c.totalRate == 0.0) || any_lane(c.gI == 0.0)) { done = true; break; }
                    if (c.ld_any) {
                        bool cross = false;
#pragma unroll
                        for (int q = 0; q < NPR; ++q) cross = cross || (lane + 64 * q == pi && (c.totI[q] - c.thrCur[q]) * c.sgnLD[q] > 0.0);
                        if (any_lane(cross)) {
                            if (c.check_lockdowns(ka, rep, pi, pi + 1)) c.rebuild_all(ka);
                            if (c.zero_w) { done = true; break; }
                        }
                    }
                }
                // ---- end of the segment: the 64-bit bookkeeping ----
                if (record_events) c.stage_flush(ka, rep);
                c.stage_n = 0;
                if (lane == 0) {
                    c.ldCold[C_EV_PTR] = ev_ptr + (c.ev_left0 - c.ev_left);
                    c.ldCold[C_LOOPS] = loops + (c.loop_left0 - c.loop_left);
                    c.ldCold[C_ATT_LOOPS] = att_loops + (c.loop_left0 - c.loop_left);
                }
                c.ev_left0 = 0; c.ev_left = 0; c.loop_left0 = 0; c.loop_left = 0;
                WSYNC();
            }
        }
        if (c.zero_w) { error = (c.zero_w >> 32) == 2ull ? ERR_CAPACITY : ERR_ZERO_WEIGHT; }
        if (error) break;
        const int64_t ev_ptr = c.cold_get(C_EV_PTR);
        if (ev_ptr <= 100 && a.iterations > 100) {
            // Restart (pyx:714-738); swapLockdown survives.  Lockdown records of the failed attempt stay in the log: keep the
            // (rate, iteration) pairs the host clock needs for them
            const int64_t loc_n = c.cold_get(C_LOC_N), att_loc0 = c.cold_get(C_ATT_LOC0), att_ev0 = c.cold_get(C_ATT_EV0);
            int64_t fa_n = c.cold_get(C_FA_N);
            if (loc_n > att_loc0 && a.record_events && r.fa_cap > 0) {
                const int64_t n = ev_ptr - att_ev0;
                for (int64_t k = lane; k < n; k += 64) {
                    const int64_t slot = att_ev0 + k - r.ev_base;
                    if (fa_n + k < r.fa_cap && slot >= 0 && slot < r.evcap) {
                        r.fa_rate[rep * r.fa_cap + fa_n + k] = r.ev_rate[rep * r.evcap + slot];
                        r.fa_key[rep * r.fa_cap + fa_n + k] = (att << 40) | (int64_t)(uint32_t)r.ev_cols[(rep * r.evcap + slot) * VGX_EV_COLS + 5];
                    }
                }
                fa_n += n;
            }
            if (lane == 0) {
                c.ldCold[C_FA_N] = fa_n; c.ldCold[C_ATT_EV0] = 0; c.ldCold[C_EV_PTR] = 0; c.ldCold[C_TRAJ_NEXT] = 0;
                c.ldCold[C_ATT_LOC0] = loc_n; c.ldCold[C_RESTARTS] = c.ldCold[C_RESTARTS] + 1;
                c.ldCold[C_ATT] = att + 1; c.ldCold[C_ATT_LOOPS] = 0;   // the CheckLockdown below belongs to the next attempt, before its first iteration
            }
            c.cnt = lane == CNT_SWAP ? c.cnt : 0ull;
            c.currentTime = 0.0;
            c.next_tg = (CLOCK && has_traj && r.traj_points > 0) ? r.traj_t0 : __builtin_inf();
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
            if (c.ld_any) c.check_lockdowns(ka, rep, 0, P);
            c.rebuild_all(ka);
        } else {
            if (lane == 0) c.ldCold[C_GOOD] = att + 1;
            WSYNC();
            break;
        }
    }
    if (has_traj) c.traj_emit(ka, rep, 0.0, true);

    // ---- end state back in the layout of the other direct kernels ----
    c.row_store();
    for (int pn = 0; pn < P; ++pn) {
        int32_t *lh = r.lhap + (rep * P + pn) * r.cap, *lc = r.lcls + (rep * P + pn) * r.cap;
        int64_t *ln = r.lcnt + (rep * P + pn) * r.cap, *lt = r.ltsum + (rep * P + pn) * r.capT;
        const double v = lane < H ? c.rowI[pn * H + lane] : 0.0;
        const unsigned long long nz = __builtin_amdgcn_ballot_w64(v != 0.0);
        const int n = __builtin_popcountll(nz);
        const int pos = __builtin_popcountll(nz & ((1ull << lane) - 1ull));
        if (n > r.cap) error = ERR_CAPACITY;
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
            gI[PI_LOCK * P + pn] = c.sgnLD[q] < 0.0 ? 1 : 0;
        }
    }
    for (int i = lane; i < P * S; i += 64) {
        r.sus[rep * P * S + i] = (int64_t)c.susS[i];
        r.immSrc[rep * P * S + i] = c.susImm[i];
    }
#ifdef VGX_PROFILE
    if (lane < VGX_PROF_SLOTS && r.prof) r.prof[rep * VGX_PROF_SLOTS + lane] = c.prof_acc;
#endif
    {
        const int64_t cB = bcast_i64((int64_t)c.cnt, 0), cD = bcast_i64((int64_t)c.cnt, 1), cS = bcast_i64((int64_t)c.cnt, 2), cM = bcast_i64((int64_t)c.cnt, 3);
        const int64_t cI = bcast_i64((int64_t)c.cnt, 4), cMigP = bcast_i64((int64_t)c.cnt, 5), cMigN = bcast_i64((int64_t)c.cnt, 6), cSwap = bcast_i64((int64_t)c.cnt, 7);
        if (lane == 0) {
            sc->currentTime = c.currentTime; sc->totalRate = c.totalRate; sc->totalMig = c.totalMig;
            sc->globalInfectious = (int64_t)c.gI;
            sc->bCounter = cB; sc->dCounter = cD; sc->sCounter = cS; sc->mCounter = cM; sc->iCounter = cI;
            sc->swapLockdown = cSwap; sc->migPlus = cMigP; sc->migNonPlus = cMigN;
            sc->good_attempt = c.ldCold[C_GOOD];
            sc->ev_ptr = c.ldCold[C_EV_PTR]; sc->loop_iterations = c.ldCold[C_LOOPS]; sc->restarts = c.ldCold[C_RESTARTS];
            sc->loc_n = c.ldCold[C_LOC_N]; sc->error = error; sc->traj_next = c.ldCold[C_TRAJ_NEXT];
            sc->last_attempt = c.ldCold[C_LAST_ATT]; sc->last_attempt_loops = c.ldCold[C_ATT_LOOPS];
            sc->rec_n = c.ldCold[C_REC_N];   // (like upstream's `rec`, records of failed attempts stay: Restart does not clear them)
            sc->fa_n = c.ldCold[C_FA_N];
        }
    }
}

}  // namespace

// kernel <-> (population registers, device clock, reciprocal division, compact layout's term registers, tiny shape)
#ifndef VGX_SOLO_WAVES
#define VGX_SOLO_WAVES 2     // wavefronts per SIMD the register allocation aims at.  Measured (tools/probe_solo_ens.py, Table-3 model, 16 384
                             // replicates): 2 costs a lone wavefront nothing (the compact kernels need 243-256 registers anyway) and lets
                             // ensembles of the 10-deme model run two wavefronts per SIMD (8.1e8 -> 1.37e9 events/s); 3 and 4 spill 100-300
                             // registers into the event loop and lose (K = 2: 1.68e9 / 1.31e9 / 1.06e9 at 2 / 3 / 4)
#endif
#define SOLO_KERNEL(name, NPR, CLOCK, RCPDIV, NT, TINY, UNIT) \
    extern "C" __global__ void __launch_bounds__(64, VGX_SOLO_WAVES) name(VgxSoloKArgs) { solo_body<NPR, CLOCK, RCPDIV, NT, TINY, UNIT>(); }
SOLO_KERNEL(vgx_solo_kernel_unit, 1, false, true, 1, true, true)
SOLO_KERNEL(vgx_solo_kernel_tiny, 1, false, true, 1, true, false)
SOLO_KERNEL(vgx_solo_kernel_c1, 1, false, true, 1, false, false)
SOLO_KERNEL(vgx_solo_kernel_c2, 1, false, true, 2, false, false)
SOLO_KERNEL(vgx_solo_kernel_p64, 1, false, true, 0, false, false)
SOLO_KERNEL(vgx_solo_kernel_p128, 2, false, true, 0, false, false)
SOLO_KERNEL(vgx_solo_kernel_c1_clock, 1, true, true, 1, false, false)
SOLO_KERNEL(vgx_solo_kernel_p64_clock, 1, true, true, 0, false, false)
SOLO_KERNEL(vgx_solo_kernel_p128_clock, 2, true, true, 0, false, false)
// validation: x / actualSizes as the compiler's division instead of the reciprocal sequence (VGX_SOLO_PLAIN_DIV=1; calls with an event log
// and no time limit only)
SOLO_KERNEL(vgx_solo_kernel_tiny_plaindiv, 1, false, false, 1, true, false)
SOLO_KERNEL(vgx_solo_kernel_c2_plaindiv, 1, false, false, 2, false, false)
SOLO_KERNEL(vgx_solo_kernel_p64_plaindiv, 1, false, false, 0, false, false)
SOLO_KERNEL(vgx_solo_kernel_p128_plaindiv, 2, false, false, 0, false, false)

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_solo(const VgxDirectArgs *a, const VgxSoloArgs *sa, int clock,
                                                                            hipStream_t stream) {
    const VgxSoloLayout L = vgx_solo_layout(a->p.P, a->p.H, a->p.S, a->p.sites, sa->mig_in_lds);
    const bool tiny = sa->compact == 1 && a->p.P <= 4 && a->p.S <= 4 && sa->maxterms <= 4;
    void (*k)(VgxSoloKArgs);
    if (a->p.P > 64) k = clock ? vgx_solo_kernel_p128_clock : sa->exact_rcp_div ? vgx_solo_kernel_p128 : vgx_solo_kernel_p128_plaindiv;
    else if (clock) k = sa->compact == 1 ? vgx_solo_kernel_c1_clock : vgx_solo_kernel_p64_clock;
    else if (!sa->exact_rcp_div) k = tiny ? vgx_solo_kernel_tiny_plaindiv : sa->compact == 2 ? vgx_solo_kernel_c2_plaindiv : vgx_solo_kernel_p64_plaindiv;
    else if (tiny && a->p.P == 1 && a->p.H == 1 && a->p.S == 1 && sa->maxterms <= 1 && !getenv("VGX_SOLO_NO_UNIT")) k = vgx_solo_kernel_unit;
    else k = tiny ? vgx_solo_kernel_tiny : sa->compact == 1 ? vgx_solo_kernel_c1 : sa->compact == 2 ? vgx_solo_kernel_c2 : vgx_solo_kernel_p64;
    hipError_t err = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (err != hipSuccess) return err;
    VgxSoloKArgs ka;
    ka.a = *a; ka.sa = *sa;
    if (getenv("VGX_TIMING")) {   // diagnostics: wavefronts of this instantiation a CU holds
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k, 64, (size_t)L.total) == hipSuccess)
            fprintf(stderr, "vgx_solo: LDS %d B per wavefront, %d wavefronts per CU\n", (int)L.total, nb);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)a->n_replicates), dim3(64), (size_t)L.total, stream, ka);
    return hipGetLastError();
}

// ---- test hook: the two division forms of this file against the division, on the device ----
extern "C" __global__ void vgx_solo_divtest_kernel(const double *n, const double *b, double *q_seq, double *q_lean, double *q_div, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double y = 1.0 / b[i];
    q_seq[i] = div_by_const(n[i], b[i], y);
    q_lean[i] = fdiv(n[i], b[i]);
    q_div[i] = n[i] / b[i];
}
extern "C" int vgx_test_div_by_const(const double *n, const double *b, int64_t count, double *q_seq, double *q_lean, double *q_div) {
    if (!n || !b || !q_seq || !q_lean || !q_div || count < 0) return 1;
    if (count == 0) return 0;
    double *d = nullptr;
    const size_t c = (size_t)(count > 0 ? count : 1);
    if (hipMalloc((void **)&d, c * 40) != hipSuccess) return 2;
    int rc = 0;
    if (hipMemcpy(d, n, (size_t)count * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + c, b, (size_t)count * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = 2;
    if (!rc) {
        hipLaunchKernelGGL(vgx_solo_divtest_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, d, d + c, d + 2 * c, d + 3 * c, d + 4 * c, count);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(q_seq, d + 2 * c, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(q_lean, d + 3 * c, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(q_div, d + 4 * c, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = 2;
    }
    (void)hipFree(d);
    return rc;
}
