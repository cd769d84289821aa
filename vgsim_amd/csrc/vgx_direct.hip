// vgx_direct.hip — persistent direct-Gillespie kernel for gfx950 (MI355X).
//
// Replaces the event loop of BirthDeathModel.SimulatePopulation (reference src/_BirthDeath.pyx:396-429)
// and everything it calls: SampleTime pyx:476, GenerateEvent pyx:483, UpdateRates pyx:516,
// ImmunityTransition pyx:550, Birth pyx:568, Death/Sampling pyx:616/630, Mutation pyx:640,
// GenerateMigration pyx:672, CheckLockdown pyx:698, Restart pyx:714, UpdateAllRates pyx:279,
// fastChoose / fastChoose_skip fast_choose.pxi:18/36, Events.AddEvent events.pxi:37.
//
// Execution model: ONE 64-lane wavefront per replicate (workgroup = 1 wave, grid = replicates), alive for
// the whole simulate call.  The trajectory itself is sequential (event k+1 depends on event k), so the
// wave exploits the parallelism INSIDE an event: lanes map to populations (P-wide arrays live in LDS),
// to the entries of the chosen population's ordered occupancy list (coalesced 16 B/entry stream from
// HBM/L2), and to the terms of the transmission sum.  Everything that the reference sums left-to-right
// is summed left-to-right here too (seq_sum / seq_scan: a lane-ordered chain of f64 adds), with no FMA
// contraction (-ffp-contract=off), so every rate, every comparison and the recycled random number are
// bit-identical to the reference's; only haplotypes with a non-zero count are visited, which is exact
// because x + 0.0 == x and a zero weight can never stop a scan.
//
// Per-haplotype rates are factored through CLASSES of identical parameter rows (bRate, susceptibility
// row, dRate, sRate, sum of mRate): BirthRate (pyx:382-392) depends on the haplotype only through its
// class, so it is evaluated once per class and population instead of once per haplotype — same
// operations on the same operands, hence the same bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"

#define LANES VGX_WAVE

enum { ERR_ZERO_WEIGHT = 3, ERR_CAPACITY = 4, ERR_LOOP_GUARD = 5 };
enum { EV_BIRTH = 0, EV_DEATH, EV_SAMPLING, EV_MUTATION, EV_SUSCCHANGE, EV_MIGRATION };

// compiler-level ordering of cross-lane traffic through LDS / global memory inside the wave.  The
// hardware executes one wave's LDS and vector-memory instructions in issue order, so no wait is needed.
#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

static __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

static __device__ __forceinline__ double bcast(double v, int k) {  // value of lane k (k wave-uniform)
    int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ int64_t bcast_i64(int64_t v, int k) {
    int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, k);
    int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), k);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// acc + v[0] + v[1] + ... + v[n-1], added strictly in lane order (the reference's loop order).
static __device__ __forceinline__ double seq_sum(double v, int n, double acc) {

    for (int k = 0; k < n; ++k) acc += bcast(v, k);
    return acc;
}
// lane L gets carry + v[0] + ... + v[min(L, n-1)] with the same rounding sequence as the serial loop:
// adding +0.0 after lane L's own term is exact.
static __device__ __forceinline__ double seq_scan(double v, int n, double carry, int lane) {
    double acc = carry;

    for (int k = 0; k < n; ++k) {
        double wk = bcast(v, k);
        acc += (k <= lane) ? wk : 0.0;
    }
    return acc;
}
// inclusive integer scan across the wave (order-free: int64 addition is associative)
static __device__ __forceinline__ int64_t iscan(int64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < LANES; d <<= 1) {
        int64_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

struct Ctx {
    // dims
    int P, S, H, C, CB, sites, lane;
    // parameters
    const VgxDevParams *p;
    // LDS
    double *popRate, *infectPopRate, *immunePopRate, *migPopRate, *maxEBM, *cd, *as, *immSrc, *tE;
    int64_t *totalSus, *totalInf, *lockON, *sus;
    // global, this replicate
    double *birthC, *xC, *effMig;
    int32_t *nocc, *lhap, *lcls;
    int64_t *lcnt;
    int64_t cap;
    double *ev_time;
    int32_t *ev_cols;
    int32_t *loc_rec;
    double *loc_time;
    double *traj;
    int64_t traj_points, traj_next;
    double traj_t0, traj_dt;
    int64_t evcap, ev_base;
    int record_events;
    // scalars
    double currentTime, totalRate, totalMig, rn;
    int64_t gI;
    int64_t bC, dC, sC, mC, iC, swapLD, migPlus, migNon;
    int64_t ev_ptr, ev_size, loc_n;
    int error;
};

// ------------------------------------------------------------------------------------------------
// fastChoose restated (fast_choose.pxi:18-31) for short arrays: every lane runs the same serial loop.
template <typename LoadW>
static __device__ __forceinline__ int choose_serial(Ctx &c, LoadW w, int n, double tw, double &rn) {
    double r = tw * rn;
    int i = 0;
    double total = w(0);
    while (total < r && i < n - 1) {
        i += 1;
        total += w(i);
    }
    double wi = w(i);
    if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
    rn = (r - (total - wi)) / wi;
    return i;
}
template <typename LoadW>
static __device__ __forceinline__ int choose_serial_i64(Ctx &c, LoadW w, int n, int64_t tw, double &rn) {
    double r = (double)tw * rn;
    int i = 0;
    int64_t total = w(0);
    while ((double)total < r && i < n - 1) {
        i += 1;
        total += w(i);
    }
    int64_t wi = w(i);
    if (wi == 0) c.error = ERR_ZERO_WEIGHT;
    rn = (r - (double)(total - wi)) / (double)wi;
    return i;
}

// fastChoose over a P-sized f64 array held in LDS: lane-ordered prefix sums, one ballot per 64 entries.
static __device__ int choose_lds(Ctx &c, const double *w, int n, double tw, double &rn) {
    const int lane = c.lane;
    double r = tw * rn;
    double carry = 0.0;
    for (int base = 0; base < n; base += LANES) {
        int k = base + lane;
        int nn = min(LANES, n - base);
        double wk = (k < n) ? w[k] : 0.0;
        double pre = seq_scan(wk, nn, carry, lane);
        unsigned long long hit = __ballot((k < n) && !(pre < r));
        if (hit) {
            int j = __ffsll((long long)hit) - 1;
            double total = bcast(pre, j), wi = bcast(wk, j);
            if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
            rn = (r - (total - wi)) / wi;
            return base + j;
        }
        carry = bcast(pre, LANES - 1);
    }
    // ran to the end: the reference loop clamps at n-1 (fast_choose.pxi:26)
    double wi = w[n - 1];
    if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
    rn = (r - (carry - wi)) / wi;
    return n - 1;
}

// ------------------------------------------------------------------------------------------------
// Occupancy list of a population: ordered (hap, class, count) entries.
static __device__ __forceinline__ int32_t *LH(Ctx &c, int pi) { return c.lhap + (int64_t)pi * c.cap; }
static __device__ __forceinline__ int32_t *LC(Ctx &c, int pi) { return c.lcls + (int64_t)pi * c.cap; }
static __device__ __forceinline__ int64_t *LN(Ctx &c, int pi) { return c.lcnt + (int64_t)pi * c.cap; }

// first index whose haplotype is >= target (n if none); *found tells whether it is the target itself
static __device__ int list_lower_bound(Ctx &c, int pi, int target, bool &found) {
    const int n = c.nocc[pi];
    const int32_t *lh = LH(c, pi);
    for (int base = 0; base < n; base += LANES) {
        int k = base + c.lane;
        int h = (k < n) ? lh[k] : 0x7fffffff;
        unsigned long long ge = __ballot(h >= target);
        if (ge) {
            int j = __ffsll((long long)ge) - 1;
            int hj = __builtin_amdgcn_readlane(h, j);
            found = (hj == target);
            return base + j;
        }
    }
    found = false;
    return n;
}

static __device__ void list_insert_at(Ctx &c, int pi, int pos, int hap, int cls, int64_t cnt) {
    int n = c.nocc[pi];
    if (n >= c.cap) { c.error = ERR_CAPACITY; return; }
    int32_t *lh = LH(c, pi), *lc = LC(c, pi);
    int64_t *ln = LN(c, pi);
    // shift [pos, n) one slot up, highest chunk first; each chunk is read completely before it is written
    for (int hi = n; hi > pos; hi -= LANES) {
        int lo = max(pos, hi - LANES);
        int k = lo + c.lane;
        bool ok = k < hi;
        int h = 0, cl = 0;
        int64_t ct = 0;
        if (ok) { h = lh[k]; cl = lc[k]; ct = ln[k]; }
        WSYNC();
        if (ok) { lh[k + 1] = h; lc[k + 1] = cl; ln[k + 1] = ct; }
        WSYNC();
    }
    if (c.lane == 0) { lh[pos] = hap; lc[pos] = cls; ln[pos] = cnt; c.nocc[pi] = n + 1; }
    WSYNC();
}

static __device__ void list_remove_at(Ctx &c, int pi, int pos) {
    int n = c.nocc[pi];
    int32_t *lh = LH(c, pi), *lc = LC(c, pi);
    int64_t *ln = LN(c, pi);
    for (int lo = pos + 1; lo < n; lo += LANES) {
        int k = lo + c.lane;
        bool ok = k < n;
        int h = 0, cl = 0;
        int64_t ct = 0;
        if (ok) { h = lh[k]; cl = lc[k]; ct = ln[k]; }
        WSYNC();
        if (ok) { lh[k - 1] = h; lc[k - 1] = cl; ln[k - 1] = ct; }
        WSYNC();
    }
    if (c.lane == 0) c.nocc[pi] = n - 1;
    WSYNC();
}

// infectious[pi, hap] += delta (delta = +1 / -1), keeping the list ordered and free of zero counts
static __device__ void list_add(Ctx &c, int pi, int hap, int64_t delta) {
    bool found;
    int pos = list_lower_bound(c, pi, hap, found);
    if (found) {
        int64_t *ln = LN(c, pi);
        int64_t v = ln[pos] + delta;  // uniform address
        if (v == 0) {
            list_remove_at(c, pi, pos);
        } else {
            if (c.lane == 0) ln[pos] = v;
            WSYNC();
        }
    } else {
        list_insert_at(c, pi, pos, hap, c.p->cls[hap], delta);
    }
}

// ------------------------------------------------------------------------------------------------
// BirthRate per class (pyx:382-392): ps += susceptHapPopRate * m * m * cd / as over (sn, pn), in order.
static __device__ void birth_update(Ctx &c, int pi) {
    const VgxDevParams &p = *c.p;
    const int P = c.P, S = c.S, lane = c.lane;
    const double *mrow = p.mig + (int64_t)pi * P;
    for (int cb = 0; cb < c.CB; ++cb) {
        double ps = 0.0;
        for (int sn = 0; sn < S; ++sn) {
            double x = (double)c.sus[pi * S + sn] * p.cb_sigma[cb * S + sn];
            if (lane == 0) c.xC[((int64_t)pi * c.CB + cb) * S + sn] = x;
            for (int base = 0; base < P; base += LANES) {
                int pn = base + lane;
                double t = 0.0;
                if (pn < P) {
                    double m = mrow[pn];
                    t = x * m * m * c.cd[pn] / c.as[pn];
                }
                ps = seq_sum(t, min(LANES, P - base), ps);
            }
        }
        if (lane == 0) c.birthC[(int64_t)pi * c.CB + cb] = p.cb_b[cb] * ps;
    }
    WSYNC();
}

// tEventHapPopRate per class (pyx:522-526) from the cached birth rates of population pi
static __device__ void tE_fill(Ctx &c, int pi) {
    const VgxDevParams &p = *c.p;
    const double mult = p.sampMult[pi];
    for (int base = 0; base < c.C; base += LANES) {
        int k = base + c.lane;
        if (k < c.C) {
            double b = c.birthC[(int64_t)pi * c.CB + p.c_bidx[k]];
            c.tE[k] = ((b + p.c_d[k]) + p.c_s[k] * mult) + p.c_tm[k];
        }
    }
    WSYNC();
}

// infectPopRate[pi] = sum over occupied haplotypes, in haplotype order, of tEvent * infectious (pyx:519-528)
static __device__ double row_sum(Ctx &c, int pi) {
    const int n = c.nocc[pi];
    const int32_t *lc = LC(c, pi);
    const int64_t *ln = LN(c, pi);
    double acc = 0.0;
    for (int base = 0; base < n; base += LANES) {
        int k = base + c.lane;
        double w = 0.0;
        if (k < n) w = c.tE[lc[k]] * (double)ln[k];
        acc = seq_sum(w, min(LANES, n - base), acc);
    }
    return acc;
}

// fastChoose(hapPopRate[pi], infectPopRate[pi], rn) over the occupied entries; returns the list index
static __device__ int row_select(Ctx &c, int pi, double tw, double &rn, double &w_sel) {
    const int n = c.nocc[pi];
    const int32_t *lc = LC(c, pi);
    const int64_t *ln = LN(c, pi);
    const int lane = c.lane;
    double r = tw * rn;
    double carry = 0.0;
    for (int base = 0; base < n; base += LANES) {
        int k = base + lane;
        int nn = min(LANES, n - base);
        double w = 0.0;
        if (k < n) w = c.tE[lc[k]] * (double)ln[k];
        double pre = seq_scan(w, nn, carry, lane);
        unsigned long long hit = __ballot((k < n) && !(pre < r));
        if (hit) {
            int j = __ffsll((long long)hit) - 1;
            double total = bcast(pre, j);
            w_sel = bcast(w, j);
            if (w_sel == 0.0) c.error = ERR_ZERO_WEIGHT;
            rn = (r - (total - w_sel)) / w_sel;
            return base + j;
        }
        carry = bcast(pre, LANES - 1);
    }
    // nothing reached r: the dense loop runs on to index H-1 (fast_choose.pxi:26); that is a valid pick
    // only if haplotype H-1 is occupied, otherwise the reference reports a zero weight
    if (n > 0 && LH(c, pi)[n - 1] == c.H - 1) {
        w_sel = c.tE[lc[n - 1]] * (double)ln[n - 1];
        rn = (r - (carry - w_sel)) / w_sel;
        return n - 1;
    }
    c.error = ERR_ZERO_WEIGHT;
    w_sel = 1.0;
    return 0;
}

// UpdateRates (pyx:516-546)
static __device__ void update_rates(Ctx &c, int pi, bool infect, bool immune, bool migration) {
    const int P = c.P, lane = c.lane;
    if (infect) {
        birth_update(c, pi);
        tE_fill(c, pi);
        double v = row_sum(c, pi);
        if (lane == 0) c.infectPopRate[pi] = v;
    }
    if (immune) {
        double v = 0.0;
        for (int sn = 0; sn < c.S; ++sn) v += c.immSrc[pi * c.S + sn];
        if (lane == 0) c.immunePopRate[pi] = v;
    }
    WSYNC();
    if (infect || immune) {
        if (lane == 0) c.popRate[pi] = c.infectPopRate[pi] + c.immunePopRate[pi];
        WSYNC();
        double tr = 0.0;
        for (int base = 0; base < P; base += LANES) {
            int pn = base + lane;
            double w = (pn < P) ? c.popRate[pn] : 0.0;
            tr = seq_sum(w, min(LANES, P - base), tr);
        }
        c.totalRate = tr;
    }
    if (migration) {
        double tm = 0.0;
        for (int base = 0; base < P; base += LANES) {
            int pn = base + lane;
            double w = 0.0;
            if (pn < P) {
                w = c.maxEBM[pn] * (double)c.totalSus[pn] * (double)(c.gI - c.totalInf[pn]);
                c.migPopRate[pn] = w;
            }
            tm = seq_sum(w, min(LANES, P - base), tm);
        }
        c.totalMig = tm;
        WSYNC();
    }
}

// UpdateAllRates (pyx:279-351).  suscepCumulTransition, the migration diagonal, actualSizes and
// maxEffectiveBirth depend on parameters only and are computed on the host (vgx_api.hip), in the same order.
static __device__ void update_all_rates(Ctx &c) {
    const VgxDevParams &p = *c.p;
    const int P = c.P, S = c.S, lane = c.lane;
    double tr = 0.0;
    for (int pn = 0; pn < P; ++pn) {
        birth_update(c, pn);
        tE_fill(c, pn);
        double inf = row_sum(c, pn);
        double imm = 0.0;
        for (int sn = 0; sn < S; ++sn) {
            double v = p.suscepCumul[sn] * (double)c.sus[pn * S + sn];
            if (lane == 0) c.immSrc[pn * S + sn] = v;
            imm += v;
        }
        double pr = inf + imm;
        if (lane == 0) { c.infectPopRate[pn] = inf; c.immunePopRate[pn] = imm; c.popRate[pn] = pr; }
        tr += pr;
    }
    c.totalRate = tr;
    WSYNC();
    // effectiveMigration[pn1, pn2] (pyx:327-338): lane <-> target pn2, serial over sources and the inner sum
    for (int base = 0; base < P; base += LANES) {
        int pn2 = base + lane;
        double mx = 0.0;
        if (pn2 < P) {
            const double *m2 = p.mig + (int64_t)pn2 * P;
            for (int pn1 = 0; pn1 < P; ++pn1) {
                if (pn1 == pn2) continue;
                const double *m1 = p.mig + (int64_t)pn1 * P;
                double e = 0.0;
                for (int pn3 = 0; pn3 < P; ++pn3) e += m1[pn3] * m2[pn3] * c.cd[pn3] / c.as[pn3];
                c.effMig[(int64_t)pn1 * P + pn2] = e;
                if (e > mx) mx = e;
            }
            c.maxEBM[pn2] = mx * p.maxEffectiveBirth;
        }
    }
    WSYNC();
    double tm = 0.0;
    for (int base = 0; base < P; base += LANES) {
        int pn = base + lane;
        double w = 0.0;
        if (pn < P) {
            w = c.maxEBM[pn] * (double)c.totalSus[pn] * (double)(c.gI - c.totalInf[pn]);
            c.migPopRate[pn] = w;
        }
        tm = seq_sum(w, min(LANES, P - base), tm);
    }
    c.totalMig = tm;
    WSYNC();
}

static __device__ void add_event(Ctx &c, int type, int hap, int pop, int nh, int np) {  // events.pxi:37-44
    if (c.record_events) {
        int64_t slot = c.ev_ptr - c.ev_base;
        if (slot >= 0 && slot < c.evcap) {
            int lane = c.lane;
            if (lane < 5) {
                int v = lane == 0 ? type : lane == 1 ? hap : lane == 2 ? pop : lane == 3 ? nh : np;
                c.ev_cols[slot * 5 + lane] = v;
            } else if (lane == 5) {
                c.ev_time[slot] = c.currentTime;
            }
        } else {
            c.error = ERR_CAPACITY;
        }
    }
    c.ev_ptr += 1;
}

static __device__ void check_lockdown(Ctx &c, int pi) {  // pyx:698-710
    const VgxDevParams &p = *c.p;
    for (int pass = 0; pass < 2; ++pass) {
        bool flip;
        if (pass == 0) flip = ((double)c.totalInf[pi] > p.startLD[pi] * (double)p.sizes[pi]) && c.lockON[pi] == 0;
        else flip = ((double)c.totalInf[pi] < p.endLD[pi] * (double)p.sizes[pi]) && c.lockON[pi] == 1;
        if (flip) {
            WSYNC();
            if (c.lane == 0) {
                c.cd[pi] = pass == 0 ? p.cdAfter[pi] : p.cdBefore[pi];
                c.lockON[pi] = pass == 0 ? 1 : 0;
            }
            c.swapLD += 1;
            WSYNC();
            update_all_rates(c);
            if (c.loc_n < VGX_LOC_CAP) {
                if (c.lane == 0) {
                    c.loc_rec[c.loc_n * 2 + 0] = pass == 0 ? 1 : 0;
                    c.loc_rec[c.loc_n * 2 + 1] = pi;
                    c.loc_time[c.loc_n] = c.currentTime;
                }
            } else {
                c.error = ERR_CAPACITY;
            }
            c.loc_n += 1;
        }
    }
}

static __device__ __forceinline__ int mutate(const Ctx &c, int hi, int s, int DS) {  // pyx:2420-2427
    int digit4 = 1 << (2 * (c.sites - s - 1));
    int AS = (hi / digit4) % 4;
    if (DS >= AS) DS += 1;
    return hi + (DS - AS) * digit4;
}

// NewInfections / NewRecoveries on the population counters (pyx:246-260); the list is updated by the caller
static __device__ __forceinline__ void counters_infect(Ctx &c, int pi, int si, int64_t num) {
    if (c.lane == 0) {
        c.sus[pi * c.S + si] -= num;
        c.totalSus[pi] -= num;
        c.totalInf[pi] += num;
    }
    c.gI += num;
    WSYNC();
}

// summary trajectories: emit the state for every grid point passed by the time step (state before the event)
static __device__ void traj_emit(Ctx &c, double t_new, bool final_fill) {
    while (c.traj_next < c.traj_points) {
        double tg = c.traj_t0 + (double)c.traj_next * c.traj_dt;
        if (!final_fill && !(tg < t_new)) break;
        double *o = c.traj + c.traj_next * (int64_t)c.P * 2;
        for (int base = 0; base < c.P; base += LANES) {
            int pn = base + c.lane;
            if (pn < c.P) {
                o[pn * 2 + 0] = (double)c.totalInf[pn];
                o[pn * 2 + 1] = (double)c.totalSus[pn];
            }
        }
        c.traj_next += 1;
    }
}

// ------------------------------------------------------------------------------------------------
static __device__ void immunity_transition(Ctx &c, int pi) {  // pyx:550-564
    const VgxDevParams &p = *c.p;
    const int S = c.S;
    int ssi = choose_serial(c, [&](int i) { return c.immSrc[pi * S + i]; }, S, c.immunePopRate[pi], c.rn);
    int tsi = choose_serial(c, [&](int i) { return p.suscepTransition[ssi * S + i]; }, S, p.suscepCumul[ssi], c.rn);
    WSYNC();
    if (c.lane == 0) {
        c.sus[pi * S + ssi] -= 1;
        c.sus[pi * S + tsi] += 1;
        c.immSrc[pi * S + ssi] = (double)c.sus[pi * S + ssi] * p.suscepCumul[ssi];
        c.immSrc[pi * S + tsi] = (double)c.sus[pi * S + tsi] * p.suscepCumul[tsi];
    }
    WSYNC();
    update_rates(c, pi, false, true, false);
    c.iC += 1;
    add_event(c, EV_SUSCCHANGE, ssi, pi, tsi, 0);
}

static __device__ void birth(Ctx &c, int pi, int k, int hi, int cb) {  // pyx:568-605 (recombination off)
    const VgxDevParams &p = *c.p;
    const int S = c.S;
    const double *x = c.xC + ((int64_t)pi * c.CB + cb) * S;
    double ws = 0.0;
    for (int sn = 0; sn < S; ++sn) ws += x[sn];
    int si = choose_serial(c, [&](int i) { return x[i]; }, S, ws, c.rn);
    counters_infect(c, pi, si, 1);
    if (c.lane == 0) LN(c, pi)[k] += 1;
    add_event(c, EV_BIRTH, hi, pi, si, c.H);
    if (c.lane == 0) c.immSrc[pi * S + si] = p.suscepCumul[si] * (double)c.sus[pi * S + si];
    WSYNC();
    update_rates(c, pi, true, true, true);
    c.bC += 1;
}

static __device__ void death(Ctx &c, int pi, int k, int hi, bool sampling) {  // pyx:616-635
    const VgxDevParams &p = *c.p;
    const int S = c.S;
    int st = (int)p.suscType[hi];
    if (c.lane == 0) {
        c.sus[pi * S + st] += 1;
        c.totalSus[pi] += 1;
        c.totalInf[pi] -= 1;
    }
    c.gI -= 1;
    int64_t left = LN(c, pi)[k] - 1;
    if (left == 0) {
        list_remove_at(c, pi, k);
    } else {
        if (c.lane == 0) LN(c, pi)[k] = left;
    }
    WSYNC();
    if (c.lane == 0) c.immSrc[pi * S + st] = (double)c.sus[pi * S + st] * p.suscepCumul[st];
    WSYNC();
    update_rates(c, pi, true, true, true);
    if (sampling) {
        c.sC += 1;
        add_event(c, EV_SAMPLING, hi, pi, st, 0);
    } else {
        c.dC += 1;
        add_event(c, EV_DEATH, hi, pi, st, 0);
    }
}

static __device__ void mutation(Ctx &c, int pi, int k, int hi, int cls) {  // pyx:640-667
    const VgxDevParams &p = *c.p;
    const int sites = c.sites;
    const double *mr = p.mRate + (int64_t)hi * sites;
    int mi = choose_serial(c, [&](int i) { return mr[i]; }, sites, p.c_tm[cls], c.rn);
    const double *hm = p.hapMutType + ((int64_t)hi * sites + mi) * 3;
    int DS = choose_serial(c, [&](int i) { return hm[i]; }, 3, hm[0] + hm[1] + hm[2], c.rn);
    int nhi = mutate(c, hi, mi, DS);
    list_add(c, pi, nhi, +1);
    if (c.error) return;
    list_add(c, pi, hi, -1);
    update_rates(c, pi, true, false, false);
    c.mC += 1;
    add_event(c, EV_MUTATION, hi, pi, nhi, 0);
}

static __device__ int generate_migration(Ctx &c) {  // pyx:672-694
    const VgxDevParams &p = *c.p;
    const int P = c.P, S = c.S, lane = c.lane;
    int tpi = choose_lds(c, c.migPopRate, P, c.totalMig, c.rn);
    // fastChoose_skip(totalInfectious, globalInfectious - totalInfectious[tpi], rn, skip=tpi), fast_choose.pxi:36-52
    int spi;
    {
        double r = (double)(c.gI - c.totalInf[tpi]) * c.rn;
        int start = (tpi == 0) ? 1 : 0;
        int64_t carry = 0;
        spi = -1;
        int64_t total = 0;
        for (int base = 0; base < P && spi < 0; base += LANES) {
            int k = base + lane;
            int64_t w = (k < P && k != tpi && k >= start) ? c.totalInf[k] : 0;
            int64_t pre = iscan(w, lane) + carry;
            unsigned long long hit = __ballot(k < P && k != tpi && k >= start && !((double)pre < r));
            if (hit) {
                int j = __ffsll((long long)hit) - 1;
                spi = base + j;
                total = bcast_i64(pre, j);
            }
            carry = bcast_i64(pre, LANES - 1);
        }
        if (spi < 0) { spi = P - 1; total = carry; }  // clamp at n-1 (may equal skip only then)
        if (start >= P) { spi = P - 1; total = 0; }
        int64_t wi = c.totalInf[spi];
        if (wi == 0) c.error = ERR_ZERO_WEIGHT;
        c.rn = (r - (double)(total - wi)) / (double)wi;
    }
    // fastChoose(infectious[spi], totalInfectious[spi], rn): int64 weights over the occupancy list
    int hi;
    {
        const int n = c.nocc[spi];
        const int64_t *ln = LN(c, spi);
        double r = (double)c.totalInf[spi] * c.rn;
        int64_t carry = 0, total = 0, wi = 0;
        int kk = -1;
        for (int base = 0; base < n && kk < 0; base += LANES) {
            int k = base + lane;
            int64_t w = (k < n) ? ln[k] : 0;
            int64_t pre = iscan(w, lane) + carry;
            unsigned long long hit = __ballot(k < n && !((double)pre < r));
            if (hit) {
                int j = __ffsll((long long)hit) - 1;
                kk = base + j;
                total = bcast_i64(pre, j);
                wi = bcast_i64(w, j);
            }
            carry = bcast_i64(pre, LANES - 1);
        }
        if (kk < 0) {
            if (n > 0 && LH(c, spi)[n - 1] == c.H - 1) { kk = n - 1; total = carry; wi = ln[n - 1]; }
            else { c.error = ERR_ZERO_WEIGHT; return tpi; }
        }
        hi = LH(c, spi)[kk];
        c.rn = (r - (double)(total - wi)) / (double)wi;
    }
    int si = choose_serial_i64(c, [&](int i) { return c.sus[tpi * S + i]; }, S, c.totalSus[tpi], c.rn);
    double p_accept = c.effMig[(int64_t)spi * P + tpi] * p.bRate[hi] * p.susc[(int64_t)hi * S + si] / c.maxEBM[tpi];
    if (c.rn < p_accept) {
        counters_infect(c, tpi, si, 1);
        list_add(c, tpi, hi, +1);
        update_rates(c, tpi, true, true, true);
        c.migPlus += 1;
        add_event(c, EV_MIGRATION, hi, spi, si, tpi);
    } else {
        c.migNon += 1;
    }
    return tpi;
}

static __device__ int generate_event(Ctx &c, double u) {  // pyx:483-512
    const VgxDevParams &p = *c.p;
    int pi;
    c.rn = u;
    double choose = c.rn * (c.totalRate + c.totalMig);
    if (c.totalRate > choose) {
        c.rn = choose / c.totalRate;
        pi = choose_lds(c, c.popRate, c.P, c.totalRate, c.rn);
        choose = c.rn * c.popRate[pi];
        if (c.immunePopRate[pi] > choose) {
            c.rn = choose / c.immunePopRate[pi];
            immunity_transition(c, pi);
        } else {
            c.rn = (choose - c.immunePopRate[pi]) / c.infectPopRate[pi];
            tE_fill(c, pi);  // from the cached per-class birth rates: the same values UpdateRates stored
            double w_sel;
            int k = row_select(c, pi, c.infectPopRate[pi], c.rn, w_sel);
            if (c.error) return pi;
            int hi = LH(c, pi)[k];
            int cls = LC(c, pi)[k];
            int cb = p.c_bidx[cls];
            double e0 = c.birthC[(int64_t)pi * c.CB + cb], e1 = p.c_d[cls], e2 = p.c_s[cls] * p.sampMult[pi], e3 = p.c_tm[cls];
            double tEv = c.tE[cls];
            int ei = choose_serial(c, [&](int i) { return i == 0 ? e0 : i == 1 ? e1 : i == 2 ? e2 : e3; }, 4, tEv, c.rn);
            if (ei == 0) birth(c, pi, k, hi, cb);
            else if (ei == 1) death(c, pi, k, hi, false);
            else if (ei == 2) death(c, pi, k, hi, true);
            else mutation(c, pi, k, hi, cls);
        }
    } else {
        c.rn = (choose - c.totalRate) / c.totalMig;
        pi = generate_migration(c);
    }
    return pi;
}

static __device__ void restart(Ctx &c, const VgxDevRep &r) {  // pyx:714-738
    const int P = c.P, S = c.S, lane = c.lane;
    c.ev_ptr = 0;
    c.bC = c.dC = c.sC = c.mC = c.iC = 0;
    c.migPlus = c.migNon = 0;
    c.currentTime = 0.0;
    c.traj_next = 0;
    int64_t g = 0;
    for (int pn = 0; pn < P; ++pn) {
        int64_t ts = 0;
        for (int sn = 0; sn < S; ++sn) {
            int64_t v = r.i_sus[pn * S + sn];
            if (lane == 0) c.sus[pn * S + sn] = v;
            ts += v;
        }
        int n = r.i_nocc[pn];
        int64_t ti = 0;
        for (int base = 0; base < n; base += LANES) {
            int k = base + lane;
            int64_t ct = 0;
            if (k < n) {
                ct = r.i_cnt[(int64_t)pn * r.i_cap + k];
                LH(c, pn)[k] = r.i_hap[(int64_t)pn * r.i_cap + k];
                LC(c, pn)[k] = r.i_cls[(int64_t)pn * r.i_cap + k];
                LN(c, pn)[k] = ct;
            }
            ti += bcast_i64(iscan(ct, lane), LANES - 1);
        }
        if (lane == 0) { c.nocc[pn] = n; c.totalSus[pn] = ts; c.totalInf[pn] = ti; }
        g += ti;
    }
    c.gI = g;
    WSYNC();
    for (int pn = 0; pn < P; ++pn) check_lockdown(c, pn);
    update_all_rates(c);
}

extern "C" __global__ void __launch_bounds__(LANES) vgx_direct_kernel(VgxDirectArgs a) {
    const int rep = blockIdx.x;
    if (rep >= a.n_replicates) return;
    const int lane = threadIdx.x;
    const VgxDevParams &p = a.p;
    const VgxDevRep &r = a.r;
    const int P = p.P, S = p.S;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx c;
    c.P = P; c.S = S; c.H = p.H; c.C = p.C; c.CB = p.CB; c.sites = p.sites; c.lane = lane;
    c.p = &a.p;
    double *ld = (double *)smem;
    c.popRate = ld; ld += P;
    c.infectPopRate = ld; ld += P;
    c.immunePopRate = ld; ld += P;
    c.migPopRate = ld; ld += P;
    c.maxEBM = ld; ld += P;
    c.cd = ld; ld += P;
    c.as = ld; ld += P;
    c.immSrc = ld; ld += P * S;
    c.tE = ld; ld += p.C;
    int64_t *li = (int64_t *)ld;
    c.totalSus = li; li += P;
    c.totalInf = li; li += P;
    c.lockON = li; li += P;
    c.sus = li; li += P * S;

    double *gD = r.popD + (int64_t)rep * PD_COUNT * P;
    int64_t *gI = r.popI + (int64_t)rep * PI_COUNT * P;
    for (int pn = lane; pn < P; pn += LANES) {
        c.popRate[pn] = gD[PD_POPRATE * P + pn];
        c.infectPopRate[pn] = gD[PD_INFECT * P + pn];
        c.immunePopRate[pn] = gD[PD_IMMUNE * P + pn];
        c.migPopRate[pn] = gD[PD_MIG * P + pn];
        c.maxEBM[pn] = gD[PD_MAXEBM * P + pn];
        c.cd[pn] = gD[PD_CD * P + pn];
        c.as[pn] = p.actualSizes[pn];
        c.totalSus[pn] = gI[PI_TOTSUS * P + pn];
        c.totalInf[pn] = gI[PI_TOTINF * P + pn];
        c.lockON[pn] = gI[PI_LOCK * P + pn];
    }
    for (int i = lane; i < P * S; i += LANES) {
        c.sus[i] = r.sus[(int64_t)rep * P * S + i];
        c.immSrc[i] = r.immSrc[(int64_t)rep * P * S + i];
    }
    c.birthC = r.birthC + (int64_t)rep * P * p.CB;
    c.xC = r.xC + (int64_t)rep * P * p.CB * S;
    c.effMig = r.effMig + (int64_t)rep * P * P;
    c.nocc = r.nocc + (int64_t)rep * P;
    c.cap = r.cap;
    c.lhap = r.lhap + (int64_t)rep * P * r.cap;
    c.lcls = r.lcls + (int64_t)rep * P * r.cap;
    c.lcnt = r.lcnt + (int64_t)rep * P * r.cap;
    c.evcap = r.evcap; c.ev_base = r.ev_base;
    c.ev_time = r.ev_time + (int64_t)rep * r.evcap;
    c.ev_cols = r.ev_cols + (int64_t)rep * r.evcap * 5;
    c.loc_rec = r.loc_rec + (int64_t)rep * VGX_LOC_CAP * 2;
    c.loc_time = r.loc_time + (int64_t)rep * VGX_LOC_CAP;
    c.traj_points = r.traj_points; c.traj_t0 = r.traj_t0; c.traj_dt = r.traj_dt;
    c.traj = r.traj ? r.traj + (int64_t)rep * r.traj_points * P * 2 : nullptr;
    c.record_events = a.record_events;
    VgxRepScalars *sc = r.sc + rep;
    c.currentTime = sc->currentTime; c.totalRate = 0.0; c.totalMig = 0.0; c.rn = 0.0;
    c.gI = sc->globalInfectious;
    c.bC = sc->bCounter; c.dC = sc->dCounter; c.sC = sc->sCounter; c.mC = sc->mCounter; c.iC = sc->iCounter;
    c.swapLD = sc->swapLockdown; c.migPlus = sc->migPlus; c.migNon = sc->migNonPlus;
    c.ev_ptr = sc->ev_ptr; c.ev_size = a.ev_size; c.loc_n = 0; c.error = 0;
    c.traj_next = 0;
    WSYNC();

    // PrepareParameters tail (pyx:449-451); FirstInfection and the initial snapshot were done by the host
    for (int pn = 0; pn < P; ++pn) check_lockdown(c, pn);
    update_all_rates(c);

    const double tlimit = (double)a.time;
    const bool has_tlimit = !(a.time == -1.0f);
    const int64_t seed = r.seeds[rep];
    int64_t loops = 0, restarts = 0, good_attempt = sc->good_attempt;
    for (int64_t att = 0; att < a.attempts; ++att) {  // pyx:402-418
        VgxPcg64 g;
        vgx_pcg64_seed(g, (uint64_t)seed, (uint32_t)att);
        if (c.totalRate + c.totalMig != 0.0 && c.gI != 0) {
            while (c.ev_ptr < c.ev_size && (a.sample_size == -1 || c.sC <= a.sample_size) &&
                   (!has_tlimit || c.currentTime < tlimit)) {
                if (loops >= a.max_loop) { c.error = ERR_LOOP_GUARD; break; }
                loops += 1;
                double u1 = vgx_pcg64_double(g);
                double t_new = c.currentTime + (-vgx_log(u1) / (c.totalRate + c.totalMig));  // SampleTime pyx:476-478
                if (c.traj) traj_emit(c, t_new, false);
                c.currentTime = t_new;
                double u2 = vgx_pcg64_double(g);
                int pi = generate_event(c, u2);
                if (c.error) break;
                if (c.totalRate == 0.0 || c.gI == 0) break;
                check_lockdown(c, pi);
                if (c.error) break;
            }
        }
        if (c.error) break;
        if (c.ev_ptr <= 100 && a.iterations > 100) {
            restart(c, r);
            restarts += 1;
        } else {
            good_attempt = att + 1;
            break;
        }
    }
    if (c.traj) traj_emit(c, 0.0, true);

    WSYNC();
    for (int pn = lane; pn < P; pn += LANES) {
        gD[PD_POPRATE * P + pn] = c.popRate[pn];
        gD[PD_INFECT * P + pn] = c.infectPopRate[pn];
        gD[PD_IMMUNE * P + pn] = c.immunePopRate[pn];
        gD[PD_MIG * P + pn] = c.migPopRate[pn];
        gD[PD_MAXEBM * P + pn] = c.maxEBM[pn];
        gD[PD_CD * P + pn] = c.cd[pn];
        gI[PI_TOTSUS * P + pn] = c.totalSus[pn];
        gI[PI_TOTINF * P + pn] = c.totalInf[pn];
        gI[PI_LOCK * P + pn] = c.lockON[pn];
    }
    for (int i = lane; i < P * S; i += LANES) {
        r.sus[(int64_t)rep * P * S + i] = c.sus[i];
        r.immSrc[(int64_t)rep * P * S + i] = c.immSrc[i];
    }
    if (lane == 0) {
        sc->currentTime = c.currentTime; sc->totalRate = c.totalRate; sc->totalMig = c.totalMig;
        sc->globalInfectious = c.gI;
        sc->bCounter = c.bC; sc->dCounter = c.dC; sc->sCounter = c.sC; sc->mCounter = c.mC; sc->iCounter = c.iC;
        sc->swapLockdown = c.swapLD; sc->migPlus = c.migPlus; sc->migNonPlus = c.migNon;
        sc->good_attempt = good_attempt;
        sc->ev_ptr = c.ev_ptr; sc->loop_iterations = loops; sc->restarts = restarts;
        sc->loc_n = c.loc_n; sc->error = c.error; sc->traj_next = c.traj_next;
    }
}

// Gives every replicate the same start state (the host model's state at the beginning of the call):
// occupancy lists, susceptible counts, contact densities, population totals and lockdown flags.
extern "C" __global__ void __launch_bounds__(LANES) vgx_init_reps_kernel(
    VgxDevRep r, int P, int S, int64_t R, const int32_t *s_nocc, const int32_t *s_hap, const int32_t *s_cls,
    const int64_t *s_cnt, int64_t s_cap, const int64_t *s_sus, const double *s_cd, const int64_t *s_tot) {
    const int64_t rep = blockIdx.x;
    if (rep >= R) return;
    const int lane = threadIdx.x;
    for (int pn = 0; pn < P; ++pn) {
        int n = s_nocc[pn];
        int32_t *lh = r.lhap + (rep * P + pn) * r.cap, *lc = r.lcls + (rep * P + pn) * r.cap;
        int64_t *ln = r.lcnt + (rep * P + pn) * r.cap;
        for (int k = lane; k < n; k += LANES) {
            lh[k] = s_hap[(int64_t)pn * s_cap + k];
            lc[k] = s_cls[(int64_t)pn * s_cap + k];
            ln[k] = s_cnt[(int64_t)pn * s_cap + k];
        }
    }
    for (int pn = lane; pn < P; pn += LANES) {
        r.nocc[rep * P + pn] = s_nocc[pn];
        r.popD[(rep * PD_COUNT + PD_CD) * P + pn] = s_cd[pn];
        r.popI[(rep * PI_COUNT + PI_TOTSUS) * P + pn] = s_tot[pn];
        r.popI[(rep * PI_COUNT + PI_TOTINF) * P + pn] = s_tot[P + pn];
        r.popI[(rep * PI_COUNT + PI_LOCK) * P + pn] = s_tot[2 * P + pn];
    }
    for (int i = lane; i < P * S; i += LANES) r.sus[rep * P * S + i] = s_sus[i];
}
