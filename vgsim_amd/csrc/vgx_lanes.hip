// vgx_lanes.hip — direct Gillespie for SMALL models: one replicate per LANE (64 replicates per wavefront).
//
// The wave-per-replicate kernel (vgx_direct.hip) spends a whole 64-lane wavefront on the parallelism inside one
// event; when hapNum x popNum is small there is none to exploit and its chains run mostly empty (BASELINE config 2:
// one haplotype, one population).  Here every lane carries its own trajectory and runs the reference's loops
// serially in the reference's order — BirthDeathModel.SimulatePopulation (src/_BirthDeath.pyx:396-429), SampleTime
// pyx:476, GenerateEvent pyx:483, UpdateRates pyx:516, UpdateAllRates pyx:279, BirthRate pyx:382, ImmunityTransition
// pyx:550, Birth pyx:568, Death/Sampling pyx:616/630, Mutation pyx:640, GenerateMigration pyx:672, CheckLockdown
// pyx:698, Restart pyx:714, fastChoose / fastChoose_skip fast_choose.pxi:18/36, Events.AddEvent events.pxi:37 —
// so every sum, comparison and recycled random number is the reference's by construction (-ffp-contract=off).
//
// State: dense per-replicate arrays in HBM, interleaved across replicates (element i of replicate r at [i * R + r]),
// so that the lanes of a wave touch consecutive addresses whenever they walk an array in step (the loops over
// haplotypes, populations and groups).  Start and end state are exchanged with the rest of libvgx in the layout of
// the wave kernel (ordered occupancy lists, population blocks), so the host side is the same for both kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"

namespace {

enum { ERR_ZERO_WEIGHT = 3, ERR_CAPACITY = 4, ERR_LOOP_GUARD = 5 };
enum { EV_BIRTH = 0, EV_DEATH, EV_SAMPLING, EV_MUTATION, EV_SUSCCHANGE, EV_MIGRATION };

struct Lane {
    const VgxDevParams *p;
    int H, P, S, sites;
    int64_t R, rep;
    // per-replicate arrays (stride R)
    int64_t *inf, *sus, *totS, *totI, *lock;
    double *cd, *birth, *tE, *hpr, *shpr, *immSrc, *infP, *immP, *popR, *migR, *maxEBM, *effMig;
    // scalars
    double currentTime, totalRate, totalMig, rn;
    int64_t gI, ev_ptr, loc_n;
    int64_t cB, cD, cS, cM, cI, cSwap, cMigP, cMigN;
    int error;
    // logs
    double *ev_rate; int32_t *ev_cols; int64_t evcap, ev_base; int record_events;
    int32_t *loc_rec; double *loc_time; int64_t *loc_iter; int64_t loc_cap;
    double den; int64_t iter_key;   // this iteration's totalRate + totalMigrationRate; (attempt << 40) | loop iteration
    int64_t *rec; int64_t rec_cap, rec_n;
    double *traj; int64_t traj_points, traj_next; double traj_t0, traj_dt;
};

#define AT(a, i) (a)[(int64_t)(i) * L.R + L.rep]

__device__ __forceinline__ double d_rate(const Lane &L, int h) { return L.p->c_d[L.p->cls[h]]; }
__device__ __forceinline__ double s_rate(const Lane &L, int h) { return L.p->c_s[L.p->cls[h]]; }
__device__ __forceinline__ double tm_rate(const Lane &L, int h) { return L.p->c_tm[L.p->cls[h]]; }

__device__ __forceinline__ void add_event(Lane &L, int type, int hap, int pop, int nh, int np) {  // events.pxi:37-44
    if (L.record_events) {
        int64_t slot = L.ev_ptr - L.ev_base;
        if (slot >= 0 && slot < L.evcap) {
            int32_t *c = L.ev_cols + slot * VGX_EV_COLS;
            c[0] = type; c[1] = hap; c[2] = pop; c[3] = nh; c[4] = np; c[5] = (int32_t)(uint32_t)L.iter_key;
            L.ev_rate[slot] = L.den;
        } else {
            L.error = ERR_CAPACITY;
        }
    }
    L.ev_ptr += 1;
}

__device__ __forceinline__ double birth_rate(Lane &L, int pi, int hi) {  // pyx:382-392
    const VgxDevParams &p = *L.p;
    double ps = 0.0;
    for (int sn = 0; sn < L.S; ++sn) {
        double x = (double)AT(L.sus, pi * L.S + sn) * p.susc[(int64_t)hi * L.S + sn];
        AT(L.shpr, (pi * L.H + hi) * L.S + sn) = x;
        for (int pn = 0; pn < L.P; ++pn) {
            double m = p.mig[pi * L.P + pn];
            ps += x * m * m * AT(L.cd, pn) / p.actualSizes[pn];
        }
    }
    return p.bRate[hi] * ps;
}

// UpdateAllRates (pyx:279-351); the parameter-only parts (suscepCumulTransition, migration diagonal, actualSizes,
// maxEffectiveBirth) come from the host, computed in the reference's order (vgx_api.hip)
__device__ void update_all_rates(Lane &L) {
    const VgxDevParams &p = *L.p;
    const int P = L.P, H = L.H, S = L.S;
    L.totalRate = 0.0;
    for (int pn = 0; pn < P; ++pn) {
        double infect = 0.0, immune = 0.0;
        for (int hn = 0; hn < H; ++hn) {
            double e0 = birth_rate(L, pn, hn);
            AT(L.birth, pn * H + hn) = e0;
            double t = 0.0;
            t += e0; t += d_rate(L, hn); t += s_rate(L, hn) * p.sampMult[pn]; t += tm_rate(L, hn);   // pyx:313-316
            AT(L.tE, pn * H + hn) = t;
            double w = t * (double)AT(L.inf, pn * H + hn);
            AT(L.hpr, pn * H + hn) = w;
            infect += w;
        }
        for (int sn = 0; sn < S; ++sn) {
            double v = p.suscepCumul[sn] * (double)AT(L.sus, pn * S + sn);
            AT(L.immSrc, pn * S + sn) = v;
            immune += v;
        }
        AT(L.infP, pn) = infect;
        AT(L.immP, pn) = immune;
        double pr = infect + immune;
        AT(L.popR, pn) = pr;
        L.totalRate += pr;
    }
    for (int pn = 0; pn < P; ++pn) AT(L.maxEBM, pn) = 0.0;   // holds maxEffectiveMigration until scaled below
    for (int pn1 = 0; pn1 < P; ++pn1)
        for (int pn2 = 0; pn2 < P; ++pn2) {
            if (pn1 == pn2) continue;
            double e = 0.0;
            for (int pn3 = 0; pn3 < P; ++pn3) e += p.mig[pn1 * P + pn3] * p.mig[pn2 * P + pn3] * AT(L.cd, pn3) / p.actualSizes[pn3];
            AT(L.effMig, pn1 * P + pn2) = e;
            if (e > AT(L.maxEBM, pn2)) AT(L.maxEBM, pn2) = e;
        }
    L.totalMig = 0.0;
    for (int pn = 0; pn < P; ++pn) {
        double mx = AT(L.maxEBM, pn) * p.maxEffectiveBirth;
        AT(L.maxEBM, pn) = mx;
        double w = mx * (double)AT(L.totS, pn) * (double)(L.gI - AT(L.totI, pn));
        AT(L.migR, pn) = w;
        L.totalMig += w;
    }
}

__device__ void update_rates(Lane &L, int pi, bool infect, bool immune, bool migration) {  // pyx:516-546
    const VgxDevParams &p = *L.p;
    const int P = L.P, H = L.H, S = L.S;
    if (infect) {
        double acc = 0.0;
        for (int hn = 0; hn < H; ++hn) {
            double e0 = birth_rate(L, pi, hn);
            AT(L.birth, pi * H + hn) = e0;
            double t = (e0 + d_rate(L, hn) + s_rate(L, hn) * p.sampMult[pi] + tm_rate(L, hn));   // pyx:522-525
            AT(L.tE, pi * H + hn) = t;
            double w = t * (double)AT(L.inf, pi * H + hn);
            AT(L.hpr, pi * H + hn) = w;
            acc += w;
        }
        AT(L.infP, pi) = acc;
    }
    if (immune) {
        double acc = 0.0;
        for (int sn = 0; sn < S; ++sn) acc += AT(L.immSrc, pi * S + sn);
        AT(L.immP, pi) = acc;
    }
    if (infect || immune) {
        AT(L.popR, pi) = AT(L.infP, pi) + AT(L.immP, pi);
        double t = 0.0;
        for (int pn = 0; pn < P; ++pn) t += AT(L.popR, pn);
        L.totalRate = t;
    }
    if (migration) {
        double t = 0.0;
        for (int pn = 0; pn < P; ++pn) {
            double w = AT(L.maxEBM, pn) * (double)AT(L.totS, pn) * (double)(L.gI - AT(L.totI, pn));
            AT(L.migR, pn) = w;
            t += w;
        }
        L.totalMig = t;
    }
}

// fastChoose (fast_choose.pxi:18-31) over a per-replicate f64 array starting at element `off`
__device__ __forceinline__ int choose_lane_f64(Lane &L, const double *a, int off, int n, double tw) {
    double r = tw * L.rn;
    int i = 0;
    double total = AT(a, off);
    while (total < r && i < n - 1) { i += 1; total += AT(a, off + i); }
    double wi = AT(a, off + i);
    if (wi == 0.0) L.error = ERR_ZERO_WEIGHT;
    L.rn = (r - (total - wi)) / wi;
    return i;
}
// ... over a shared (parameter) array
__device__ __forceinline__ int choose_shared_f64(Lane &L, const double *w, int n, double tw) {
    double r = tw * L.rn;
    int i = 0;
    double total = w[0];
    while (total < r && i < n - 1) { i += 1; total += w[i]; }
    double wi = w[i];
    if (wi == 0.0) L.error = ERR_ZERO_WEIGHT;
    L.rn = (r - (total - wi)) / wi;
    return i;
}
__device__ __forceinline__ int choose_lane_i64(Lane &L, const int64_t *a, int off, int n, int64_t tw) {
    double r = (double)tw * L.rn;
    int i = 0;
    int64_t total = AT(a, off);
    while ((double)total < r && i < n - 1) { i += 1; total += AT(a, off + i); }
    int64_t wi = AT(a, off + i);
    if (wi == 0) L.error = ERR_ZERO_WEIGHT;
    L.rn = (r - (double)(total - wi)) / (double)wi;
    return i;
}
__device__ __forceinline__ int choose_skip_lane_i64(Lane &L, const int64_t *a, int n, int64_t tw, int skip) {  // fast_choose.pxi:36-52
    double r = (double)tw * L.rn;
    int i = 0;
    if (skip == 0) i += 1;
    int64_t total = AT(a, i);
    while ((double)total < r && i < n - 1) {
        i += 1;
        if (i != skip) total += AT(a, i);
    }
    int64_t wi = AT(a, i);
    if (wi == 0) L.error = ERR_ZERO_WEIGHT;
    L.rn = (r - (double)(total - wi)) / (double)wi;
    return i;
}

__device__ __forceinline__ void new_infection(Lane &L, int pi, int si, int hi) {  // pyx:246-251
    AT(L.sus, pi * L.S + si) -= 1;
    AT(L.totS, pi) -= 1;
    AT(L.inf, pi * L.H + hi) += 1;
    AT(L.totI, pi) += 1;
    L.gI += 1;
}
__device__ __forceinline__ void new_recovery(Lane &L, int pi, int si, int hi) {  // pyx:255-260
    AT(L.sus, pi * L.S + si) += 1;
    AT(L.totS, pi) += 1;
    AT(L.inf, pi * L.H + hi) -= 1;
    AT(L.totI, pi) -= 1;
    L.gI -= 1;
}

__device__ __forceinline__ int mutate(const Lane &L, int hi, int s, int DS) {  // pyx:2420-2427
    int digit4 = 1 << (2 * (L.sites - s - 1));
    int AS = (hi / digit4) % 4;
    if (DS >= AS) DS += 1;
    return hi + (DS - AS) * digit4;
}

// Recombination branch of Birth (pyx:575-596).  birthInf[hn] = eventHapPopRate[pi, hn, 0] * infectious[pi, hn] with one
// host of `hi` set aside (pyx:578-582) is not stored: the same products are formed again for the scan, in the same order.
__device__ __forceinline__ void recombinant_birth(Lane &L, int pi, int hi, int si) {
    const VgxDevParams &p = *L.p;
    const int H = L.H;
    L.rn = L.rn / p.recombination;
    double hs = 0.0;
    for (int hn = 0; hn < H; ++hn)
        hs += AT(L.birth, pi * H + hn) * (double)(AT(L.inf, pi * H + hn) - (hn == hi ? 1 : 0));
    // fastChoose(birthInf, hs, rn), fast_choose.pxi:18-31
    double r = hs * L.rn;
    int i = 0;
    double wi = AT(L.birth, pi * H) * (double)(AT(L.inf, pi * H) - (hi == 0 ? 1 : 0));
    double total = wi;
    while (total < r && i < H - 1) {
        i += 1;
        wi = AT(L.birth, pi * H + i) * (double)(AT(L.inf, pi * H + i) - (i == hi ? 1 : 0));
        total += wi;
    }
    if (wi == 0.0) { L.error = ERR_ZERO_WEIGHT; return; }
    L.rn = (r - (total - wi)) / wi;
    const int hi2 = i;
    const int64_t posRecomb = (int64_t)((double)p.genome_length * L.rn);
    // pyx:586-591 as written: `4**k * floor(h / 4**k) % 4` is (4^k * floor(h / 4^k)) % 4, which is 0 for every site but
    // the last (k = 0), where it is h % 4 -- the recombinant carries only the last site of one parent
    int nhi = 0;
    if (L.sites > 0) nhi = (p.sitesPosition[L.sites - 1] < posRecomb ? hi : hi2) % 4;
    if (L.rec) {
        if (L.rec_n < L.rec_cap) {
            int64_t *o = L.rec + L.rec_n * 5;
            o[0] = L.ev_ptr; o[1] = hi; o[2] = hi2; o[3] = nhi; o[4] = posRecomb;
        } else {
            L.error = ERR_CAPACITY;
        }
    }
    L.rec_n += 1;
    new_infection(L, pi, si, nhi);
    add_event(L, EV_BIRTH, hi, pi, si, hi2);
}

// GenerateEvent (pyx:483-512); returns the population whose lockdown state has to be checked.  RECOMB: the build with the
// recombination branch of Birth (the default kernels do not carry it: same code as before, same registers).
template <bool RECOMB>
__device__ int generate_event(Lane &L, double u) {
    const VgxDevParams &p = *L.p;
    const int P = L.P, H = L.H, S = L.S;
    int pi;
    L.rn = u;
    double choose = L.rn * (L.totalRate + L.totalMig);
    if (L.totalRate > choose) {
        L.rn = choose / L.totalRate;
        pi = choose_lane_f64(L, L.popR, 0, P, L.totalRate);
        choose = L.rn * AT(L.popR, pi);
        if (AT(L.immP, pi) > choose) {
            // ImmunityTransition (pyx:550-564)
            L.rn = choose / AT(L.immP, pi);
            int ssi = choose_lane_f64(L, L.immSrc, pi * S, S, AT(L.immP, pi));
            int tsi = choose_shared_f64(L, p.suscepTransition + ssi * S, S, p.suscepCumul[ssi]);
            AT(L.sus, pi * S + ssi) -= 1;
            AT(L.sus, pi * S + tsi) += 1;
            AT(L.immSrc, pi * S + ssi) = (double)AT(L.sus, pi * S + ssi) * p.suscepCumul[ssi];
            AT(L.immSrc, pi * S + tsi) = (double)AT(L.sus, pi * S + tsi) * p.suscepCumul[tsi];
            update_rates(L, pi, false, true, false);
            L.cI += 1;
            add_event(L, EV_SUSCCHANGE, ssi, pi, tsi, 0);
        } else {
            L.rn = (choose - AT(L.immP, pi)) / AT(L.infP, pi);
            int hi = choose_lane_f64(L, L.hpr, pi * H, H, AT(L.infP, pi));
            // fastChoose over eventHapPopRate[pi, hi, 0..3] (pyx:500)
            double e0 = AT(L.birth, pi * H + hi), e1 = d_rate(L, hi), e2 = s_rate(L, hi) * p.sampMult[pi], e3 = tm_rate(L, hi);
            int ei;
            {
                double r = AT(L.tE, pi * H + hi) * L.rn;
                int i = 0;
                double total = e0;
                while (total < r && i < 3) { i += 1; total += (i == 1 ? e1 : i == 2 ? e2 : e3); }
                double wi = i == 0 ? e0 : i == 1 ? e1 : i == 2 ? e2 : e3;
                if (wi == 0.0) L.error = ERR_ZERO_WEIGHT;
                L.rn = (r - (total - wi)) / wi;
                ei = i;
            }
            if (ei == 0) {
                // Birth (pyx:568-605)
                double ws = 0.0;
                for (int sn = 0; sn < S; ++sn) ws += AT(L.shpr, (pi * H + hi) * S + sn);
                int si = choose_lane_f64(L, L.shpr, (pi * H + hi) * S, S, ws);
                if (RECOMB && L.rn < p.recombination && AT(L.totI, pi) > 1) {
                    recombinant_birth(L, pi, hi, si);
                } else {
                    new_infection(L, pi, si, hi);
                    add_event(L, EV_BIRTH, hi, pi, si, H);
                }
                AT(L.immSrc, pi * S + si) = p.suscepCumul[si] * (double)AT(L.sus, pi * S + si);
                update_rates(L, pi, true, true, true);
                L.cB += 1;
            } else if (ei == 1 || ei == 2) {
                // Death / Sampling (pyx:616-635)
                int st = (int)p.suscType[hi];
                new_recovery(L, pi, st, hi);
                AT(L.immSrc, pi * S + st) = (double)AT(L.sus, pi * S + st) * p.suscepCumul[st];
                update_rates(L, pi, true, true, true);
                if (ei == 1) { L.cD += 1; add_event(L, EV_DEATH, hi, pi, st, 0); }
                else { L.cS += 1; add_event(L, EV_SAMPLING, hi, pi, st, 0); }
            } else {
                // Mutation (pyx:640-667)
                int mi = choose_shared_f64(L, p.mRate + (int64_t)hi * L.sites, L.sites, tm_rate(L, hi));
                const double *hm = p.hapMutType + ((int64_t)hi * L.sites + mi) * 3;
                int DS = choose_shared_f64(L, hm, 3, hm[0] + hm[1] + hm[2]);
                int nhi = mutate(L, hi, mi, DS);
                AT(L.inf, pi * H + nhi) += 1;
                AT(L.inf, pi * H + hi) -= 1;
                update_rates(L, pi, true, false, false);
                L.cM += 1;
                add_event(L, EV_MUTATION, hi, pi, nhi, 0);
            }
        }
    } else {
        // GenerateMigration (pyx:672-694)
        L.rn = (choose - L.totalRate) / L.totalMig;
        int tpi = choose_lane_f64(L, L.migR, 0, P, L.totalMig);
        pi = tpi;
        int spi = choose_skip_lane_i64(L, L.totI, P, L.gI - AT(L.totI, tpi), tpi);
        int hi = choose_lane_i64(L, L.inf, spi * H, H, AT(L.totI, spi));
        int si = choose_lane_i64(L, L.sus, tpi * S, S, AT(L.totS, tpi));
        if (L.error) return pi;
        double p_accept = AT(L.effMig, spi * P + tpi) * p.bRate[hi] * p.susc[(int64_t)hi * S + si] / AT(L.maxEBM, tpi);
        if (L.rn < p_accept) {
            new_infection(L, tpi, si, hi);
            update_rates(L, tpi, true, true, true);
            L.cMigP += 1;
            add_event(L, EV_MIGRATION, hi, spi, si, tpi);
        } else {
            L.cMigN += 1;
        }
    }
    return pi;
}

// CheckLockdown (pyx:698-710) for populations [lo, hi): applies and logs the switches; returns whether any happened
// (UpdateAllRates is a pure function of the state: the caller runs it once after the last switch)
__device__ bool check_lockdowns(Lane &L, int lo, int hi) {
    const VgxDevParams &p = *L.p;
    bool any = false;
    for (int pi = lo; pi < hi; ++pi)
        for (int pass = 0; pass < 2; ++pass) {
            bool flip = pass == 0 ? ((double)AT(L.totI, pi) > p.startLD[pi] * (double)p.sizes[pi] && AT(L.lock, pi) == 0)
                                  : ((double)AT(L.totI, pi) < p.endLD[pi] * (double)p.sizes[pi] && AT(L.lock, pi) == 1);
            if (!flip) continue;
            AT(L.cd, pi) = pass == 0 ? p.cdAfter[pi] : p.cdBefore[pi];
            AT(L.lock, pi) = pass == 0 ? 1 : 0;
            if (L.loc_n < L.loc_cap) {
                L.loc_rec[L.loc_n * 2 + 0] = pass == 0 ? 1 : 0;
                L.loc_rec[L.loc_n * 2 + 1] = pi;
                L.loc_time[L.loc_n] = L.currentTime;
                L.loc_iter[L.loc_n] = L.iter_key;
            } else {
                L.error = ERR_CAPACITY;
            }
            L.cSwap += 1;
            L.loc_n += 1;
            any = true;
        }
    return any;
}

__device__ void traj_emit(Lane &L, double t_new, bool final_fill) {
    while (L.traj_next < L.traj_points) {
        double tg = L.traj_t0 + (double)L.traj_next * L.traj_dt;
        if (!final_fill && !(tg < t_new)) break;
        double *o = L.traj + L.traj_next * (int64_t)L.P * 2;
        for (int pn = 0; pn < L.P; ++pn) {
            o[pn * 2 + 0] = (double)AT(L.totI, pn);
            o[pn * 2 + 1] = (double)AT(L.totS, pn);
        }
        L.traj_next += 1;
    }
}

}  // namespace

// Body shared by the two kernels below: `L` already knows where this lane's dense state lives.
template <bool RECOMB>
static __device__ __forceinline__ void lanes_body(const VgxDirectArgs &a, Lane &L, const int64_t rep) {
    const VgxDevParams &p = a.p;
    const VgxDevRep &r = a.r;
    L.p = &a.p;
    L.H = p.H; L.P = p.P; L.S = p.S; L.sites = p.sites;
    const int H = L.H, P = L.P, S = L.S;

    // ---- start state from the wave kernel's layout ----
    double *gD = r.popD + rep * PD_COUNT * P;
    int64_t *gI = r.popI + rep * PI_COUNT * P;
    int32_t *gN = r.nocc + rep * P;
    for (int i = 0; i < P * H; ++i) AT(L.inf, i) = 0;
    for (int pn = 0; pn < P; ++pn) {
        const int n = gN[pn];
        const int32_t *lh = r.lhap + (rep * P + pn) * r.cap;
        const int64_t *ln = r.lcnt + (rep * P + pn) * r.cap;
        for (int k = 0; k < n; ++k) AT(L.inf, pn * H + lh[k]) = ln[k];
        AT(L.cd, pn) = gD[PD_CD * P + pn];
        AT(L.totS, pn) = gI[PI_TOTSUS * P + pn];
        AT(L.totI, pn) = gI[PI_TOTINF * P + pn];
        AT(L.lock, pn) = gI[PI_LOCK * P + pn];
        for (int sn = 0; sn < S; ++sn) AT(L.sus, pn * S + sn) = r.sus[rep * P * S + pn * S + sn];
    }
    VgxRepScalars *sc = r.sc + rep;
    L.currentTime = sc->currentTime; L.totalRate = 0.0; L.totalMig = 0.0; L.rn = 0.0;
    L.gI = sc->globalInfectious;
    L.cB = sc->bCounter; L.cD = sc->dCounter; L.cS = sc->sCounter; L.cM = sc->mCounter; L.cI = sc->iCounter;
    L.cSwap = sc->swapLockdown; L.cMigP = sc->migPlus; L.cMigN = sc->migNonPlus;
    L.ev_ptr = sc->ev_ptr; L.loc_n = 0; L.error = 0;
    L.evcap = r.evcap; L.ev_base = r.ev_base; L.record_events = a.record_events;
    L.ev_rate = r.ev_rate + rep * r.evcap;
    L.ev_cols = r.ev_cols + rep * r.evcap * VGX_EV_COLS;
    L.loc_cap = r.loc_cap;
    L.loc_rec = r.loc_rec + rep * r.loc_cap * 2;
    L.loc_time = r.loc_time + rep * r.loc_cap;
    L.loc_iter = r.loc_iter + rep * r.loc_cap;
    L.den = 0.0; L.iter_key = 0;
    L.rec = (RECOMB && r.rec) ? r.rec + rep * r.rec_cap * 5 : nullptr; L.rec_cap = r.rec_cap; L.rec_n = 0;
    L.traj_points = r.traj_points; L.traj_t0 = r.traj_t0; L.traj_dt = r.traj_dt; L.traj_next = 0;
    L.traj = r.traj ? r.traj + rep * r.traj_points * P * 2 : nullptr;

    const double tlimit = (double)a.time;
    const bool has_tlimit = !(a.time == -1.0f);
    const int64_t seed = r.seeds[rep];
    int64_t loops = 0, restarts = 0, good_attempt = sc->good_attempt, last_att = -1, last_att_loops = 0;
    int64_t att_ev0 = sc->ev_ptr, att_loc0 = 0, fa_n = 0;   // first log index / lockdown record of the current attempt

    // PrepareParameters tail (pyx:449-451): CheckLockdown for every population, UpdateAllRates
    check_lockdowns(L, 0, P);
    update_all_rates(L);

    for (int64_t att = 0; att < a.attempts && !L.error; ++att) {  // pyx:399-418
        VgxPcg64 g;
        vgx_pcg64_seed(g, (uint64_t)seed, (uint32_t)att);
        last_att = att; last_att_loops = 0;
        if (L.totalRate + L.totalMig != 0.0 && L.gI != 0) {
            while (L.ev_ptr < a.ev_size && (a.sample_size == -1 || L.cS <= a.sample_size) && (!has_tlimit || L.currentTime < tlimit)) {
                if (loops >= a.max_loop) { L.error = ERR_LOOP_GUARD; break; }
                loops += 1;
                last_att_loops += 1;
                double u1 = vgx_pcg64_double(g);
                L.den = L.totalRate + L.totalMig;
                L.iter_key = (att << 40) | last_att_loops;
                double t_new = L.currentTime + (-vgx_log(u1) / L.den);   // SampleTime pyx:476-478
                if (L.traj) traj_emit(L, t_new, false);
                L.currentTime = t_new;
                double u2 = vgx_pcg64_double(g);
                int pi = generate_event<RECOMB>(L, u2);
                if (L.error) break;
                if (L.totalRate == 0.0 || L.gI == 0) break;   // pyx:410-411
                if (check_lockdowns(L, pi, pi + 1)) update_all_rates(L);   // pyx:412
                if (L.error) break;
            }
        }
        if (L.error) break;
        if (L.ev_ptr <= 100 && a.iterations > 100) {
            // Restart (pyx:714-738); swapLockdown survives.  Lockdown records of the failed attempt stay in the log: keep
            // the (rate, iteration) pairs the host clock needs for them
            if (L.loc_n > att_loc0 && L.record_events && r.fa_cap > 0) {
                const int64_t n = L.ev_ptr - att_ev0;
                for (int64_t k = 0; k < n; ++k) {
                    const int64_t slot = att_ev0 + k - L.ev_base;
                    if (fa_n + k < r.fa_cap && slot >= 0 && slot < L.evcap) {
                        r.fa_rate[rep * r.fa_cap + fa_n + k] = L.ev_rate[slot];
                        r.fa_key[rep * r.fa_cap + fa_n + k] = (att << 40) | (int64_t)(uint32_t)L.ev_cols[slot * VGX_EV_COLS + 5];
                    }
                }
                fa_n += n;
            }
            att_ev0 = 0;
            L.iter_key = (att + 1) << 40;   // the CheckLockdown below belongs to the next attempt, before its first iteration
            L.ev_ptr = 0;
            L.cB = L.cD = L.cS = L.cM = L.cI = 0; L.cMigP = L.cMigN = 0;
            L.currentTime = 0.0;
            L.traj_next = 0;
            L.gI = 0;
            for (int pn = 0; pn < P; ++pn) {
                int64_t ts = 0, ti = 0;
                for (int sn = 0; sn < S; ++sn) { int64_t v = r.i_sus[pn * S + sn]; AT(L.sus, pn * S + sn) = v; ts += v; }
                for (int hn = 0; hn < H; ++hn) AT(L.inf, pn * H + hn) = 0;
                const int n = r.i_nocc[pn];
                for (int k = 0; k < n; ++k) {
                    int64_t v = r.i_cnt[(int64_t)pn * r.i_cap + k];
                    AT(L.inf, pn * H + r.i_hap[(int64_t)pn * r.i_cap + k]) = v;
                    ti += v;
                }
                AT(L.totS, pn) = ts; AT(L.totI, pn) = ti;
                L.gI += ti;
            }
            restarts += 1;
            att_loc0 = L.loc_n;
            check_lockdowns(L, 0, P);
            update_all_rates(L);
        } else {
            good_attempt = att + 1;
            break;
        }
    }
    if (L.traj) traj_emit(L, 0.0, true);

    // ---- end state back in the wave kernel's layout ----
    for (int pn = 0; pn < P; ++pn) {
        int32_t *lh = r.lhap + (rep * P + pn) * r.cap, *lc = r.lcls + (rep * P + pn) * r.cap;
        int64_t *ln = r.lcnt + (rep * P + pn) * r.cap, *lt = r.ltsum + (rep * P + pn) * r.capT;
        int n = 0;
        int64_t tsum = 0;
        for (int hn = 0; hn < H; ++hn) {
            int64_t v = AT(L.inf, pn * H + hn);
            if (v == 0) continue;
            if (n < r.cap) { lh[n] = hn; lc[n] = p.cls[hn]; ln[n] = v; }
            else L.error = ERR_CAPACITY;
            tsum += v;
            n += 1;
            if ((n & 63) == 0) { lt[(n >> 6) - 1] = tsum; tsum = 0; }
        }
        if (n & 63) lt[n >> 6] = tsum;
        for (int j = (n + 63) / 64; j < r.capT; ++j) lt[j] = 0;
        gN[pn] = n < r.cap ? n : (int)r.cap;
        gD[PD_POPRATE * P + pn] = AT(L.popR, pn);
        gD[PD_INFECT * P + pn] = AT(L.infP, pn);
        gD[PD_IMMUNE * P + pn] = AT(L.immP, pn);
        gD[PD_MIG * P + pn] = AT(L.migR, pn);
        gD[PD_MAXEBM * P + pn] = AT(L.maxEBM, pn);
        gD[PD_CD * P + pn] = AT(L.cd, pn);
        gI[PI_TOTSUS * P + pn] = AT(L.totS, pn);
        gI[PI_TOTINF * P + pn] = AT(L.totI, pn);
        gI[PI_LOCK * P + pn] = AT(L.lock, pn);
        for (int sn = 0; sn < S; ++sn) {
            r.sus[rep * P * S + pn * S + sn] = AT(L.sus, pn * S + sn);
            r.immSrc[rep * P * S + pn * S + sn] = AT(L.immSrc, pn * S + sn);
        }
    }
    sc->currentTime = L.currentTime; sc->totalRate = L.totalRate; sc->totalMig = L.totalMig;
    sc->globalInfectious = L.gI;
    sc->bCounter = L.cB; sc->dCounter = L.cD; sc->sCounter = L.cS; sc->mCounter = L.cM; sc->iCounter = L.cI;
    sc->swapLockdown = L.cSwap; sc->migPlus = L.cMigP; sc->migNonPlus = L.cMigN;
    sc->good_attempt = good_attempt;
    sc->ev_ptr = L.ev_ptr; sc->loop_iterations = loops; sc->restarts = restarts;
    sc->loc_n = L.loc_n; sc->error = L.error; sc->traj_next = L.traj_next;
    sc->last_attempt = last_att; sc->last_attempt_loops = last_att_loops;
    sc->rec_n = L.rec_n;
    sc->fa_n = fa_n;
}

// state in HBM, interleaved across all replicates
template <bool RECOMB>
static __device__ __forceinline__ void lanes_hbm(const VgxDirectArgs &a, const VgxLaneWs &w) {
    const int64_t rep = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (rep >= a.n_replicates) return;
    Lane L;
    L.R = a.n_replicates; L.rep = rep;
    L.inf = w.inf; L.sus = w.sus; L.totS = w.totS; L.totI = w.totI; L.lock = w.lock;
    L.cd = w.cd; L.birth = w.birth; L.tE = w.tE; L.hpr = w.hpr; L.shpr = w.shpr; L.immSrc = w.immSrc;
    L.infP = w.infP; L.immP = w.immP; L.popR = w.popR; L.migR = w.migR; L.maxEBM = w.maxEBM; L.effMig = w.effMig;
    lanes_body<RECOMB>(a, L, rep);
}
extern "C" __global__ void __launch_bounds__(64) vgx_lanes_kernel(VgxDirectArgs a, VgxLaneWs w) { lanes_hbm<false>(a, w); }
extern "C" __global__ void __launch_bounds__(64) vgx_lanes_recomb_kernel(VgxDirectArgs a, VgxLaneWs w) { lanes_hbm<true>(a, w); }

// state in LDS (minimal models: the whole dense state of 64 replicates fits), interleaved across the 64 lanes
template <bool RECOMB>
static __device__ __forceinline__ void lanes_lds(const VgxDirectArgs &a, unsigned char *lsm) {
    const int64_t rep = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (rep >= a.n_replicates) return;   // no barrier in this kernel: lanes are independent
    const int64_t P = a.p.P, H = a.p.H, S = a.p.S, PH = P * H;
    Lane L;
    L.R = 64; L.rep = threadIdx.x;
    int64_t *wi = (int64_t *)lsm;
    L.inf = wi; wi += PH * 64; L.sus = wi; wi += P * S * 64; L.totS = wi; wi += P * 64; L.totI = wi; wi += P * 64; L.lock = wi; wi += P * 64;
    double *wd = (double *)wi;
    L.cd = wd; wd += P * 64; L.birth = wd; wd += PH * 64; L.tE = wd; wd += PH * 64; L.hpr = wd; wd += PH * 64;
    L.shpr = wd; wd += PH * S * 64; L.immSrc = wd; wd += P * S * 64; L.infP = wd; wd += P * 64; L.immP = wd; wd += P * 64;
    L.popR = wd; wd += P * 64; L.migR = wd; wd += P * 64; L.maxEBM = wd; wd += P * 64; L.effMig = wd; wd += P * P * 64;
    lanes_body<RECOMB>(a, L, rep);
}
extern "C" __global__ void __launch_bounds__(64) vgx_lanes_lds_kernel(VgxDirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    lanes_lds<false>(a, lsm);
}
extern "C" __global__ void __launch_bounds__(64) vgx_lanes_lds_recomb_kernel(VgxDirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    lanes_lds<true>(a, lsm);
}

// ---- host-side launcher ----
extern "C" __attribute__((visibility("hidden"))) size_t vgxi_lanes_elems(int64_t P, int64_t H, int64_t S) {
    const int64_t PH = P * H;
    return (size_t)((PH + P * S + 3 * P) + (P + 3 * PH + PH * S + P * S + 5 * P + P * P));   // 8-byte elements per replicate
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_lanes(const VgxDirectArgs *a, const VgxLaneWs *w, hipStream_t stream) {
    const unsigned blocks = (unsigned)((a->n_replicates + 63) / 64);
    const size_t lds = vgxi_lanes_elems(a->p.P, a->p.H, a->p.S) * 64 * 8;
    const bool recomb = a->p.recombination != 0.0;
    if (lds <= 40 * 1024) {   // four or more blocks per CU keep their state in LDS
        const void *fn = recomb ? (const void *)vgx_lanes_lds_recomb_kernel : (const void *)vgx_lanes_lds_kernel;
        hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        if (recomb) hipLaunchKernelGGL(vgx_lanes_lds_recomb_kernel, dim3(blocks), dim3(64), lds, stream, *a);
        else hipLaunchKernelGGL(vgx_lanes_lds_kernel, dim3(blocks), dim3(64), lds, stream, *a);
    } else {
        if (recomb) hipLaunchKernelGGL(vgx_lanes_recomb_kernel, dim3(blocks), dim3(64), 0, stream, *a, *w);
        else hipLaunchKernelGGL(vgx_lanes_kernel, dim3(blocks), dim3(64), 0, stream, *a, *w);
    }
    return hipGetLastError();
}
