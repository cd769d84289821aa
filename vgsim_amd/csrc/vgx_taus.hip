// vgx_taus.hip — Poisson tau-leaping of SMALL models with the step loop ON THE DEVICE (gfx950).
//
// Same path as vgx_tau.hip (SimulatePopulation_tau pyx:2293-2346: Propensities pyx:2351-2417, ChooseTau pyx:2432-2450,
// GenerateEvents_tau pyx:2454-2529, UpdateCompartmentCounts_tau pyx:2536-2593, CheckLockdown pyx:698-710, Restart pyx:714-738).
// The step kernels of vgx_tau.hip are built for states of 10^5 .. 10^9 compartments: about 15 launches and one host
// synchronisation per try, 0.15-0.2 ms per step whatever the model — on a 16-haplotype model that is twenty times slower than
// one CPU core.  Here ONE workgroup per replicate runs the whole call: compartments and the deltas of a try live in LDS, a step
// is propensities -> tau -> {draw every channel, bounds check, halve tau} -> apply -> lockdowns with block barriers between the
// phases, and the host reads the log of the accepted steps when the kernel is through.
//
// Work mapping.  Drift: wavefront w of the block takes populations w, w + 8, ...; its lanes take the haplotypes of the population.
// Draws: one work item per CHANNEL (recovery, sampling, 3 x sites mutations, S transmissions, (P-1) S migrations out of every
// compartment and the immunity transitions; the reference's channels, pyx:2464-2520), each with its own Poisson draw from its own
// Philox stream, keyed by (seed, attempt) with the counter (channel, step, try): reproducible whatever the launch geometry.  The net drifts that
// ChooseTau reads are evaluated per compartment by gathering (incoming mutation and migration), the susceptible compartments'
// by a fixed-order reduction over the population's wavefront: tau does not depend on any scheduling order.  Deltas are integers
// (LDS atomics commute).  The bounds check books migrants on their SOURCE compartment as upstream does (pyx:2473 vs pyx:2548).
// Distributional parity with the oracle: tests/test_hip_tau.py.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/vgx.h"
#include "vgx_dev.h"
#include "vgx_taus.h"
#include "vgx_tau_rng.h"

enum { TY_BIRTH = 0, TY_DEATH = 1, TY_SAMPLING = 2, TY_MUTATION = 3, TY_SUSCCHANGE = 4, TY_MIGRATION = 5 };

// In-kernel stamps of the diagnostic build (-DVGX_PROFILE; tools/profile_taus.py): shader cycles per phase of the step loop, thread 0 of
// replicate 0: [0] loop condition + contact densities, [1] drift + ChooseTau, [2] a try's zeroing, [3] draws, [4] bounds check,
// [5] apply + totals + record, [6] steps, [7] tries; inside the draws (wavefront 0): [8] rates, [9] sampler, [10] bookkeeping of a round.
#ifdef VGX_PROFILE
__device__ unsigned long long vgx_taus_prof[12];
extern "C" int vgx_taus_get_profile(unsigned long long *out, int clear) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vgx_taus_prof), sizeof(unsigned long long) * 12) != hipSuccess) return 1;
    if (clear) {
        unsigned long long z[12] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vgx_taus_prof), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#define TSPROF(i) do { const long long t_ = clock64(); pacc[i] += t_ - pt; pt = t_; } while (0)
#else
#define TSPROF(i)
#endif
// TT = threads of the workgroup.  One trajectory (or a few) wants many lanes per step (512: two rounds of the 528 channels of the 16 x 3
// model); ensembles that fill the chip want many small workgroups per CU instead — measured (tools/probe_taus_threads.py, steps/s of all
// replicates): 16 x 3, 2048 replicates: 6.2e7 at 64 threads, 3.6e7 at 256, 2.2e7 at 512; 512 replicates: 1.8e7 / 3.5e7 / 2.2e7; 64:
// 2.3e6 / 5.0e6 / 5.5e6; the 64 x 4 and 256 x 5 models switch at the same ensemble sizes.
// (the arguments are read through the kernarg segment pointer, as in vgx_solo.hip / vgx_quadg.hip: by-value parameters referenced all over
// the loop were held in scalar registers for the whole kernel, 330 of them spilled)
typedef const VgxTausArgs __attribute__((address_space(4))) *TausKA;
template <int TT>
static __device__ __forceinline__ void taus_body() {
    const auto &a = *(TausKA)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &p = a.p;
    const int rep = blockIdx.x;
    const int P = p.P, H = p.H, S = p.S, sites = p.sites, PH = P * H;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = TT / 64;

    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    int32_t *I = (int32_t *)tsm, *dChk = I + PH, *dApp = dChk + PH;
    int64_t *Sv = (int64_t *)(dApp + PH + (PH & 1)), *dS = Sv + P * S, *tot = dS + P * S;
    double *dSd = (double *)(tot + P), *cd = dSd + P * S, *F = cd + P, *mdg = F + P, *mult = mdg + P, *eff = mult + P;
    double *thS = eff + P * P, *thE = thS + P;                    // lockdown thresholds startLD * sizes, endLD * sizes (pyx:699, 705)
    int64_t *szs = (int64_t *)(thE + P);
    // parameter tables: class of a haplotype, per rate class (recovery, sampling x, total mutation, birth class, group), per birth
    // class (transmission rate, susceptibility row)
    const int C = p.C, CB = p.CB;
    double *l_cd = (double *)(szs + P), *l_cs = l_cd + C, *l_ctm = l_cs + C, *l_cbb = l_ctm + C, *l_sig = l_cbb + CB;
    int32_t *lock = (int32_t *)(l_sig + CB * S), *l_bidx = lock + P, *l_stype = l_bidx + C, *l_cls = l_stype + C;
    __shared__ unsigned long long s_tau;
    __shared__ int s_fail, s_dirty, s_flips;
    __shared__ unsigned int s_nrows;
    __shared__ long long s_cnt[8];        // tallies of the try: births, recoveries, samples, mutations, immunity, migrations
    __shared__ double l_mutp[48];         // a.mutp (kernel argument) where the lanes index it

    int32_t *gIr = a.I + (int64_t)rep * PH;
    for (int i = tid; i < PH; i += TT) I[i] = gIr[i];
    for (int i = tid; i < P * S; i += TT) Sv[i] = a.S[(int64_t)rep * P * S + i];
    for (int i = tid; i < P; i += TT) {
        cd[i] = a.cd[(int64_t)rep * P + i];
        lock[i] = a.lock[(int64_t)rep * P + i];
        mdg[i] = p.mig[(int64_t)i * P + i];
        mult[i] = p.sampMult[i];
        tot[i] = a.totInf[(int64_t)rep * P + i];
        thS[i] = p.startLD[i] * (double)p.sizes[i];
        thE[i] = p.endLD[i] * (double)p.sizes[i];
        szs[i] = p.sizes[i];
    }
    for (int i = tid; i < C; i += TT) { l_cd[i] = p.c_d[i]; l_cs[i] = p.c_s[i]; l_ctm[i] = p.c_tm[i]; l_bidx[i] = p.c_bidx[i]; l_stype[i] = p.c_stype[i]; }
    for (int i = tid; i < CB; i += TT) l_cbb[i] = p.cb_b[i];
    for (int i = tid; i < CB * S; i += TT) l_sig[i] = p.cb_sigma[i];
    for (int i = tid; i < H; i += TT) l_cls[i] = p.cls[i];
    if (tid < 48) l_mutp[tid] = a.mutp[tid / 3][tid % 3];
    if (tid == 0) { s_dirty = 1; s_flips = 0; s_fail = 0; s_tau = (unsigned long long)__double_as_longlong(1.0); }
    __syncthreads();

    // ---- call state (thread-uniform copies in every thread) ----
    double tnow = a.t0, tau_l = 0.0;
    long long gI = a.gI0, ev_ptr = a.ev_ptr0, good = a.good0, restarts = 0, steps = 0, mev_base = 0, tries_total = 0;
    long long cnt[8];
    for (int i = 0; i < 8; ++i) cnt[i] = a.base_cnt[i];
    int att = 0, err = 0;
    uint32_t step = 0;
    long long ev_ptr_start = a.ev_ptr0;
    const bool has_tl = !(a.time == -1.0f);
    const uint64_t seed = (uint64_t)a.seeds[rep];
    int64_t *slog = a.slog + (int64_t)rep * a.slog_cap * 3;
    int64_t *mev = a.mev + (int64_t)rep * (a.mev_cap > 0 ? a.mev_cap : 0) * 6;
    bool running = a.attempts > 0 && a.start_ok;
    bool fresh = true, finished = a.attempts <= 0;
    int64_t guard = 0;

    // CheckLockdown for every population (pyx:698-710), by one thread in population order; `t` = the time the records carry
    auto check_lockdowns = [&](double t) {
        if (tid == 0) {
            for (int pn = 0; pn < P; ++pn) {
                for (int pass = 0; pass < 2; ++pass) {
                    const double ti = (double)tot[pn];
                    const bool flip = pass == 0 ? (ti > thS[pn] && lock[pn] == 0) : (ti < thE[pn] && lock[pn] == 1);
                    if (!flip) continue;
                    cd[pn] = pass == 0 ? p.cdAfter[pn] : p.cdBefore[pn];
                    lock[pn] = pass == 0 ? 1 : 0;
                    const unsigned long long slot = a.loc_n[rep];
                    if (slot < VGX_LOC_CAP) {
                        a.loc_rec[((int64_t)rep * VGX_LOC_CAP + slot) * 2 + 0] = pass == 0 ? 1 : 0;
                        a.loc_rec[((int64_t)rep * VGX_LOC_CAP + slot) * 2 + 1] = pn;
                        a.loc_time[(int64_t)rep * VGX_LOC_CAP + slot] = t;
                    } else {
                        s_fail = 7;
                    }
                    a.loc_n[rep] = slot + 1;
                    s_flips += 1;
                    s_dirty = 1;
                }
            }
        }
    };

#ifdef VGX_PROFILE
    long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt = clock64();
#endif
    while (!finished) {
        // ---- loop condition (pyx:2312) / end of an attempt (pyx:2331-2335) ----
        const bool go = running && ev_ptr < a.ev_size && (a.sample_size == -1 || cnt[2] < a.sample_size) &&
                        (!has_tl || tnow < (double)a.time) && (fresh || gI != 0);
        fresh = false;
        if (!go) {
            running = false;
            if (ev_ptr <= 100 && a.iterations > 100) {
                // Restart (pyx:714-738): compartments back to the initial snapshot, counters to zero (swapLockdown stays), then
                // CheckLockdown for every population at time 0; the lockdown records of the failed attempt stay
                restarts += 1;
                __syncthreads();
                for (int i = tid; i < PH; i += TT) I[i] = a.i_I[i];
                for (int i = tid; i < P * S; i += TT) Sv[i] = a.i_S[i];
                if (tid == 0) { s_flips = 0; s_fail = 0; s_tau = (unsigned long long)__double_as_longlong(1.0); }
                __syncthreads();
                for (int pn = wv; pn < P; pn += NW) {
                    long long t = 0;
                    for (int hn = lane; hn < H; hn += 64) t += I[pn * H + hn];
                    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
                    if (lane == 0) tot[pn] = t;
                }
                __syncthreads();
                check_lockdowns(0.0);
                __syncthreads();
                for (int i = 0; i < 6; ++i) cnt[i] = 0;
                cnt[6] += s_flips;
                cnt[7] = 0;
                if (s_fail == 7) err = 7;
                long long g0 = 0;
                for (int pn = 0; pn < P; ++pn) g0 += tot[pn];
                gI = g0;
                tnow = 0.0; ev_ptr = 0; ev_ptr_start = 0; mev_base = 0; steps = 0;
                att += 1;
                if (att < a.attempts && err == 0) { running = g0 != 0 && a.rates_nonzero_initial; fresh = true; }
                else finished = true;
            } else {
                good = att + 1;
                finished = true;
            }
            continue;
        }
        if (++guard > (int64_t)4 * (a.iterations + 16) * (a.attempts > 1 ? a.attempts : 1)) { err = 6; break; }

        // ---- what depends on the contact densities (pyx:327-338 and the transmission factor of pyx:2410-2414) ----
        if (s_dirty) {
            __syncthreads();
            for (int pn = tid; pn < P; pn += TT) {
                double f = 0.0;
                for (int q = 0; q < P; ++q) { const double m = p.mig[(int64_t)pn * P + q]; f += m * m * cd[q] / p.actualSizes[q]; }
                F[pn] = f;
            }
            for (int i = tid; i < P * P; i += TT) {
                const int t = i / P, s2 = i - t * P;
                double e = 0.0;
                if (t != s2)
                    for (int q = 0; q < P; ++q) e += p.mig[(int64_t)t * P + q] * p.mig[(int64_t)s2 * P + q] * cd[q] / p.actualSizes[q];
                eff[i] = e;      // effectiveMigration[t][s2]
            }
            __syncthreads();
            if (tid == 0) s_dirty = 0;
        }

        TSPROF(0);
        // ---- Propensities + ChooseTau: net drift of every compartment, tau candidates ----
        double cand = 1.0;
        for (int pn = wv; pn < P; pn += NW) {
            double ds_part[VGX_TAUS_MAX_S];
#pragma unroll
            for (int sn = 0; sn < VGX_TAUS_MAX_S; ++sn) ds_part[sn] = 0.0;
            const double Fp = F[pn];
            for (int hn = lane; hn < H; hn += 64) {
                const int c = pn * H + hn;
                const int Ic = I[c];
                const double Ih = (double)Ic;
                const int cl = l_cls[hn], cb = l_bidx[cl], st = l_stype[cl];
                const double dec = l_cd[cl] + l_cs[cl] * mult[pn];
                double drift = -(dec + l_ctm[cl]) * Ih;
#pragma unroll
                for (int sn = 0; sn < VGX_TAUS_MAX_S; ++sn) ds_part[sn] += sn == st ? dec * Ih : 0.0;
                // incoming mutation: the 3 x sites single-site neighbours
                for (int s = 0; s < sites; ++s) {
                    const int sh = 2 * (sites - 1 - s), AS = (hn >> sh) & 3;
                    for (int x = 1; x < 4; ++x) {
                        const int nb = hn ^ (x << sh);
                        const int In = I[pn * H + nb];
                        if (In == 0) continue;
                        const int al = AS ^ x, i = AS - (AS > al ? 1 : 0);
                        double r;
                        if (a.mut_uniform) r = l_mutp[s * 3 + i];
                        else {
                            const double *hm = p.hapMutType + ((int64_t)nb * sites + s) * 3;
                            r = p.mRate[(int64_t)nb * sites + s] * hm[i] / (hm[0] + hm[1] + hm[2]);
                        }
                        drift += r * (double)In;
                    }
                }
                const double b = l_cbb[cb];
#pragma unroll
                for (int sn = 0; sn < VGX_TAUS_MAX_S; ++sn) {
                    if (sn < S) {
                        const double bs = b * l_sig[cb * S + sn] * (double)Sv[pn * S + sn];
                        double v = bs * Fp * Ih;                                             // transmission (pyx:2410-2414)
                        for (int q = 0; q < P; ++q)                                          // migration into pn (pyx:2366-2367)
                            if (q != pn) v += eff[pn * P + q] * bs * (double)I[q * H + hn] * mdg[q];
                        drift += v;
                        ds_part[sn] -= v;
                    }
                }
                const double ad = fabs(drift);
                if (ad >= 1e-8) cand = fmin(cand, fmax((double)(0.03f * (float)Ic) / 2.0, 1.0) / ad);
            }
            // the population's susceptible compartments: wavefront reduction in a fixed order, then the immunity transitions
#pragma unroll
            for (int sn = 0; sn < VGX_TAUS_MAX_S; ++sn) {
                if (sn >= S) break;
                double v = ds_part[sn];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
                if (lane == 0) {
                    for (int s2 = 0; s2 < S; ++s2)
                        if (s2 != sn) v += p.suscepTransition[s2 * S + sn] * (double)Sv[pn * S + s2] - p.suscepTransition[sn * S + s2] * (double)Sv[pn * S + sn];
                    dSd[pn * S + sn] = v;
                    const double ad = fabs(v);
                    if (ad >= 1e-8) cand = fmin(cand, fmax((double)(0.03f * (float)Sv[pn * S + sn]) / 2.0, 1.0) / ad);
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) cand = fmin(cand, __shfl_down(cand, o));
        if (lane == 0) atomicMin(&s_tau, (unsigned long long)__double_as_longlong(cand));   // (positive doubles order like their bits)
        // the deltas of the first try (nothing above reads them; the previous step is through with its counters)
        for (int i = tid; i < PH; i += TT) { dChk[i] = 0; dApp[i] = 0; }
        for (int i = tid; i < P * S; i += TT) dS[i] = 0;
        if (tid < 8) s_cnt[tid] = 0;
        if (tid == 0) { s_fail = 0; s_nrows = 0; }
        __syncthreads();
        tau_l = __longlong_as_double((long long)s_tau);

        TSPROF(1);
        // ---- GenerateEvents_tau until a try passes the bounds check (pyx:2316-2321) ----
        uint32_t retry = 0;
        while (true) {
            if (retry > 0) {     // (the first try's deltas were cleared before the barrier that closed ChooseTau)
                __syncthreads();
                for (int i = tid; i < PH; i += TT) { dChk[i] = 0; dApp[i] = 0; }
                for (int i = tid; i < P * S; i += TT) dS[i] = 0;
                if (tid < 8) s_cnt[tid] = 0;
                if (tid == 0) { s_fail = 0; s_nrows = 0; }
                __syncthreads();
            }
            TSPROF(2);
            long long tl[6] = {0, 0, 0, 0, 0, 0};
            auto row = [&](long long num, int type, int hap, int pop, int nh, int np) {
                if (a.mev_cap <= 0) return;
                const long long slot = mev_base + (long long)atomicAdd(&s_nrows, 1u);
                if (slot < a.mev_cap) {
                    int64_t *r6 = mev + slot * 6;
                    r6[0] = num; r6[1] = type; r6[2] = hap; r6[3] = pop; r6[4] = nh; r6[5] = np;
                }
            };
            // one work item per CHANNEL (pyx:2464-2520): channel ch of compartment c = (pn, hn) — 0 recovery, 1 sampling,
            // 2 .. 2 + 3 sites mutations (site, derived state), then S transmissions, then (P - 1) S migrations (target population
            // in order without pn, group) — with its own stream, keyed by the channel's index; then the P S S immunity transitions
            const int NCH = 2 + 3 * sites + S + (P - 1) * S;
            const int64_t nwork = (int64_t)PH * NCH + (int64_t)P * S * S;
            for (int64_t base = 0; base < nwork; base += TT) {      // (all lanes of a wavefront stay together: the sums below)
                const int64_t w = base + tid;
                // what the item adds: to its own compartment as the check books it / as it is applied (summed over the wavefront's
                // lanes of the same compartment before they touch LDS: a compartment's channels sit in neighbouring lanes, and
                // atomics of many lanes on one address are worked off one after the other), to a susceptible compartment
                int own_c = -2 - (int)(w >= nwork), own_chk = 0, own_app = 0, ds_idx = -1;
                long long ds_val = 0;
                // the channel's rate (lanes of a wavefront hold channels of different kinds): ONE draw after the branches — a sampler
                // inlined into every branch would be run through once per kind by every wavefront
                double lam = 0.0;
                int kind = -1, pn = 0, hn = 0, x0 = 0, x1 = 0;     // kind: 0 recovery, 1 sampling, 2 mutation, 3 transmission, 4 migration, 5 immunity
                if (w < nwork && w >= (int64_t)PH * NCH) {      // immunity transition (pyx:2477-2486): x0 = source, x1 = target group
                    const int i = (int)(w - (int64_t)PH * NCH);
                    pn = i / (S * S);
                    const int j = i - pn * S * S;
                    x0 = j / S; x1 = j - x0 * S;
                    own_c = -1;
                    if (x0 != x1) { kind = 5; lam = p.suscepTransition[x0 * S + x1] * (double)Sv[pn * S + x0] * tau_l; }
                } else if (w < nwork) {
                    // items in CHANNEL-major order, the kinds with the large rates first (recovery, sampling, transmissions, then the
                    // mutations, then the migrations): the lanes of a wavefront draw from rates of one scale (one branch of the sampler),
                    // and what spills over the TT threads into a second round are the cheapest draws
                    const int chm = (int)(w / PH), c = (int)(w - (int64_t)chm * PH);
                    const int ch = chm < 2 ? chm : chm < 2 + S ? 2 + 3 * sites + (chm - 2) : chm < 2 + S + 3 * sites ? 2 + (chm - 2 - S) : chm;
                    own_c = c;
                    const int Ic = I[c];
                    if (Ic != 0) {                 // (propensity 0: no draw, as random_poisson)
                        pn = c / H; hn = c - pn * H;
                        const double Ih = (double)Ic;
                        const int cl = l_cls[hn], cb = l_bidx[cl];
                        if (ch < 2) {   // recovery, sampling (pyx:2483-2495): x0 = the class's group
                            kind = ch; x0 = l_stype[cl];
                            lam = (ch == 0 ? l_cd[cl] : l_cs[cl] * mult[pn]) * Ih * tau_l;
                        } else if (ch < 2 + 3 * sites) {   // mutation (pyx:2497-2504): x0 = the mutant's haplotype
                            const int s2 = (ch - 2) / 3, i = (ch - 2) - 3 * s2;
                            double r;
                            if (a.mut_uniform) r = l_mutp[s2 * 3 + i];
                            else {
                                const double *hm = p.hapMutType + ((int64_t)hn * sites + s2) * 3;
                                r = p.mRate[(int64_t)hn * sites + s2] * hm[i] / (hm[0] + hm[1] + hm[2]);
                            }
                            kind = 2; x0 = tau_mutate(sites, hn, s2, i);
                            lam = r * Ih * tau_l;
                        } else if (ch < 2 + 3 * sites + S) {   // transmission (pyx:2506-2512): x0 = group
                            kind = 3; x0 = ch - 2 - 3 * sites;
                            lam = l_cbb[cb] * l_sig[cb * S + x0] * (double)Sv[pn * S + x0] * F[pn] * Ih * tau_l;
                        } else {   // migration out of (pn, hn) (pyx:2464-2474): x0 = group, x1 = target population
                            const int j = ch - 2 - 3 * sites - S, qq = j / S;
                            kind = 4; x0 = j - qq * S; x1 = qq + (qq >= pn ? 1 : 0);
                            lam = eff[x1 * P + pn] * mdg[pn] * l_cbb[cb] * Ih * tau_l * l_sig[cb * S + x0] * (double)Sv[x1 * S + x0];
                        }
                    }
                }
                TSPROF(8);
                long long k = 0;
                if (kind >= 0 && lam > 0.0) {
                    TauRng g;
                    g.init(seed, (uint32_t)att, (uint64_t)w, step, retry);
                    k = tau_poisson(g, lam);
                }
                TSPROF(9);
                if (k) {
                    if (kind <= 1) {
                        if (kind == 0) tl[1] += k; else tl[2] += k;
                        row(k, kind == 0 ? TY_DEATH : TY_SAMPLING, hn, pn, x0, 0);
                        ds_idx = pn * S + x0; ds_val = k;
                        own_chk = -(int)k; own_app = -(int)k;
                    } else if (kind == 2) {
                        atomicAdd(&dChk[pn * H + x0], (int)k);
                        atomicAdd(&dApp[pn * H + x0], (int)k);
                        own_chk = -(int)k; own_app = -(int)k;
                        tl[3] += k;
                        row(k, TY_MUTATION, hn, pn, x0, 0);
                    } else if (kind == 3) {
                        own_chk = (int)k; own_app = (int)k;
                        ds_idx = pn * S + x0; ds_val = -k;
                        tl[0] += k;
                        row(k, TY_BIRTH, hn, pn, x0, 0);
                    } else if (kind == 4) {      // booked on the SOURCE by the check, applied to the target
                        own_chk = (int)k;
                        atomicAdd(&dApp[x1 * H + hn], (int)k);
                        ds_idx = x1 * S + x0; ds_val = -k;
                        tl[5] += k;
                        row(k, TY_MIGRATION, hn, pn, x0, x1);
                    } else {
                        atomicAdd((unsigned long long *)&dS[pn * S + x0], (unsigned long long)(-k));
                        ds_idx = pn * S + x1; ds_val = k;
                        tl[4] += k;
                        row(k, TY_SUSCCHANGE, x0, pn, x1, 0);
                    }
                }
                // the compartments' own changes (neighbouring lanes hold different compartments: no two lanes of a wavefront meet on
                // an address here unless the model has fewer compartments than a wavefront has lanes)
                if (own_c >= 0) {
                    if (own_chk) atomicAdd(&dChk[own_c], own_chk);
                    if (own_app) atomicAdd(&dApp[own_c], own_app);
                }
                // the susceptible compartments: one sum per distinct compartment among the wavefront's lanes
                {
                    unsigned long long todo = __ballot(ds_idx >= 0);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const int key = __shfl(ds_idx, leader);
                        const bool mine = ds_idx == key;
                        long long v = mine ? ds_val : 0;
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
                        v = __shfl(v, 0);
                        if (lane == leader && v) atomicAdd((unsigned long long *)&dS[key], (unsigned long long)v);
                        todo &= ~__ballot(mine);
                    }
                }
                TSPROF(10);
            }
            for (int i = 0; i < 6; ++i) {
                long long v = tl[i];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
                if (lane == 0 && v) atomicAdd((unsigned long long *)&s_cnt[i], (unsigned long long)v);
            }
            __syncthreads();
            TSPROF(3);
            // bounds check (pyx:2522-2528)
            int bad = 0;
            for (int i = tid; i < PH; i += TT) {
                const long long v = (long long)I[i] + dChk[i];
                if (v < 0 || v > szs[i / H]) bad = 1;
            }
            for (int i = tid; i < P * S; i += TT) {
                const long long v = Sv[i] + dS[i];
                if (v < 0 || v > szs[i / S]) bad = 1;
            }
            if (bad) s_fail = 1;
            __syncthreads();
            tries_total += 1;
            TSPROF(4);
#ifdef VGX_PROFILE
            pacc[7] += 1;
#endif
            if (a.mev_cap > 0 && mev_base + (long long)s_nrows > a.mev_cap) { err = VGX_ERR_CAPACITY; break; }
            if (!s_fail) break;
            tau_l *= 0.5;
            retry += 1;
            if (retry > 200) { err = 5; break; }
        }
        if (err) break;

        // ---- UpdateCompartmentCounts_tau (pyx:2536-2593), the MULTITYPE record (pyx:2325) ----
        for (int i = tid; i < PH; i += TT) I[i] += dApp[i];
        for (int i = tid; i < P * S; i += TT) Sv[i] += dS[i];
        __syncthreads();
        for (int pn = wv; pn < P; pn += NW) {
            long long t = 0;
            for (int hn = lane; hn < H; hn += 64) t += I[pn * H + hn];
            for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
            if (lane == 0) tot[pn] = t;
        }
        tnow += tau_l;
        long long drawn = 0;
        for (int i = 0; i < 6; ++i) { cnt[i] += s_cnt[i]; drawn += s_cnt[i]; }
        cnt[7] += drawn;
        const long long nrows = (long long)s_nrows;
        if (tid == 0) {
            const long long slot = ev_ptr - ev_ptr_start;
            if (slot >= 0 && slot < a.slog_cap) {
                slog[slot * 3 + 0] = __double_as_longlong(tnow);
                slog[slot * 3 + 1] = mev_base | ((long long)retry << 56);   // (+ the step's rejected tries, <= 200, in the top byte: vgx_get_tau_tries)
                slog[slot * 3 + 2] = mev_base + nrows;
            }
        }
        mev_base += nrows;
        ev_ptr += 1; steps += 1; step += 1;
        TSPROF(5);
#ifdef VGX_PROFILE
        pacc[6] += 1;
#endif
        __syncthreads();
        long long g = 0;
        for (int pn = 0; pn < P; ++pn) g += tot[pn];
        gI = g;
        if (tid == 0) { s_flips = 0; s_tau = (unsigned long long)__double_as_longlong(1.0); }     // (s_tau: the next step's ChooseTau starts from 1)
        __syncthreads();
        if (gI != 0) {          // pyx:2326-2329: no lockdown check after extinction
            check_lockdowns(tnow);
            __syncthreads();
            cnt[6] += s_flips;
            if (s_fail == 7) { err = 7; break; }
        }
    }

#ifdef VGX_PROFILE
    if (tid == 0 && rep == 0) for (int i = 0; i < 12; ++i) atomicAdd(&vgx_taus_prof[i], (unsigned long long)pacc[i]);
#endif
    // ---- state and results back ----
    __syncthreads();
    for (int i = tid; i < PH; i += TT) gIr[i] = I[i];
    for (int i = tid; i < P * S; i += TT) a.S[(int64_t)rep * P * S + i] = Sv[i];
    for (int i = tid; i < P; i += TT) {
        a.cd[(int64_t)rep * P + i] = cd[i];
        a.lock[(int64_t)rep * P + i] = lock[i];
        a.totInf[(int64_t)rep * P + i] = tot[i];
    }
    if (tid == 0) {
        int64_t *o = a.res + (int64_t)rep * 24;
        o[TS_TAU] = __double_as_longlong(tau_l);
        o[TS_GI] = gI;
        for (int i = 0; i < 8; ++i) o[TS_CNT0 + i] = cnt[i];
        o[TS_EVPTR] = ev_ptr; o[TS_ATT] = att; o[TS_GOOD] = good; o[TS_RESTARTS] = restarts; o[TS_STEPS] = steps;
        o[TS_MEVROWS] = mev_base; o[TS_ERROR] = err; o[TS_TIME] = __double_as_longlong(tnow); o[TS_EVPTR0] = ev_ptr_start;
        o[TS_TRIES] = tries_total;
    }
}

extern "C" __global__ void __launch_bounds__(64) vgx_taus_kernel_t64(VgxTausArgs) { taus_body<64>(); }
extern "C" __global__ void __launch_bounds__(256) vgx_taus_kernel_t256(VgxTausArgs) { taus_body<256>(); }
extern "C" __global__ void __launch_bounds__(512) vgx_taus_kernel(VgxTausArgs) { taus_body<512>(); }

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_taus(const VgxTausArgs *a, hipStream_t s) {
    const size_t lds = vgx_taus_lds_bytes(a->p.P, a->p.H, a->p.S, a->p.C, a->p.CB);
    // the workgroup size: see taus_body (VGX_TAUS_THREADS = 64 / 256 / 512 forces one, for comparisons)
    // (tools/probe_taus_threads.py, 16 x 3 / 64 x 4 / 256 x 5 models at 64 ... 4096 replicates: 512 threads lead below two workgroups per CU,
    // 256 from two to eight, 64 from eight on — whatever the model's size)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    // workgroups a CU will hold at once: what the launch offers, and what fits its LDS (a model of 8192 compartments fills it alone)
    const int64_t fit = std::max<int64_t>(1, (int64_t)(160 * 1024) / (int64_t)std::max<size_t>(lds, 1));
    const int64_t per_cu = std::min<int64_t>(std::max<int64_t>(a->R / cus, 1), fit);
    int tt = per_cu >= 8 ? 64 : per_cu >= 2 ? 256 : 512;
    if (const char *ft = getenv("VGX_TAUS_THREADS")) { const int v = atoi(ft); if (v == 64 || v == 256 || v == 512) tt = v; }
    void (*k)(VgxTausArgs) = tt == 64 ? vgx_taus_kernel_t64 : tt == 256 ? vgx_taus_kernel_t256 : vgx_taus_kernel;
    hipError_t err = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k, dim3((unsigned)a->R), dim3((unsigned)tt), lds, s, *a);
    return hipGetLastError();
}
