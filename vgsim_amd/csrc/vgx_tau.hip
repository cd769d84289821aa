// vgx_tau.hip — Poisson tau-leaping kernels for gfx950 (MI355X).
//
// Replaces the body of BirthDeathModel.SimulatePopulation_tau (reference src/_BirthDeath.pyx:2293-2346):
// Propensities pyx:2351-2417, ChooseTau pyx:2432-2450, GenerateEvents_tau pyx:2454-2529 (+DrawEventsNum
// pyx:2531), UpdateCompartmentCounts_tau pyx:2536-2593.
//
// The reference materialises one propensity and one event count per channel (propNum of them, pyx:2301:
// 3.8e8 at 65 536 x 64, 7.7e10 at 2^20 x 256) and draws them one after the other from one PCG64 stream.
// Here one thread owns one compartment (population pn, haplotype hn) and evaluates its channels in
// registers; nothing per-channel is stored:
//   * every channel out of (pn, hn) has a propensity proportional to infectious[pn, hn], so empty
//     compartments draw nothing (exactly like random_poisson(lam=0), which consumes no random numbers);
//   * the (P-1)*S migration channels out of a compartment are drawn as ONE Poisson with the summed rate and
//     then split multinomially over (target population, susceptibility group): the same joint law as
//     independent Poissons per channel;
//   * draws come from Philox4x32-10 keyed by (seed, attempt) with the counter (compartment, step, retry):
//     every compartment owns an independent, reproducible stream, whatever the launch geometry;
//   * one pass books both the deltas the reference's bounds check inspects and the deltas it applies
//     (they differ for migrants, pyx:2473 vs pyx:2548); the state is only touched by the commit kernel;
//   * net drifts (infectiousAuxTau / susceptibleAuxTau, the input of ChooseTau) are evaluated per
//     compartment by gathering the incoming mutation and migration terms; tau_l is a grid-wide minimum.
// Arithmetic is f64 with the reference's formulas but not its summation order: the tau path is checked
// distributionally (tests/test_hip_tau.py), as BASELINE.json asks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/vgx.h"
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_tau_lf.h"
#include "vgx_tau_rng.h"

// The tau path is validated distributionally (different random streams anyway), so FMA contraction is allowed
// here, unlike in the bit-exact direct kernel (the library is built with -ffp-contract=off).
#pragma clang fp contract(fast)

#define TB 256  // threads per block of the compartment kernels
#ifndef VGX_DRAW_GX
#define VGX_DRAW_GX 32u   // thread blocks of the draw kernel per (population, replicate)
#endif

// the one-byte copy of four counts (255 = "255 or more": vgx_tau_scan_kernel then reads the count proper)
static __device__ __forceinline__ uint32_t tau_pack8(int a, int b, int c, int d) {
    return (uint32_t)min(a, 255) | ((uint32_t)min(b, 255) << 8) | ((uint32_t)min(c, 255) << 16) | ((uint32_t)min(d, 255) << 24);
}

static __device__ __forceinline__ void atomic_min_pos_double(unsigned long long *addr, double v) {
    // for non-negative doubles the bit patterns order like the values
    atomicMin(addr, (unsigned long long)__double_as_longlong(v));
}

// a launch of a round whose state has passed (VgxTauArgs.gate / spec): nothing to do
static __device__ __forceinline__ bool tau_gate_closed(const VgxTauArgs &a, int rep) {
    if (a.gate == 0) return false;
    const int st = a.spec[rep];
    return a.gate == 1 ? st != 0 : a.gate == 2 ? st != 1 : false;
}

// ------------------------------------------------------------------------------------------------
// effectiveMigration (pyx:327-338), its transposed/padded product with the diagonal, and the transmission factor,
// rebuilt only when the contact densities changed (start of the call, lockdown switch).  grid = (P, R): one block
// per row t.
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_eff_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y, t = blockIdx.x;
    const VgxDevParams &p = a.p;
    const int P = p.P, Pp = a.Ppad;
    if (!a.active[rep] || !a.eff_dirty[rep]) return;
    const double *cd = a.cd + (int64_t)rep * P;
    double *eff = a.effMig + (int64_t)rep * P * P;
    double *AeffT = a.Aeff + (int64_t)rep * P * Pp;   // [spn][tpn], rows padded to Pp with zeros
    for (int s = threadIdx.x; s < P; s += TB) {
        double e = 0.0;
        if (t != s && a.has_mig)
            for (int q = 0; q < P; ++q) e += p.mig[(int64_t)t * P + q] * p.mig[(int64_t)s * P + q] * cd[q] / p.actualSizes[q];
        eff[(int64_t)t * P + s] = e;
        AeffT[(int64_t)s * Pp + t] = e * p.mig[(int64_t)s * P + s];
    }
    for (int k = threadIdx.x; k < Pp - P; k += TB) AeffT[(int64_t)t * Pp + P + k] = 0.0;
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int q = 0; q < P; ++q) {
            double m = p.mig[(int64_t)t * P + q];
            f += m * m * cd[q] / p.actualSizes[q];
        }
        a.F[(int64_t)rep * P + t] = f;
    }
}

// Per-step, per-replicate preparation: transmission factor F[pn] = sum_spn m[pn,spn]^2 cd[spn]/as[spn]
// (pyx:2412-2414), effectiveMigration (pyx:327-338) under the current contact densities,
// Aeff[tpn][spn] = effMig[tpn,spn] * m[spn,spn] (pyx:2366-2367), and the out-migration weight of a source
// population per birth class Gout[spn][cb] = sum_{tpn != spn, sn} effMig[tpn,spn] S[tpn,sn] sigma_cb[sn].
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_prep_kernel(VgxTauArgs a) {
    // grid = (P, R): one block per source population spn; the running sums over the P * S channels (target population,
    // susceptibility group) are a blocked scan: every thread sums a run of consecutive channels, the runs' totals are
    // scanned in LDS (fixed order: the result does not depend on the launch)
    const int rep = blockIdx.y, spn = blockIdx.x;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, CB = p.CB;
    if (!a.active[rep]) return;
    const double *eff = a.effMig + (int64_t)rep * P * P;
    const int64_t *Sus = a.S + (int64_t)rep * P * S;
    __shared__ double part[TB];
    const int n = P * S, per = (n + TB - 1) / TB;
    const int i0 = min(n, (int)threadIdx.x * per), i1 = min(n, i0 + per);
    for (int cb = 0; cb < CB; ++cb) {
        // cumulative weights over (tpn, sn) of the out-migration channels of (spn, cb): target lookup by bisection
        double *cdf = a.migcdf + (((int64_t)rep * P + spn) * CB + cb) * (int64_t)P * S;
        double loc = 0.0;
        for (int i = i0; i < i1; ++i) {
            const int t = i / S, sn = i - t * S;
            if (t != spn && a.has_mig) loc += eff[(int64_t)t * P + spn] * (double)Sus[t * S + sn] * p.cb_sigma[cb * S + sn];
        }
        part[threadIdx.x] = loc;
        __syncthreads();
        for (int off = 1; off < TB; off <<= 1) {
            const double v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0.0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        double acc = threadIdx.x > 0 ? part[threadIdx.x - 1] : 0.0;
        for (int i = i0; i < i1; ++i) {
            const int t = i / S, sn = i - t * S;
            if (t != spn && a.has_mig) acc += eff[(int64_t)t * P + spn] * (double)Sus[t * S + sn] * p.cb_sigma[cb * S + sn];
            cdf[i] = acc;
        }
        // out-migration weight of the source population per birth class = the last running sum
        if (threadIdx.x == TB - 1) a.Gout[((int64_t)rep * P + spn) * CB + cb] = part[TB - 1];
        __syncthreads();
    }
    if (spn != 0) return;
    if (threadIdx.x == 0) {
        a.tau_bits[rep] = (unsigned long long)__double_as_longlong(1.0);  // tau_l starts at 1.0 (pyx:2437)
        a.ok[rep] = 1;
        a.eff_dirty[rep] = 0;
    }
    for (int i = threadIdx.x; i < P * S; i += TB) a.dS[(int64_t)rep * P * S + i] = 0.0;
}

// Incoming-migration term of the drift, migIn[tpn][hn] = sum_spn Aeff[tpn][spn] * I[spn][hn] (pyx:2366-2369):
// a [P x P] x [P x H] product.  Block = 4 waves, lane <-> haplotype of a 64-wide tile, persistent over tiles.
// The tile I[:, h0:h0+64] of ALL source populations is staged once in LDS (int32, coalesced 512-B global reads);
// each wave keeps 8 target populations in registers, reads I[spn][lane] once per source population from LDS and
// the 8 coefficients of the transposed, padded matrix as ONE wave-uniform scalar load (the pointers are
// __restrict__ kernel arguments and the only store goes to migIn, so the loads are provably read-only).
#define TH 64
#define TPW 16
#ifndef MIGIN_TB
#define MIGIN_TB 1024   // waves of a block share one LDS tile of infectious counts: more waves per tile hide the scalar loads
#endif
extern "C" __global__ void __launch_bounds__(MIGIN_TB) vgx_tau_migin_kernel(const double *__restrict__ AeffT_all,
                                                                      const int32_t *__restrict__ I_all,
                                                                      double *__restrict__ migIn_all,
                                                                      const int32_t *__restrict__ active, int P, int Pp,
                                                                      int H) {
    const int rep = blockIdx.y;
    if (!active[rep]) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t *__restrict__ I = I_all + (int64_t)rep * P * H;
    const double *__restrict__ AeffT = AeffT_all + (int64_t)rep * P * Pp;
    double *__restrict__ out = migIn_all + (int64_t)rep * P * H;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    int32_t *It = (int32_t *)tsm;  // [P][TH]
    const int ntiles = (H + TH - 1) / TH;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int h0 = tile * TH;
        __syncthreads();
        for (int idx = threadIdx.x; idx < P * TH; idx += MIGIN_TB) {
            int spn = idx >> 6, h = idx & 63;
            It[idx] = (h0 + h < H) ? I[(int64_t)spn * H + h0 + h] : 0;
        }
        __syncthreads();
        for (int tp0 = wave * TPW; tp0 < P; tp0 += (MIGIN_TB / 64) * TPW) {
            double acc[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = 0.0;
#pragma unroll 4
            for (int spn = 0; spn < P; ++spn) {
                double x = (double)It[spn * TH + lane];
                const double *__restrict__ arow = AeffT + (int64_t)spn * Pp + tp0;  // 8 consecutive wave-uniform coefficients
#pragma unroll
                for (int j = 0; j < TPW; ++j) acc[j] += arow[j] * x;
            }
            if (h0 + lane < H) {
#pragma unroll
                for (int j = 0; j < TPW; ++j)
                    if (tp0 + j < P) out[(int64_t)(tp0 + j) * H + h0 + lane] = acc[j];
            }
        }
    }
}


// ---- uniform migration -----------------------------------------------------------------------------------------------------
// With one common off-diagonal migration probability b (diagonal d = 1 - (P-1) b; what set_migration_probability(p) and
// set_total_migration_probability produce) effectiveMigration (pyx:327-338) has the closed form
//     effMig[t][s] = sum_q m[t][q] m[s][q] w[q] = b^2 W + (d b - b^2)(w[t] + w[s]),   w = cd / actualSizes,  W = sum_q w[q],
// so the incoming migration pressure sum_{s != t} effMig[t][s] m[s][s] I[s][h] (pyx:2366-2367) is
//     d { (b^2 W + (d b - b^2) w[t]) (T[h] - I[t][h]) + (d b - b^2) (TW[h] - w[t] I[t][h]) }
// with the two column sums T = sum_s I[s][h], TW = sum_s w[s] I[s][h]: O(P H) work and no [P][H] f64 array, instead of
// the [P x P] x [P x H] product of vgx_tau_migin_kernel.
#define VGX_MIGU_PMAX 1024
struct MigU { double c1, c2, wt; bool weq; };   // weq: one common weight cd / actualSizes for all populations
// block-uniform (all threads of the block call it); s_part: 16 doubles of shared scratch; ends with a barrier
static __device__ __forceinline__ MigU tau_migu_setup(const VgxTauArgs &a, int rep, int pn, double *s_part) {
    const VgxDevParams &p = a.p;
    const int P = p.P;
    double part = 0.0;
    const double w0 = a.cd[(int64_t)rep * P] / p.actualSizes[0];
    int ne = 0;
    for (int q = threadIdx.x; q < P; q += blockDim.x) {
        const double w = a.cd[(int64_t)rep * P + q] / p.actualSizes[q];
        part += w;
        ne |= w != w0;
    }
    ne = __syncthreads_or(ne);
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    double W = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) W += s_part[i];
    MigU u;
    const double b = a.mig_b, d = a.mig_d, g = d * b - b * b;
    u.wt = a.cd[(int64_t)rep * P + pn] / p.actualSizes[pn];
    u.c1 = d * (b * b * W + g * u.wt);
    u.c2 = d * g;
    u.weq = ne == 0;
    __syncthreads();
    return u;
}
static __device__ __forceinline__ double tau_migu(const MigU &u, double T, double TW, double Ih) {
    return u.c1 * (T - Ih) + u.c2 * (TW - u.wt * Ih);
}

// Column sums over the populations.  grid = (ceil(H / (4 TB)), R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_colsum_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, H = p.H;
    __shared__ double s_w[VGX_MIGU_PMAX];
    for (int q = threadIdx.x; q < P; q += TB) s_w[q] = a.cd[(int64_t)rep * P + q] / p.actualSizes[q];
    __syncthreads();
    const int h0 = (blockIdx.x * TB + threadIdx.x) * 4;
    if (h0 >= H) return;
    const int32_t *I = a.I + (int64_t)rep * P * H;
    long long t[4] = {0, 0, 0, 0};
    double tw[4] = {0.0, 0.0, 0.0, 0.0};
    const bool full = h0 + 3 < H && (H & 3) == 0;
    for (int q = 0; q < P; ++q) {
        int x[4] = {0, 0, 0, 0};
        if (full) { const int4 v = *(const int4 *)(I + (int64_t)q * H + h0); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
        else for (int j = 0; j < 4; ++j) if (h0 + j < H) x[j] = I[(int64_t)q * H + h0 + j];
        const double w = s_w[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) { t[j] += x[j]; tw[j] += w * (double)x[j]; }
    }
    for (int j = 0; j < 4; ++j)
        if (h0 + j < H) { a.colT[(int64_t)rep * H + h0 + j] = (double)t[j]; a.colTW[(int64_t)rep * H + h0 + j] = tw[j]; }
}

// Net drift of every infectious compartment and its tau candidate (Propensities + ChooseTau); partial
// sums of the susceptible drift.  Thread <-> compartment, grid = (ceil(H/TB), P, R); every gather of a
// neighbouring haplotype is coalesced across the lanes (neighbours of consecutive haplotypes are consecutive).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_drift_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H, sites = p.sites, C = p.C, CB = p.CB;
    const int lane = threadIdx.x & 63;
    const int32_t *I = a.I + (int64_t)rep * P * H + (int64_t)pn * H;
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    __shared__ double sdS[TB / 64][64];   // S <= 64 susceptibility groups; one row per wavefront (fixed summation order: no atomics)
    __shared__ unsigned long long smin;
    // class tables and this population's per-class factors in LDS when they are small (else global)
    __shared__ double l_cd[256], l_cs[256], l_ctm[256], l_base[16 * 64], l_mutp[48];
    __shared__ int32_t l_bidx[256], l_stype[256], l_site_flat[16];
    const bool useL = C <= 256 && CB <= 16;
    sdS[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
    if (threadIdx.x == 0) smin = (unsigned long long)__double_as_longlong(1.0);
    if (useL) {
        for (int i = threadIdx.x; i < C; i += TB) {
            l_cd[i] = p.c_d[i]; l_cs[i] = p.c_s[i] * p.sampMult[pn]; l_ctm[i] = p.c_tm[i];
            l_bidx[i] = p.c_bidx[i]; l_stype[i] = p.c_stype[i];
        }
        for (int i = threadIdx.x; i < CB * S; i += TB) l_base[i] = p.cb_b[i / S] * p.cb_sigma[i] * (double)Sus[i % S];
    }
    for (int i = threadIdx.x; i < 3 * sites && i < 48; i += TB) l_mutp[i] = a.mutp[i / 3][i % 3];
    for (int i = threadIdx.x; i < sites && i < 16; i += TB)
        l_site_flat[i] = (a.mutp[i][0] == a.mutp[i][1] && a.mutp[i][1] == a.mutp[i][2]) ? 1 : 0;
    __syncthreads();
    const double F = a.F[(int64_t)rep * P + pn];
    __shared__ double s_wu[16];
    MigU mu = {0.0, 0.0, 0.0, true};
    if (a.has_mig && a.mig_uniform) mu = tau_migu_setup(a, rep, pn, s_wu);
    double cand_min = 1.0;
    for (int hn = blockIdx.x * TB + threadIdx.x; hn - (int)threadIdx.x < H; hn += gridDim.x * TB) {
        const bool live = hn < H;
        const int hh = live ? hn : H - 1;
        const int c = p.cls[hh];
        const int cb = useL ? l_bidx[c] : p.c_bidx[c];
        const int st = useL ? l_stype[c] : p.c_stype[c];
        const int64_t Icell = live ? I[hh] : 0;
        if (live) a.I8[(int64_t)rep * P * H + (int64_t)pn * H + hh] = (uint8_t)(Icell < 255 ? Icell : 255);
        const double Ih = (double)Icell;
        double drift = 0.0;
        // recovery and sampling (pyx:2386-2395), outgoing mutation (pyx:2398-2404: sum_i w_i / sum w == 1)
        const double rec = (useL ? l_cd[c] : p.c_d[c]) * Ih;
        const double samp = useL ? l_cs[c] * Ih : p.c_s[c] * Ih * p.sampMult[pn];
        drift -= rec;
        drift -= samp;
        drift -= (useL ? l_ctm[c] : p.c_tm[c]) * Ih;
        // incoming mutation: sources differ from hn in exactly one site
        for (int s = 0; s < sites; ++s) {
            const int sh = 2 * (sites - s - 1);
            const int AS = (hh >> sh) & 3;
            if (a.mut_uniform && l_site_flat[s]) {
                // the three derived states of this site are equally likely for every source: add the three
                // neighbouring counts as integers, convert and scale once
                // the other three alleles of this site are hh with the site's two bits XORed by 1, 2, 3; their sum stays
                // below the population size (< 2^31)
                int nb = 0;
                if (live) nb = I[hh ^ (1 << sh)] + I[hh ^ (2 << sh)] + I[hh ^ (3 << sh)];
                drift += l_mutp[s * 3] * (double)nb;
                continue;
            }
            for (int al = 0; al < 4; ++al) {
                if (al == AS) continue;
                const int src = hh + ((al - AS) << sh);
                const int64_t Is = live ? I[src] : 0;
                if (Is == 0) continue;
                const int i = AS - (AS > al ? 1 : 0);  // derived-state index of `AS` in the source's numbering
                double rate;
                if (a.mut_uniform) rate = l_mutp[s * 3 + i];
                else {
                    const double *hm = p.hapMutType + ((int64_t)src * sites + s) * 3;
                    rate = p.mRate[(int64_t)src * sites + s] * hm[i] / (hm[0] + hm[1] + hm[2]);
                }
                drift += rate * (double)Is;
            }
        }
        // transmission (pyx:2407-2417) and incoming migration (pyx:2360-2370)
        double migI = 0.0;
        if (a.has_mig && live)
            migI = a.mig_uniform ? tau_migu(mu, a.colT[(int64_t)rep * H + hh], a.colTW[(int64_t)rep * H + hh], Ih)
                                 : a.migIn[((int64_t)rep * P + pn) * H + hh];
        const double to_st = live ? rec + samp : 0.0;
        for (int sn = 0; sn < S; ++sn) {
            double base = useL ? l_base[cb * S + sn] : p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn];
            double v = live ? base * Ih * F + base * migI : 0.0;
            drift += v;
            double red = -v + (st == sn ? to_st : 0.0);   // susceptible drift of (pn, sn)
            for (int o = 32; o > 0; o >>= 1) red += __shfl_down(red, o);
            if (lane == 0) sdS[threadIdx.x >> 6][sn] += red;
        }
        if (live && fabs(drift) >= 1e-8) {  // pyx:2440-2444, epsilon*X in single precision
            float eps = 0.03f;
            double v = (double)(eps * (float)Icell) / 2.0;
            double cand = (v > 1.0 ? v : 1.0) / fabs(drift);
            if (cand < cand_min) cand_min = cand;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_down(cand_min, o);
        if (other < cand_min) cand_min = other;
    }
    if (lane == 0) atomic_min_pos_double(&smin, cand_min);
    __syncthreads();
    // the block's part of the susceptible drift: its own slot, summed over the blocks in a fixed order by the choose kernel
    // (floating-point atomics would make tau, hence the whole run, depend on the order in which blocks finish)
    if (threadIdx.x < S) {
        double v = 0.0;
        for (int w = 0; w < TB / 64; ++w) v += sdS[w][threadIdx.x];
        a.dS_part[(((int64_t)rep * P + pn) * a.ds_nb + blockIdx.x) * S + threadIdx.x] = v;
    }
    if (threadIdx.x == 0) atomicMin(&a.tau_bits[rep], smin);
}

// ---- tiled drift (uniform mutation model, 1 <= sites <= 6 + VGX_DRIFT_HIGH_MAX) ------------------------------------------
// The gathers of the kernel above read every neighbouring 1 KiB chunk from HBM again (PMC at config 4: 17.7 GB per step for
// 1 GiB of counts).  Haplotype numbers are base-4 digit strings, so the single-site neighbours of a haplotype differ from it
// in one two-bit group: with the low VGX_DRIFT_LOW sites (4096 consecutive haplotypes) of a population in LDS all their
// neighbours are local, and the remaining high sites are handled by a first pass over tiles that hold ALL values of the high
// digits for a short run of the low ones.  Each pass reads the counts once.
#define VGX_DRIFT_LOW 6        // sites whose neighbours lie inside a 4096-haplotype tile
#define VGX_DRIFT_HIGH_MAX 4   // high sites of the first pass: up to 4^4 rows
#define VGX_MUTHIGH_CELLS 8192 // compartments per tile of the first pass (32 KB of LDS: 256 rows x 32 columns = one 128-byte line per row)
static __host__ __device__ inline bool tau_drift_tiled_ok(int sites, int mut_uniform) {
    return mut_uniform && sites >= 1 && sites <= VGX_DRIFT_LOW + VGX_DRIFT_HIGH_MAX;
}
// rate with which a source that differs from `h` at site s (allele `al` there instead of h's `AS`) mutates into h
static __device__ __forceinline__ double tau_mutp_in(const double *l_mutp, int s, int AS, int al) {
    const int i = AS - (AS > al ? 1 : 0);   // derived-state index of AS in the source's numbering
    return l_mutp[s * 3 + i];
}

// Pass 1: out[pn][h] = incoming mutation drift through the high sites.  Tile = rows (all 4^nh values of the high digits)
// x CH consecutive values of the low part; grid = (ceil(4^low_sites / CH), P, R), dynamic LDS = rows * CH * 4 bytes.
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_muthigh_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, H = p.H, sites = p.sites;
    const int nh = sites - VGX_DRIFT_LOW, rows = 1 << (2 * nh), CH = VGX_MUTHIGH_CELLS / rows < 4096 ? VGX_MUTHIGH_CELLS / rows : 4096;
    const int lowbits = 2 * VGX_DRIFT_LOW;
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    int32_t *tile = (int32_t *)hsm;   // [rows][CH]
    __shared__ double l_mutp[48];
    for (int i = threadIdx.x; i < 3 * sites && i < 48; i += TB) l_mutp[i] = a.mutp[i / 3][i % 3];
    const int32_t *I = a.I + ((int64_t)rep * P + pn) * H;
    double *out = a.mutHi + ((int64_t)rep * P + pn) * H;
    const int c0 = blockIdx.x * CH;
    for (int idx = threadIdx.x; idx < rows * CH; idx += TB) {
        const int row = idx / CH, col = idx - row * CH;
        tile[idx] = I[(row << lowbits) | (c0 + col)];
    }
    __syncthreads();
    if (a.mutHi_int) {   // one rate for every high site and derived state: the neighbour counts are summed as integers
        int32_t *outi = (int32_t *)a.mutHi + ((int64_t)rep * P + pn) * H;
        for (int idx = threadIdx.x; idx < rows * CH; idx += TB) {
            const int row = idx / CH, col = idx - row * CH;
            int sum = 0;
            for (int s = 0; s < nh; ++s) {
                const int sh = 2 * s;
                sum += tile[(row ^ (1 << sh)) * CH + col] + tile[(row ^ (2 << sh)) * CH + col] + tile[(row ^ (3 << sh)) * CH + col];
            }
            outi[(row << lowbits) | (c0 + col)] = sum;
        }
        return;
    }
    for (int idx = threadIdx.x; idx < rows * CH; idx += TB) {
        const int row = idx / CH, col = idx - row * CH;
        double d = 0.0;
        for (int s = 0; s < nh; ++s) {   // site s of the haplotype <-> two-bit group nh - 1 - s of `row`
            const int sh = 2 * (nh - 1 - s);
            const int AS = (row >> sh) & 3;
            if (l_mutp[s * 3] == l_mutp[s * 3 + 1] && l_mutp[s * 3 + 1] == l_mutp[s * 3 + 2]) {
                // equally likely derived states: the three neighbouring counts are added as integers (their sum stays
                // below the population size) and scaled once
                d += l_mutp[s * 3] * (double)(tile[(row ^ (1 << sh)) * CH + col] + tile[(row ^ (2 << sh)) * CH + col] +
                                              tile[(row ^ (3 << sh)) * CH + col]);
                continue;
            }
            for (int x = 1; x < 4; ++x) {
                const int nb = tile[(row ^ (x << sh)) * CH + col];
                if (nb != 0) d += tau_mutp_in(l_mutp, s, AS, AS ^ x) * (double)nb;
            }
        }
        out[(row << lowbits) | (c0 + col)] = d;
    }
}

// Pass 2: the drift kernel proper on tiles of 4^min(sites, VGX_DRIFT_LOW) consecutive haplotypes held in LDS.
// grid = (ceil(H / tile), P, R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_drift_tiled_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H, sites = p.sites, C = p.C, CB = p.CB;
    const int lane = threadIdx.x & 63;
    const int low = sites < VGX_DRIFT_LOW ? sites : VGX_DRIFT_LOW, nh = sites - low;
    const int TS = 1 << (2 * low);   // tile size (<= 4096 haplotypes)
    const int64_t rowoff = ((int64_t)rep * P + pn) * H;
    const int32_t *I = a.I + rowoff;
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    __shared__ int32_t tile[4096];
    __shared__ double sdS[TB / 64][64];
    __shared__ unsigned long long smin;
    __shared__ double l_cd[256], l_cs[256], l_ctm[256], l_base[16 * 64], l_mutp[48];
    __shared__ int32_t l_bidx[256], l_stype[256];
    const bool useL = C <= 256 && CB <= 16;
    sdS[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
    if (threadIdx.x == 0) smin = (unsigned long long)__double_as_longlong(1.0);
    if (useL) {
        for (int i = threadIdx.x; i < C; i += TB) {
            l_cd[i] = p.c_d[i]; l_cs[i] = p.c_s[i] * p.sampMult[pn]; l_ctm[i] = p.c_tm[i];
            l_bidx[i] = p.c_bidx[i]; l_stype[i] = p.c_stype[i];
        }
        for (int i = threadIdx.x; i < CB * S; i += TB) l_base[i] = p.cb_b[i / S] * p.cb_sigma[i] * (double)Sus[i % S];
    }
    for (int i = threadIdx.x; i < 3 * sites && i < 48; i += TB) l_mutp[i] = a.mutp[i / 3][i % 3];
    const int h0 = blockIdx.x * TS;
    for (int i = threadIdx.x * 4; i < TS; i += TB * 4) {
        if (TS >= 4) {
            const int4 v = *(const int4 *)(I + h0 + i);
            *(int4 *)(tile + i) = v;
            *(uint32_t *)(a.I8 + rowoff + h0 + i) = tau_pack8(v.x, v.y, v.z, v.w);
        } else {
            for (int j = 0; j < TS; ++j) { tile[j] = I[h0 + j]; a.I8[rowoff + h0 + j] = (uint8_t)min(I[h0 + j], 255); }
        }
    }
    __syncthreads();
    const double F = a.F[(int64_t)rep * P + pn];
    __shared__ double s_wu[16];
    MigU mu = {0.0, 0.0, 0.0, true};
    if (a.has_mig && a.mig_uniform) mu = tau_migu_setup(a, rep, pn, s_wu);
    double cand_min = 1.0;
    // a thread takes four consecutive compartments at a time (32-byte loads of the two f64 inputs); its contributions to
    // the susceptible drift of the first four groups are summed in registers and reduced once at the end
    double redS[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = threadIdx.x * 4; t0 < TS; t0 += TB * 4) {
        double mh[4] = {0.0, 0.0, 0.0, 0.0}, mg[4] = {0.0, 0.0, 0.0, 0.0};
        if (nh > 0 && a.mutHi_int) {
            const int4 v = *(const int4 *)((const int32_t *)a.mutHi + rowoff + h0 + t0);
            mh[0] = a.mutHi_rate * (double)v.x; mh[1] = a.mutHi_rate * (double)v.y; mh[2] = a.mutHi_rate * (double)v.z; mh[3] = a.mutHi_rate * (double)v.w;
        } else if (nh > 0) {
            const double4 v = *(const double4 *)(a.mutHi + rowoff + h0 + t0); mh[0] = v.x; mh[1] = v.y; mh[2] = v.z; mh[3] = v.w;
        }
        if (a.has_mig && a.mig_uniform) {   // the two column sums are shared by all populations: they stay in the caches
            const double4 v = *(const double4 *)(a.colT + (int64_t)rep * H + h0 + t0);
            const double4 u = *(const double4 *)(a.colTW + (int64_t)rep * H + h0 + t0);
            mg[0] = tau_migu(mu, v.x, u.x, (double)tile[t0]); mg[1] = tau_migu(mu, v.y, u.y, (double)tile[t0 + 1]);
            mg[2] = tau_migu(mu, v.z, u.z, (double)tile[t0 + 2]); mg[3] = tau_migu(mu, v.w, u.w, (double)tile[t0 + 3]);
        } else if (a.has_mig) {
            const double4 v = *(const double4 *)(a.migIn + rowoff + h0 + t0); mg[0] = v.x; mg[1] = v.y; mg[2] = v.z; mg[3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tt = t0 + j;
            const int hh = h0 + tt;
            const int c = (C == 1) ? 0 : p.cls[hh];
            const int cb = useL ? l_bidx[c] : p.c_bidx[c];
            const int st = useL ? l_stype[c] : p.c_stype[c];
            const int32_t Icell = tile[tt];
            const double Ih = (double)Icell;
            double drift = 0.0;
            const double rec = (useL ? l_cd[c] : p.c_d[c]) * Ih;
            const double samp = useL ? l_cs[c] * Ih : p.c_s[c] * Ih * p.sampMult[pn];
            drift -= rec;
            drift -= samp;
            drift -= (useL ? l_ctm[c] : p.c_tm[c]) * Ih;
            // incoming mutation through the low sites: neighbours inside the tile (site s <-> two-bit group sites - 1 - s)
            for (int s = nh; s < sites; ++s) {
                const int sh = 2 * (sites - s - 1);
                const int AS = (tt >> sh) & 3;
                if (l_mutp[s * 3] == l_mutp[s * 3 + 1] && l_mutp[s * 3 + 1] == l_mutp[s * 3 + 2]) {
                    drift += l_mutp[s * 3] * (double)(tile[tt ^ (1 << sh)] + tile[tt ^ (2 << sh)] + tile[tt ^ (3 << sh)]);
                    continue;
                }
                for (int x = 1; x < 4; ++x) {
                    const int nb = tile[tt ^ (x << sh)];
                    if (nb != 0) drift += tau_mutp_in(l_mutp, s, AS, AS ^ x) * (double)nb;
                }
            }
            drift += mh[j];   // ... and through the high sites (pass 1)
            const double to_st = rec + samp;
            for (int sn = 0; sn < S; ++sn) {
                const double base = useL ? l_base[cb * S + sn] : p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn];
                const double v = base * Ih * F + base * mg[j];
                drift += v;
                const double red = -v + (st == sn ? to_st : 0.0);   // susceptible drift of (pn, sn)
                if (sn < 4) redS[sn] += red;
                else if (red != 0.0) atomicAdd(&sdS[0][sn], red);   // (beyond four groups: an LDS atomic per compartment; the order of
                                                                     //  these few additions is not fixed)
            }
            if (fabs(drift) >= 1e-8) {  // pyx:2440-2444, epsilon*X in single precision
                float eps = 0.03f;
                double v = (double)(eps * (float)Icell) / 2.0;
                double cand = (v > 1.0 ? v : 1.0) / fabs(drift);
                if (cand < cand_min) cand_min = cand;
            }
        }
    }
    for (int sn = 0; sn < 4 && sn < S; ++sn) {
        double red = redS[sn];
        for (int o = 32; o > 0; o >>= 1) red += __shfl_down(red, o);
        if (lane == 0) sdS[threadIdx.x >> 6][sn] += red;
    }
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_down(cand_min, o);
        if (other < cand_min) cand_min = other;
    }
    if (lane == 0) atomic_min_pos_double(&smin, cand_min);
    __syncthreads();
    // the block's part of the susceptible drift: its own slot, summed over the blocks in a fixed order by the choose kernel
    // (floating-point atomics would make tau, hence the whole run, depend on the order in which blocks finish)
    if (threadIdx.x < S) {
        double v = 0.0;
        for (int w = 0; w < TB / 64; ++w) v += sdS[w][threadIdx.x];
        a.dS_part[(((int64_t)rep * P + pn) * a.ds_nb + blockIdx.x) * S + threadIdx.x] = v;
    }
    if (threadIdx.x == 0) atomicMin(&a.tau_bits[rep], smin);
}

// Pass 2, fast form: every low site has equally likely derived states (the usual models), at least two low sites.  A thread
// takes four consecutive compartments: their neighbours through the LAST site are the other three of the four, and through
// any other low site three 16-byte groups of the tile (t0 ^ (x << sh) keeps the low two bits), so the eighteen neighbour
// counts of a compartment cost sixteen 16-byte LDS reads per FOUR compartments and are added as integers (their sums stay
// below the population size).  The kernel also fills the histogram of compartment sizes the sieve of the halving loop works
// on (vgx_tau_sieve_hist_kernel), so that the sieve needs no pass over the compartments of its own.
// grid = (ceil(H / tile), P, R).
#define VGX_HIST_CMAX 8      // rate classes up to which the histogram is kept (else vgx_tau_sieve_kernel makes its own pass)
#define VGX_HIST_X 64        // = VGX_SIEVE_XMAX: compartment sizes 1..64
// C1: one rate class (its constants are wave-uniform), S1: one susceptibility group; the class tables are in LDS (C <= 256 and
// at most 16 transmission classes: the launcher takes the general tiled kernel otherwise).
template <bool C1, bool S1>
__global__ void __launch_bounds__(TB) vgx_tau_drift_fast_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H, sites = p.sites, C = p.C, CB = p.CB;
    const int lane = threadIdx.x & 63;
    const int low = sites < VGX_DRIFT_LOW ? sites : VGX_DRIFT_LOW, nh = sites - low;
    const int TS = 1 << (2 * low);   // tile size (16 .. 4096 haplotypes)
    const int64_t rowoff = ((int64_t)rep * P + pn) * H;
    const int32_t *I = a.I + rowoff;
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    __shared__ __attribute__((aligned(16))) int32_t tile[4096];
    __shared__ double sdS[TB / 64][64];
    __shared__ unsigned long long smin;
    __shared__ double l_cd[256], l_cs[256], l_ctm[256], l_base[16 * 64], l_rate[VGX_DRIFT_LOW];
    __shared__ int32_t l_bidx[256], l_stype[256];
    __shared__ unsigned int hist[VGX_HIST_CMAX * VGX_HIST_X * 2];   // C * 64 bins in HK copies (lanes spread over the copies)
    const bool do_hist = a.hist != nullptr;
    sdS[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
    if (threadIdx.x == 0) smin = (unsigned long long)__double_as_longlong(1.0);
    // copies of the histogram: 16 for one class ... 2 for eight (same-address LDS atomics of a wavefront serialise)
    const int HK = C == 1 ? 16 : (C == 2 ? 8 : (C <= 4 ? 4 : 2));
    if (do_hist)
        for (int i = threadIdx.x; i < C * VGX_HIST_X * HK; i += TB) hist[i] = 0;
    for (int i = threadIdx.x; i < C; i += TB) {
        l_cd[i] = p.c_d[i]; l_cs[i] = p.c_s[i] * p.sampMult[pn]; l_ctm[i] = p.c_tm[i];
        l_bidx[i] = p.c_bidx[i]; l_stype[i] = p.c_stype[i];
    }
    for (int i = threadIdx.x; i < CB * S; i += TB) l_base[i] = p.cb_b[i / S] * p.cb_sigma[i] * (double)Sus[i % S];
    if (threadIdx.x < low) l_rate[threadIdx.x] = a.mutp[nh + threadIdx.x][0];   // low site i <-> two-bit group low - 1 - i
    const int ntiles = H >> (2 * low);
    // the per-compartment inputs from global memory (high-site neighbour sums, the two column sums of uniform migration) are
    // loaded one iteration ahead — the first ones before the tile is in LDS: their latency was most of a block's life
    // (The loads are UNCONDITIONAL — an input that is not used reads the counts instead, an index past the tile its last group
    // — and the tile phase below ends with an explicit wait: vmcnt counts loads and stores together, so with a store possibly
    // pending, or with loads that only some paths issue, the compiler can only wait for everything (vmcnt(0)) where the
    // inputs are used, which would wait for the loads just issued for the NEXT iteration as well.)
    struct DIn { int4 hi; double4 cT, cTW; };
    const bool use_hi = nh > 0 && a.mutHi_int, use_col = a.has_mig && a.mig_uniform;
    const int32_t *hiP = use_hi ? (const int32_t *)a.mutHi + rowoff : I;
    const double *cTP = use_col ? a.colT + (int64_t)rep * H : (const double *)I, *cTWP = use_col ? a.colTW + (int64_t)rep * H : (const double *)I;
    auto load_in = [&](int h0, int t0) -> DIn {
        DIn d;
        const int tc = t0 < TS ? t0 : TS - 4;
        d.hi = *(const int4 *)(hiP + h0 + tc);
        const int64_t ci = use_col ? (int64_t)h0 + tc : 0;   // shared by all populations: they stay in the caches
        d.cT = *(const double4 *)(cTP + ci);
        d.cTW = *(const double4 *)(cTWP + ci);
        return d;
    };
    __syncthreads();
    const double F = a.F[(int64_t)rep * P + pn];
    __shared__ double s_wu[16];
    MigU mu = {0.0, 0.0, 0.0, true};
    if (a.has_mig && a.mig_uniform) mu = tau_migu_setup(a, rep, pn, s_wu);
    const double cd0 = l_cd[0], cs0 = l_cs[0], ctm0 = l_ctm[0], base0 = l_base[0];   // (C1: the class's constants)
    const int cb0 = l_bidx[0], st0 = l_stype[0];
    const bool one_rate = a.mutlow_same != 0;
    const double rate0 = l_rate[0];
    double cand_min = 1.0, ad_max = 0.0;
    double redS[4] = {0.0, 0.0, 0.0, 0.0};
    // a block works on several tiles of its population one after the other: its tables, the migration constants and the
    // final reductions are paid once (they were most of a one-tile block's life)
    for (int tb = blockIdx.x; tb < ntiles; tb += gridDim.x) {
    const int h0 = tb * TS;
    DIn nxt = load_in(h0, threadIdx.x * 4);
    __syncthreads();   // the previous tile has been read by everybody
    {   // the tile (at most 4096 = 4 * 4 * TB counts): all of a thread's loads in flight at once
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (threadIdx.x + u * TB) * 4;
            v[u] = *(const int4 *)(I + h0 + (i < TS ? i : 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (threadIdx.x + u * TB) * 4;
            if (i < TS) {
                *(int4 *)(tile + i) = v[u];
                *(uint32_t *)(a.I8 + rowoff + h0 + i) = tau_pack8(v[u].x, v[u].y, v[u].z, v[u].w);   // the one-byte copy vgx_tau_scan_kernel streams
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no store pending in the loop below (see load_in)
    }
    __syncthreads();
    for (int t0 = threadIdx.x * 4; t0 < TS; t0 += TB * 4) {
        const DIn in = nxt;
        nxt = load_in(h0, t0 + TB * 4);
        double mh[4] = {0.0, 0.0, 0.0, 0.0}, mg[4] = {0.0, 0.0, 0.0, 0.0};
        if (use_hi) {   // (nh > 0 without integer sums: the launcher takes the general tiled kernel)
            const int4 v = in.hi;
            mh[0] = a.mutHi_rate * (double)v.x; mh[1] = a.mutHi_rate * (double)v.y; mh[2] = a.mutHi_rate * (double)v.z; mh[3] = a.mutHi_rate * (double)v.w;
        }
        const int4 own = *(const int4 *)(tile + t0);
        const int Iv[4] = {own.x, own.y, own.z, own.w};
        if (use_col) {
            const double4 v = in.cT, u = in.cTW;
            mg[0] = tau_migu(mu, v.x, u.x, (double)Iv[0]); mg[1] = tau_migu(mu, v.y, u.y, (double)Iv[1]);
            mg[2] = tau_migu(mu, v.z, u.z, (double)Iv[2]); mg[3] = tau_migu(mu, v.w, u.w, (double)Iv[3]);
        }   // (migration that is not uniform: the general tiled kernel)
        // incoming mutation through the low sites
        const int s4 = own.x + own.y + own.z + own.w;
        int nb[4] = {s4 - own.x, s4 - own.y, s4 - own.z, s4 - own.w};   // last site: the other three of the four
        double mlow[4] = {0.0, 0.0, 0.0, 0.0};
        if (!one_rate) {
            const double r = l_rate[low - 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { mlow[j] = r * (double)nb[j]; nb[j] = 0; }
        }
        for (int g = 1; g < low; ++g) {
            const int sh = 2 * g;
            const int4 x1 = *(const int4 *)(tile + (t0 ^ (1 << sh))), x2 = *(const int4 *)(tile + (t0 ^ (2 << sh))), x3 = *(const int4 *)(tile + (t0 ^ (3 << sh)));
            const int sx[4] = {x1.x + x2.x + x3.x, x1.y + x2.y + x3.y, x1.z + x2.z + x3.z, x1.w + x2.w + x3.w};
            if (one_rate) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nb[j] += sx[j];
            } else {
                const double r = l_rate[low - 1 - g];
#pragma unroll
                for (int j = 0; j < 4; ++j) mlow[j] += r * (double)sx[j];
            }
        }
        if (one_rate) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mlow[j] = rate0 * (double)nb[j];
        }
        int cl[4] = {0, 0, 0, 0};
        if (!C1) { const int4 cc = *(const int4 *)(p.cls + h0 + t0); cl[0] = cc.x; cl[1] = cc.y; cl[2] = cc.z; cl[3] = cc.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = C1 ? 0 : cl[j];
            const int cb = C1 ? cb0 : l_bidx[c];
            const int st = C1 ? st0 : l_stype[c];
            const int32_t Icell = Iv[j];
            const double Ih = (double)Icell;
            double drift = 0.0;
            const double rec = (C1 ? cd0 : l_cd[c]) * Ih;
            const double samp = (C1 ? cs0 : l_cs[c]) * Ih;
            drift -= rec;
            drift -= samp;
            drift -= (C1 ? ctm0 : l_ctm[c]) * Ih;
            drift += mlow[j];
            drift += mh[j];   // ... and through the high sites (pass 1)
            const double to_st = rec + samp;
            if (S1) {
                const double base = C1 ? base0 : l_base[cb];
                const double v = base * Ih * F + base * mg[j];
                drift += v;
                redS[0] += to_st - v;   // susceptible drift of (pn, 0)
            } else {
                for (int sn = 0; sn < S; ++sn) {
                    const double base = l_base[cb * S + sn];
                    const double v = base * Ih * F + base * mg[j];
                    drift += v;
                    const double red = -v + (st == sn ? to_st : 0.0);   // susceptible drift of (pn, sn)
                    if (sn < 4) redS[sn] += red;
                    else if (red != 0.0) atomicAdd(&sdS[0][sn], red);
                }
            }
            // pyx:2440-2444: candidate max(eps * X / 2, 1) / |drift| (epsilon * X in single precision).  For the compartments
            // with up to 66 hosts the numerator is 1: the smallest candidate is 1 / (largest |drift|), one division per thread
            // at the end instead of one per compartment (the quotient is monotone in |drift|, so it is the same number)
            const double ad = fabs(drift);
            const double v = (double)(0.03f * (float)Icell) / 2.0;
            const bool large = v > 1.0;                       // more than 66 hosts: its own numerator (rare here)
            ad_max = fmax(ad_max, large ? 0.0 : ad);          // (|drift| < 1e-8 is sorted out at the end)
            if (__any(large && ad >= 1e-8)) {
                if (large && ad >= 1e-8) cand_min = fmin(cand_min, v / ad);
            }
            if (do_hist && Icell >= 1 && Icell <= VGX_HIST_X)
                atomicAdd(&hist[(c * VGX_HIST_X + Icell - 1) * HK + (lane & (HK - 1))], 1u);
        }
    }
    }   // tiles
    for (int sn = 0; sn < 4 && sn < S; ++sn) {
        double red = redS[sn];
        for (int o = 32; o > 0; o >>= 1) red += __shfl_down(red, o);
        if (lane == 0) sdS[threadIdx.x >> 6][sn] += red;
    }
    if (ad_max >= 1e-8 && 1.0 / ad_max < cand_min) cand_min = 1.0 / ad_max;
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_down(cand_min, o);
        if (other < cand_min) cand_min = other;
    }
    if (lane == 0) atomic_min_pos_double(&smin, cand_min);
    __syncthreads();
    // the block's part of the susceptible drift: its own slot, summed over the blocks in a fixed order by the choose kernel
    // (floating-point atomics would make tau, hence the whole run, depend on the order in which blocks finish)
    if (threadIdx.x < S) {
        double v = 0.0;
        for (int w = 0; w < TB / 64; ++w) v += sdS[w][threadIdx.x];
        a.dS_part[(((int64_t)rep * P + pn) * a.ds_nb + blockIdx.x) * S + threadIdx.x] = v;
    }
    if (threadIdx.x == 0) atomicMin(&a.tau_bits[rep], smin);
    if (do_hist)
        for (int i = threadIdx.x; i < C * VGX_HIST_X; i += TB) {
            unsigned int v = 0;
            for (int k = 0; k < HK; ++k) v += hist[i * HK + k];
            if (v) atomicAdd(&a.hist[((int64_t)rep * P + pn) * C * VGX_HIST_X + i], v);
        }
}

// ---- the drift pass on the ONE-BYTE counts (config 4's shape: 7 to 10 sites, one rate class, the uniform mutation model with
// equally likely derived states, one rate for the low and one for the high sites, no or uniform migration) -------------------
// The two-pass form above reads the 4-byte counts three times per step (column sums, high-site pass, low-site pass) and moves the
// high-site neighbour sums through memory: 5.9 GB per step at config 4 before a single event is drawn.  Here I8 = min(count, 255)
// is kept in step with the counts (vgx_tau_sync8_kernel after the accepted try's list has been applied; vgx_tau_conv8_kernel
// after an upload) and the drift pass READS it: a block stages a tile of 4^8 haplotypes = 64 KB of bytes in LDS — the neighbours
// through the last eight sites are inside it — and fetches the neighbours through the first one or two sites (three or six
// other tiles of the same population row) from global memory, where the block-to-XCD mapping keeps them in the L2 of the XCD
// the row's tiles run on.  No intermediate array, one read of 1 byte per compartment.  Neighbour counts are summed as integers
// in 16-bit lanes (two compartments per register), separately for the low six and the high sites, and enter the drift with the
// terms of vgx_tau_drift_fast_kernel collected (tau agrees with the two-pass form's to rounding).  A byte of 255 stands for
// "255 or more": every tile's largest byte is kept (tmax8: an upper bound between two conversions); blocks whose tiles may hold
// a 255 test every byte they add, and a wavefront that meets one forms its compartments' sums again from the 4-byte counts.
#define VGX_D8_LOW 8
#define D8_TB 512       // two blocks per CU (2 x 77 KB of LDS, 16 wavefronts); one block of 1024 threads: 1.8 instead of 2.4 GB fetched per launch, but 0.64 instead of 0.50 ms (a block's load phase is not covered by another block's compute phase)
static_assert(D8_TB / 64 == VGX_D8_WAVES, "regions of the occupied-compartment lists");
static __device__ __forceinline__ uint32_t d8_even(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0C020C00u); }   // bytes 0, 2 -> 16-bit lanes
static __device__ __forceinline__ uint32_t d8_odd(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0C030C01u); }    // bytes 1, 3
static __device__ __forceinline__ uint32_t d8_sat(uint32_t x) { return ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u; }   // some byte == 255

// I8 = min(I, 255) and the largest byte of every tile, for the whole state.  grid = (ceil(H / (4 TB)), P, R); tmax8 zeroed before.
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_conv8_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    const int P = a.p.P, H = a.p.H;
    const int h0 = (blockIdx.x * TB + threadIdx.x) * 4;      // (H is a multiple of 4 TB from five sites on; this pass: seven and more)
    const int64_t off = ((int64_t)rep * P + pn) * H + h0;
    const int4 v = *(const int4 *)(a.I + off);
    *(uint32_t *)(a.I8 + off) = tau_pack8(v.x, v.y, v.z, v.w);
    // the largest byte of the block's 4 TB compartments (they lie inside one tile): one atomic per block
    __shared__ unsigned int s_m[TB / 64];
    unsigned int m = (unsigned int)min(max(max(v.x, v.y), max(v.z, v.w)), 255);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_down((int)m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TB / 64; ++w) m = max(m, s_m[w]);
        atomicMax(a.tmax8 + ((int64_t)rep * P + pn) * a.nt8 + (h0 >> (2 * VGX_D8_LOW)), m);
    }
}

// Column sums over the populations from the one-byte counts (a saturated byte: the 4-byte counts).  grid = (ceil(H / (4 TB)), R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_colsum8_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, H = p.H;
    __shared__ double s_w[VGX_MIGU_PMAX];
    __shared__ int s_neq;
    if (threadIdx.x == 0) s_neq = 0;
    for (int q = threadIdx.x; q < P; q += TB) s_w[q] = a.cd[(int64_t)rep * P + q] / p.actualSizes[q];
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += TB)
        if (s_w[q] != s_w[0]) s_neq = 1;
    __syncthreads();
    const bool weq = s_neq == 0;      // one common weight w: TW = w T, formed by the drift pass (no second array to stream)
    const int h0 = (blockIdx.x * TB + threadIdx.x) * 4;
    if (h0 >= H) return;
    const uint8_t *I8 = a.I8 + (int64_t)rep * P * H;
    const int32_t *I = a.I + (int64_t)rep * P * H;
    long long t[4] = {0, 0, 0, 0};
    double tw[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q0 = 0; q0 < P; q0 += 8) {      // eight rows' loads in flight together
        uint32_t xs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xs[u] = *(const uint32_t *)(I8 + (int64_t)min(q0 + u, P - 1) * H + h0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u;
            if (q >= P) break;
            int x[4] = {(int)(xs[u] & 255u), (int)((xs[u] >> 8) & 255u), (int)((xs[u] >> 16) & 255u), (int)(xs[u] >> 24)};
            if (d8_sat(xs[u])) { const int4 v = *(const int4 *)(I + (int64_t)q * H + h0); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
            const double w = s_w[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) { t[j] += x[j]; tw[j] += w * (double)x[j]; }
        }
    }
    for (int j = 0; j < 4; ++j) a.colT[(int64_t)rep * H + h0 + j] = (double)t[j];
    if (!weq)
        for (int j = 0; j < 4; ++j) a.colTW[(int64_t)rep * H + h0 + j] = tw[j];
}

// grid = (8 * ceil(P / 8) * nt8, R): flattened (population, tile) with the tiles of one population on one XCD (workgroups go to
// the XCDs round-robin in the order of their flattened index); dynamic LDS = 4^min(sites, 8) bytes.
// The drift of a compartment is linear in its own count Ih, its neighbour counts and the column sums of uniform migration:
//   drift = kI Ih + rate_lo nlo + rate_hi nhi + B (c1 T + c2 TW),   B = sum_sn b sigma_sn S_sn,
//   kI = B F - (d + s + m) - B (c1 + c2 wt)
// (the terms of vgx_tau_drift_fast_kernel collected; fused multiply-adds), and the susceptible compartments' drift is a function
// of the block's sums of Ih and of c1 T + c2 TW, formed once per thread.  MODE: 0 = no count above 66 in the tile and the tiles it
// reads, so none of its compartments has its own numerator in ChooseTau either (the three neighbours through a site are added as
// packed bytes before they are widened; up to 85 would do for that), 1 = no count of 255 or more,
// 2 = every byte is tested for 255 and a wavefront that meets one forms its sums again from the 4-byte counts.
struct D8Ctx {
    const uint32_t *tile32, *row32;
    const int32_t *Irow;
    const double *cTP, *cTWP;
    int tl, low, ntop, TSd, sites;
    bool use_col, use_tw, do_hist, one_rate, skip_empty, listed;
    const uint16_t *queue;           // sparse states: the tile's dwords that hold a host (d8_scan), or null: all of them in order
    int nq;
    double kI, rate_lo, rate_hi, Bsum, c1, c2;
    float kI_a, rlo_a, rhi_a, B_a;   // their magnitudes in single precision, rounded up: the screen of d8_cells
    float kI_f, rlo_f, rhi_f, B_f;   // and the coefficients themselves (its second level)
    bool opposed;                    // kI < 0 and every other coefficient >= 0: the own term and the arrivals have opposite signs
    unsigned int *hist;
    int32_t *occ_dst;      // this wavefront's region of the occupied-compartment lists, or null
};
#ifdef VGX_D8_STATS
__device__ unsigned long long vgx_d8_stats[8];   // diagnostic build: wave turns, turns past level 1, past level 2, sum of thr0 > 1.5
#endif
// the coefficients of a compartment's drift in population pn (see D8Ctx): one function for the dense and the sparse form
static __device__ __forceinline__ void d8_coeffs(const MigU &mu, bool use_tw, double Bsum, double F, double out_rate, double &c1, double &c2, double &kmig,
                                                 double &kI) {
    c1 = use_tw ? mu.c1 : mu.c1 + mu.c2 * mu.wt;
    c2 = mu.c2;
    kmig = mu.c1 + mu.c2 * mu.wt;
    kI = Bsum * F - out_rate - Bsum * kmig;
}
// sparse states: the tile's occupied compartments listed region by region (as d8_cells lists them when it takes every dword in order) and
// the dwords that hold a host gathered in LDS — d8_cells then takes only those, 64 per wavefront and turn whatever their place in the
// tile (at SURVEY 8(d)'s start state six of ten turns of 256 compartments hold a host, one dword in seventy does).
#define VGX_D8_QCAP 1024
static __device__ __forceinline__ void d8_scan(const D8Ctx &c, int lane, int TSd, int low, int &occ_cnt, uint16_t *s_queue, int *s_nq) {
    for (int q = threadIdx.x; q < TSd; q += D8_TB) {
        const int h = (c.tl << (2 * low)) + 4 * q;
        const uint32_t own = c.tile32[q];
        if (!__any(own != 0u)) continue;
        if (c.occ_dst) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool nz = ((own >> (8 * j)) & 255u) != 0u;
                const unsigned long long m = __ballot(nz);
                if (m == 0ull) continue;
                if (nz) {
                    const int pos = occ_cnt + (int)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < VGX_OCC_CAP) c.occ_dst[pos] = h + j;
                }
                occ_cnt += (int)__popcll(m);
            }
        } else {
            const uint32_t y = own;
            const uint32_t zf = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
            occ_cnt += 4 - __popc(zf);
        }
        if (own != 0u) {
            const int pos = atomicAdd(s_nq, 1);
            if (pos < VGX_D8_QCAP) s_queue[pos] = (uint16_t)q;
        }
    }
}
template <int MODE, bool USUAL>
static __device__ __forceinline__ void d8_cells(const D8Ctx &c, int lane, double &cand_min, double &ad_max, long long &sumI, double &sumMg,
                                                int &occ_cnt, float thr, int &imax) {
    // USUAL: the shape of config 4 and of most models the kernel takes — eight sites inside the tile, uniform migration with one common
    // weight, one mutation rate — with the switches below known at compile time (one straight-line turn: every neighbour read in flight
    // together, no selects between forms that are not needed)
    const int k_low = USUAL ? VGX_D8_LOW : c.low, k_TSd = USUAL ? (1 << (2 * VGX_D8_LOW - 2)) : c.TSd;
    const bool k_col = USUAL ? true : c.use_col, k_tw = USUAL ? false : c.use_tw, k_one = USUAL ? true : c.one_rate;
    const int bound = c.queue ? c.nq : k_TSd;
    for (int it = threadIdx.x; (it & ~63) < bound; it += D8_TB) {      // (whole wavefronts: the turn's screen is a wavefront's)
        const bool valid = it < bound;                                  // (beyond the gathered dwords: an empty dword that asks for nothing)
        const int q = c.queue ? (valid ? (int)c.queue[it] : 0) : it;
        const int h = (c.tl << (2 * k_low)) + 4 * q;          // first of the thread's four haplotypes
        const uint32_t own = valid ? c.tile32[q] : 0u;
        // sparse states (VgxTauArgs.drift_sparse): a turn whose 256 compartments are all empty is left out — what arrives in an empty
        // compartment is looked at by vgx_tau_drift8s_heavy_kernel / _col_kernel where it can matter
        if (c.skip_empty && !__any(own != 0u)) continue;
        if (c.listed) {
        } else if (c.occ_dst) {       // (wave-uniform) the occupied compartments among the wavefront's 256: listed, region by region
            if (__any(own != 0u)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool nz = ((own >> (8 * j)) & 255u) != 0u;
                const unsigned long long m = __ballot(nz);
                if (m == 0ull) continue;
                if (nz) {
                    const int pos = occ_cnt + (int)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < VGX_OCC_CAP) c.occ_dst[pos] = h + j;
                }
                occ_cnt += (int)__popcll(m);
            }
            }
        } else {               // only counted (what decides whether the next step builds the lists)
            const uint32_t y = own;
            const uint32_t zf = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);   // bit 8 j + 7: byte j is zero
            occ_cnt += 4 - __popc(zf);
        }
        uint32_t loE = 0, loO = 0, hiE = 0, hiO = 0, bad = MODE == 2 ? d8_sat(own) : 0u;
        uint32_t hv[6];     // the neighbours through the sites above the tile (other tiles of the row), issued first
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int x = 1; x < 4; ++x) {
                const int k = g * 3 + x - 1;
                hv[k] = 0;
                if (g < c.ntop) hv[k] = c.row32[(int64_t)(c.tl ^ (x << (2 * g))) * k_TSd + q];
            }
        double4 cT = {0.0, 0.0, 0.0, 0.0}, cTW = {0.0, 0.0, 0.0, 0.0};
        if (k_col) cT = *(const double4 *)(c.cTP + h);
        if (k_tw) cTW = *(const double4 *)(c.cTWP + h);
        // inside the tile: digit g >= 1 of the cell index = two-bit group g - 1 of the dword index
#pragma unroll
        for (int g = 1; g < VGX_D8_LOW; ++g) {
            if (g < k_low) {
                const int sh = 2 * g - 2;
                // (the byte offset q * 4 is flipped, not the index: one v_xor per read, the tile's base in the read's offset field)
                const char *tb = (const char *)c.tile32;
                const uint32_t x1 = *(const uint32_t *)(tb + ((q << 2) ^ (4 << sh))), x2 = *(const uint32_t *)(tb + ((q << 2) ^ (8 << sh))),
                               x3 = *(const uint32_t *)(tb + ((q << 2) ^ (12 << sh)));
                if (MODE == 2) bad |= d8_sat(x1) | d8_sat(x2) | d8_sat(x3);
                uint32_t e, o;
                if (MODE == 0) { const uint32_t t = x1 + x2 + x3; e = d8_even(t); o = d8_odd(t); }
                else { e = d8_even(x1) + d8_even(x2) + d8_even(x3); o = d8_odd(x1) + d8_odd(x2) + d8_odd(x3); }
                if (g < VGX_DRIFT_LOW) { loE += e; loO += o; } else { hiE += e; hiO += o; }
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g < c.ntop) {
                const uint32_t x1 = hv[g * 3], x2 = hv[g * 3 + 1], x3 = hv[g * 3 + 2];
                if (MODE == 2) bad |= d8_sat(x1) | d8_sat(x2) | d8_sat(x3);
                if (MODE == 0) { const uint32_t t = x1 + x2 + x3; hiE += d8_even(t); hiO += d8_odd(t); }
                else { hiE += d8_even(x1) + d8_even(x2) + d8_even(x3); hiO += d8_odd(x1) + d8_odd(x2) + d8_odd(x3); }
            }
        }
        int Iv[4] = {(int)(own & 255u), (int)((own >> 8) & 255u), (int)((own >> 16) & 255u), (int)(own >> 24)};
        const int s4 = Iv[0] + Iv[1] + Iv[2] + Iv[3];
        int nlo[4] = {(int)(loE & 0xFFFFu) + s4 - Iv[0], (int)(loO & 0xFFFFu) + s4 - Iv[1], (int)(loE >> 16) + s4 - Iv[2], (int)(loO >> 16) + s4 - Iv[3]};
        int nhi[4] = {(int)(hiE & 0xFFFFu), (int)(hiO & 0xFFFFu), (int)(hiE >> 16), (int)(hiO >> 16)};
        if (MODE == 2 && __any(bad != 0u && valid)) {
            if (bad != 0u && valid) {      // a count of 255 or more among the bytes added: the sums again from the 4-byte counts
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hh = h + j;
                    Iv[j] = c.Irow[hh];
                    int sl = 0, shh = 0;
                    for (int g = 0; g < c.sites; ++g) {
                        const int v3 = c.Irow[hh ^ (1 << (2 * g))] + c.Irow[hh ^ (2 << (2 * g))] + c.Irow[hh ^ (3 << (2 * g))];
                        if (g < VGX_DRIFT_LOW) sl += v3; else shh += v3;
                    }
                    nlo[j] = sl; nhi[j] = shh;
                }
            }
        }
        sumI += (MODE == 2) ? (long long)Iv[0] + (long long)Iv[1] + (long long)Iv[2] + (long long)Iv[3] : (long long)s4;
        imax = max(imax, max(max(Iv[0], Iv[1]), max(Iv[2], Iv[3])));
        // pyx:2440-2444: candidate max(eps * X / 2, 1) / |drift| with eps * X in single precision; the numerator is 1 up to 66
        // hosts ((double)(0.03f * (float)X) / 2 > 1 from X = 67 on), and the smallest of those candidates is 1 / (largest |drift|).
        // All the kernel keeps of the |drift| values is the launch's smallest candidate (tau_bits, an atomic minimum that starts
        // at 1.0): a compartment of up to 66 hosts matters only if its |drift| exceeds 1 / (the smallest candidate anywhere so far).
        // So the four compartments of a turn are first bounded together in single precision — |drift| <= |kI| max Ih + rate max n +
        // B max |mg|, coefficients rounded up by 4e-6 (far above the roundings of the conversions and the three operations) — and
        // only turns whose bound reaches thr (or that hold a large compartment) take the double-precision path; thr rises with what
        // the wavefront finds.  Same minimum as without the screen, bit for bit; the kernel's vector pipes were two thirds busy
        // with the fifteen double-precision instructions per compartment of this section.
        double mg4[4] = {0.0, 0.0, 0.0, 0.0};
        float mgmax = 0.0f;
        if (k_col) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double T = j == 0 ? cT.x : j == 1 ? cT.y : j == 2 ? cT.z : cT.w, TW = j == 0 ? cTW.x : j == 1 ? cTW.y : j == 2 ? cTW.z : cTW.w;
                mg4[j] = k_tw ? __builtin_fma(c.c2, TW, c.c1 * T) : c.c1 * T;     // (one common weight: c1 holds c1 + c2 w)
                sumMg += mg4[j];
            }
            mgmax = fmaxf(fmaxf(fabsf((float)mg4[0]), fabsf((float)mg4[1])), fmaxf(fabsf((float)mg4[2]), fabsf((float)mg4[3])));
        }
        const int ihmax = max(max(Iv[0], Iv[1]), max(Iv[2], Iv[3]));
        // (the arrivals' terms are not negative; with kI < 0 — the usual sign — the drift lies in [-|kI| max Ih, rate max n + B max mg])
        const float Uneg = c.kI_a * (float)ihmax;
        float Upos = k_col ? c.B_a * mgmax : 0.0f;
        if (k_one) Upos = fmaf(c.rlo_a, (float)max(max(nlo[0] + nhi[0], nlo[1] + nhi[1]), max(nlo[2] + nhi[2], nlo[3] + nhi[3])), Upos);
        else Upos = fmaf(c.rhi_a, (float)max(max(nhi[0], nhi[1]), max(nhi[2], nhi[3])), fmaf(c.rlo_a, (float)max(max(nlo[0], nlo[1]), max(nlo[2], nlo[3])), Upos));
        const float U = c.opposed ? fmaxf(Upos, Uneg) : Upos + Uneg * (1.0f + 1e-6f);
#ifdef VGX_D8_NOSCREEN      // (timing comparisons: every compartment takes the exact path)
        unsigned int need = U > -1.0f ? 15u : 0u;
        thr = -1.0f;
#else
        unsigned int need = (U >= thr || (MODE != 0 && ihmax > 66) || !(U < 3.0e38f)) ? 15u : 0u;
#endif
        need = valid ? need : 0u;
#ifdef VGX_D8_STATS
        if (lane == 0) atomicAdd(&vgx_d8_stats[0], 1ull);
        if (__any(need != 0u) && lane == 0) atomicAdd(&vgx_d8_stats[1], 1ull);
        atomicAdd(&vgx_d8_stats[3], (unsigned long long)(need != 0u));
        if (lane == 0 && q < D8_TB) { atomicAdd(&vgx_d8_stats[5], (unsigned long long)(thr > 1.5f)); atomicAdd(&vgx_d8_stats[6], (unsigned long long)thr); atomicAdd(&vgx_d8_stats[7], (unsigned long long)U); }
#endif
        if (c.do_hist) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (Iv[j] >= 1 && Iv[j] <= VGX_HIST_X) atomicAdd(&c.hist[(Iv[j] - 1) * 16 + (lane & 15)], 1u);
        }
        if (__any(need != 0u)) {
            // second level, where the arrivals are nearly the same everywhere and the turn's bound says little: the compartments one
            // by one in single precision — off the exact drift by less than 9 * 2^-24 * (sum of the terms' magnitudes) <= 6e-7 (Upos + Uneg)
            const float cut = thr - 2e-6f * (Upos + Uneg);
            unsigned int need2 = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float af = c.kI_f * (float)Iv[j];
                if (k_one) af = fmaf(c.rlo_f, (float)(nlo[j] + nhi[j]), af);
                else af = fmaf(c.rhi_f, (float)nhi[j], fmaf(c.rlo_f, (float)nlo[j], af));
                if (k_col) af = fmaf(c.B_f, (float)mg4[j], af);
                need2 |= ((fabsf(af) >= cut || (MODE != 0 && Iv[j] > 66) || !(U < 3.0e38f)) ? 1u : 0u) << j;
            }
            need &= need2;
#ifdef VGX_D8_STATS
            if (__any(need != 0u) && lane == 0) atomicAdd(&vgx_d8_stats[2], 1ull);
            atomicAdd(&vgx_d8_stats[4], (unsigned long long)__popc(need));
#endif
        }
        if (__any(need != 0u)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((need >> j) & 1u)) continue;
                const int32_t Icell = Iv[j];
                double drift = c.kI * (double)Icell;
                if (k_one) drift = __builtin_fma(c.rate_lo, (double)(nlo[j] + nhi[j]), drift);
                else drift = __builtin_fma(c.rate_hi, (double)nhi[j], __builtin_fma(c.rate_lo, (double)nlo[j], drift));
                if (k_col) drift = __builtin_fma(c.Bsum, mg4[j], drift);
                const double ad = fabs(drift);
                const bool large = MODE != 0 && Icell > 66;
                ad_max = fmax(ad_max, large ? 0.0 : ad);          // (|drift| < 1e-8 is sorted out at the end)
                if (large && ad >= 1e-8) cand_min = fmin(cand_min, ((double)(0.03f * (float)Icell) / 2.0) / ad);
            }
            float m = (float)ad_max * (1.0f - 1e-6f);             // (below ad_max whatever the conversion's rounding)
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            thr = fmaxf(thr, m);
        }
    }
}

extern "C" __global__ void __launch_bounds__(D8_TB, 4) vgx_tau_drift8_kernel(VgxTauArgs a) {   // (four wavefronts per SIMD = two blocks per CU: at most 128 VGPRs)
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H, sites = p.sites;
    const int nt = a.nt8;
    const unsigned f = blockIdx.x;
    const int xcd = (int)(f & 7u);
    const unsigned sq = f >> 3;
    const int tl = (int)(sq % (unsigned)nt), pn = (int)(sq / (unsigned)nt) * 8 + xcd, rep = blockIdx.y;
    if (pn >= P || !a.active[rep]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int low = sites < VGX_D8_LOW ? sites : VGX_D8_LOW, ntop = sites - low;     // sites inside the tile / above it
    const int TSd = 1 << (2 * low - 2);                                              // dwords of a tile
    const int64_t rowoff = ((int64_t)rep * P + pn) * H;
    const uint32_t *row32 = (const uint32_t *)(a.I8 + rowoff);
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    extern __shared__ __attribute__((aligned(16))) unsigned char d8sm[];
    uint32_t *tile32 = (uint32_t *)d8sm;
    __shared__ double s_sum[D8_TB / 64][2];
    __shared__ unsigned long long smin;
    __shared__ double l_base[64];
    __shared__ unsigned int hist[VGX_HIST_X * 16];
    __shared__ double s_wu[16];
    __shared__ unsigned int s_mx;
    __shared__ uint16_t s_queue[VGX_D8_QCAP];
    __shared__ int s_nq;
    const bool do_hist = a.hist != nullptr;
    if (a.drift_sparse && a.tmax8[((int64_t)rep * P + pn) * nt + tl] == 0u) {
        // sparse states: a tile that has held nobody since the bytes were last converted (the early epidemic: most of them) — nothing to
        // list, no drift to form, no byte to load
        if (threadIdx.x < VGX_D8_WAVES) {
            const int64_t region = ((int64_t)rep * P + pn) * a.occ_nreg + (int64_t)tl * VGX_D8_WAVES + threadIdx.x;
            if (a.build_occ) a.occ_n[region] = 0u;
            a.d8s_regmax[region] = 0;
        }
        if (threadIdx.x == 0) a.tI_pt[((int64_t)rep * P + pn) * nt + tl] = 0ull;
        return;
    }
    if (threadIdx.x == 0) {
        s_nq = 0;
        smin = (unsigned long long)__double_as_longlong(1.0);
        const unsigned int *tm = a.tmax8 + ((int64_t)rep * P + pn) * nt;
        unsigned int c = tm[tl];
        for (int g = 0; g < ntop; ++g)
            for (int x = 1; x < 4; ++x) c = max(c, tm[tl ^ (x << (2 * g))]);
        s_mx = c;
    }
    if (do_hist)
        for (int i = threadIdx.x; i < VGX_HIST_X * 16; i += D8_TB) hist[i] = 0;
    for (int i = threadIdx.x; i < S; i += D8_TB) l_base[i] = p.cb_b[0] * p.cb_sigma[i] * (double)Sus[i];
    // the tile: all of a thread's loads in flight at once
    {
        const uint32_t *src = row32 + (int64_t)tl * TSd;
        for (int i = threadIdx.x * 4; i < TSd; i += D8_TB * 4) *(uint4 *)(tile32 + i) = *(const uint4 *)(src + i);
    }
    __syncthreads();
    const unsigned int mx = s_mx;
    const double F = a.F[(int64_t)rep * P + pn];
    const bool use_col = a.has_mig && a.mig_uniform;
    MigU mu = {0.0, 0.0, 0.0, true};
    if (use_col && !a.drift_sparse) mu = tau_migu_setup(a, rep, pn, s_wu);     // (sparse: from vgx_tau_drift8s_prep_kernel's record, below)
    double Bsum = 0.0;
    for (int sn = 0; sn < S; ++sn) Bsum += l_base[sn];
    const double cd0 = p.c_d[0], cs0 = p.c_s[0] * p.sampMult[pn], ctm0 = p.c_tm[0];
    const int st0 = p.c_stype[0];
    D8Ctx c;
    c.tile32 = tile32; c.row32 = row32; c.Irow = a.I + rowoff;
    c.cTP = a.colT + (int64_t)rep * H; c.cTWP = a.colTW + (int64_t)rep * H;
    c.tl = tl; c.low = low; c.ntop = ntop; c.TSd = TSd; c.sites = sites;
    c.use_col = use_col; c.do_hist = do_hist;
    // one common weight w = cd / actualSizes for all populations (the usual case): TW = w T, no second array
    c.use_tw = use_col && !mu.weq;
    // the LOW six sites of the model (VGX_DRIFT_LOW: the rate a.mutp[nh + .]) are digits 0..5 of the cell index, the high ones
    // digits 6.. and the tiles above
    c.rate_lo = a.mutp[sites - VGX_DRIFT_LOW][0]; c.rate_hi = a.mutHi_rate;
    c.one_rate = c.rate_lo == c.rate_hi;
    double kmig;                                       // mg = (c1 T + c2 TW) - kmig Ih
    c.Bsum = Bsum;
    d8_coeffs(mu, c.use_tw, Bsum, F, cd0 + cs0 + ctm0, c.c1, c.c2, kmig, c.kI);
    if (a.drift_sparse) {     // the same values, formed once per population by vgx_tau_drift8s_prep_kernel
        const double *pk = a.d8s_pk + ((int64_t)rep * P + pn) * 8;
        c.c1 = pk[1]; c.c2 = pk[2]; c.kI = pk[3]; kmig = pk[6]; c.use_tw = pk[7] != 0.0;
    }
    c.hist = hist;
    c.kI_a = fabsf((float)c.kI) * (1.0f + 4e-6f); c.rlo_a = fabsf((float)c.rate_lo) * (1.0f + 4e-6f); c.rhi_a = fabsf((float)c.rate_hi) * (1.0f + 4e-6f);
    c.B_a = fabsf((float)c.Bsum) * (1.0f + 4e-6f);
    c.kI_f = (float)c.kI; c.rlo_f = (float)c.rate_lo; c.rhi_f = (float)c.rate_hi; c.B_f = (float)c.Bsum;
    c.opposed = c.kI < 0.0 && c.rate_lo >= 0.0 && c.rate_hi >= 0.0 && c.Bsum >= 0.0 && c.c1 >= 0.0 && (!c.use_tw || c.c2 >= 0.0);
    // 1 / (the smallest candidate of the launch so far: the blocks that have finished), a little less; at least 1 - 1e-6
    const float thr0 = (float)(1.0 / __longlong_as_double((long long)a.tau_bits[rep])) * (1.0f - 1e-6f);
    double cand_min = 1.0, ad_max = 0.0, sumMg = 0.0;
    long long sumI = 0;
    int occ_cnt = 0;       // listing: wave-uniform, entries of this wavefront's region; counting: the lane's occupied compartments
    const int64_t region = ((int64_t)rep * P + pn) * a.occ_nreg + (int64_t)tl * (D8_TB / 64) + wv;
    c.occ_dst = a.build_occ ? a.occ + region * VGX_OCC_CAP : nullptr;
    const bool usual = low == VGX_D8_LOW && c.use_col && !c.use_tw && c.one_rate;
    c.skip_empty = a.drift_sparse != 0;
    c.listed = false; c.queue = nullptr; c.nq = 0;
    int imax = 0;          // the lane's largest count
    if (a.drift_sparse) {
        d8_scan(c, lane, TSd, low, occ_cnt, s_queue, &s_nq);
        __syncthreads();
        c.listed = true;
        if (s_nq <= VGX_D8_QCAP) { c.queue = s_queue; c.nq = s_nq; }      // (more: a tile in a large lineage's neighbourhood, every dword in order)
    }
    if (usual) {
        if (mx <= 66u) d8_cells<0, true>(c, lane, cand_min, ad_max, sumI, sumMg, occ_cnt, thr0, imax);
        else if (mx < 255u) d8_cells<1, true>(c, lane, cand_min, ad_max, sumI, sumMg, occ_cnt, thr0, imax);
        else d8_cells<2, true>(c, lane, cand_min, ad_max, sumI, sumMg, occ_cnt, thr0, imax);
    } else {
        if (mx <= 66u) d8_cells<0, false>(c, lane, cand_min, ad_max, sumI, sumMg, occ_cnt, thr0, imax);
        else if (mx < 255u) d8_cells<1, false>(c, lane, cand_min, ad_max, sumI, sumMg, occ_cnt, thr0, imax);
        else d8_cells<2, false>(c, lane, cand_min, ad_max, sumI, sumMg, occ_cnt, thr0, imax);
    }
    {
        int tot = occ_cnt;
        if (a.build_occ) { if (lane == 0) a.occ_n[region] = (unsigned int)occ_cnt; }
        if (a.drift_sparse) {      // the region's largest count; a region whose list is incomplete goes on the launch's list of such regions
            for (int o = 32; o > 0; o >>= 1) imax = max(imax, __shfl_down(imax, o));
            if (lane == 0) {
                a.d8s_regmax[region] = imax;
                if (occ_cnt > VGX_OCC_CAP) {
                    const unsigned long long slot = atomicAdd(&a.d8s_bc[(int64_t)rep * 8 + 4], 1ull);
                    a.d8s_ovf[(int64_t)rep * P * a.occ_nreg + (int64_t)slot] = pn * a.occ_nreg + (int)(region - ((int64_t)rep * P + pn) * a.occ_nreg);
                }
            }
        }
        else for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
        if (lane == 0 && tot != 0) atomicAdd(&a.occ_pop[(int64_t)rep * P + pn], (unsigned long long)tot);
    }
    // the block's sums of Ih and of c1 T + c2 TW (wavefronts in a fixed order: reproducible), from which the susceptible drift
    {
        double vI = (double)sumI, vM = sumMg;
        for (int o = 32; o > 0; o >>= 1) { vI += __shfl_down(vI, o); vM += __shfl_down(vM, o); }
        if (lane == 0) { s_sum[wv][0] = vI; s_sum[wv][1] = vM; }
    }
    if (ad_max >= 1e-8 && 1.0 / ad_max < cand_min) cand_min = 1.0 / ad_max;
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_down(cand_min, o);
        if (other < cand_min) cand_min = other;
    }
    if (lane == 0) atomic_min_pos_double(&smin, cand_min);
    __syncthreads();
    if (threadIdx.x < S) {
        // susceptible drift of (pn, sn) from this block's compartments: -base_sn (F Ih + mg) summed, plus the recoveries and
        // samplings into the class's group (pyx:2384-2394)
        double tI = 0.0, tM = 0.0;
        for (int w = 0; w < D8_TB / 64; ++w) { tI += s_sum[w][0]; tM += s_sum[w][1]; }
        const int sn = threadIdx.x;
        const double mgsum = tM - kmig * tI;
        double v = -l_base[sn] * (F * tI + mgsum);
        if (sn == st0) v += (cd0 + cs0) * tI;
        // (sparse: the sum over ALL compartments of the tile in mgsum is not formed — the turns left out — and the tile's hosts go to
        // vgx_tau_drift8s_sus_kernel instead, which has it from the tiles' integer sums)
        if (!a.drift_sparse) a.dS_part[(((int64_t)rep * P + pn) * a.ds_nb + tl) * S + sn] = v;
        else if (sn == 0) a.tI_pt[((int64_t)rep * P + pn) * nt + tl] = (unsigned long long)tI;
    }
    if (threadIdx.x == 0) atomicMin(&a.tau_bits[rep], smin);
    if (do_hist)
        for (int i = threadIdx.x; i < VGX_HIST_X; i += D8_TB) {
            unsigned int v = 0;
            for (int k = 0; k < 16; ++k) v += hist[i * 16 + k];
            if (v) atomicAdd(&a.hist[((int64_t)rep * P + pn) * VGX_HIST_X + i], v);
        }
}

// ---- the drift pass on sparse states (VgxTauArgs.drift_sparse) -----------------------------------------------------------------------
// At natural occupancy (0.65 % of config 4's compartments after SURVEY 8(d)'s warm-up) vgx_tau_drift8_kernel spends its time finding
// out that compartments are empty.  ChooseTau's minimum (pyx:2432-2450) needs
//   (1) every OCCUPIED compartment's candidate: vgx_tau_drift8_kernel still stages each tile's bytes, but gathers the dwords that hold a
//       host (d8_scan) and forms the drifts of those only (their neighbours inside the tile from LDS as before);
//   (2) of the EMPTY compartments, whose candidate is 1 / (their arrivals' rate), only those that can undercut the smallest candidate m
//       of (1): an empty compartment's drift is rate * (its neighbours' hosts) + Bsum (c1 T + c2 TW) <= 3 sites rate max-neighbour +
//       coefficient * column sum, so one whose neighbours all hold fewer than 0.49 / (m rate 3 sites) hosts and whose column's term is
//       below 0.49 / m cannot: the empty neighbours of the larger compartments (vgx_tau_drift8s_heavy_kernel over the lists,
//       vgx_tau_drift8s_ovf_kernel for the regions whose lists are incomplete) and the empty compartments of the columns with the larger
//       sums (vgx_tau_drift8s_col_kernel) are formed exactly, the rest is provably above m.
// Every drift that is formed is formed as the dense pass forms it (same coefficients: d8_coeffs, same fused operations), so the minimum
// is the same bit pattern; the susceptible compartments' drift (in the dense form a sum over ALL compartments of a tile) comes from exact
// integer sums per (population, tile) and differs from the dense form's by rounding only
// (tests/test_hip_tau.py::test_drift_over_the_lists_equals_the_dense_pass).  Where the empty neighbours of too many compartments have
// to be formed (a high mutation rate) the host goes back to the dense pass (vgx_api.hip: sparse_ban).
struct D8S {
    const uint8_t *I8row;
    const int32_t *Irow;
    const double *cT, *cTW;
    int sites;
    bool use_col, use_tw, one_rate;
    double kI, rate_lo, rate_hi, Bsum, c1, c2;
};
static __device__ __forceinline__ int d8s_count(const D8S &c, int h) {
    const int b = c.I8row[h];
    return b < 255 ? b : c.Irow[h];
}
#define VGX_D8S_MAX_SITES 12
static __device__ __forceinline__ double d8s_drift(const D8S &c, int h, int Icell) {
    // the neighbours' bytes: every load of the compartment in flight before the first is used (the sites are wave-uniform)
    int b[3 * VGX_D8S_MAX_SITES];
#pragma unroll
    for (int g = 0; g < VGX_D8S_MAX_SITES; ++g) {
        if (g < c.sites) {
#pragma unroll
            for (int x = 1; x < 4; ++x) b[3 * g + x - 1] = c.I8row[h ^ (x << (2 * g))];
        } else {
            b[3 * g] = 0; b[3 * g + 1] = 0; b[3 * g + 2] = 0;
        }
    }
    int nlo = 0, nhi = 0, sat = 0;
#pragma unroll
    for (int g = 0; g < VGX_D8S_MAX_SITES; ++g) {
        const int v3 = b[3 * g] + b[3 * g + 1] + b[3 * g + 2];
        sat |= (b[3 * g] == 255) | (b[3 * g + 1] == 255) | (b[3 * g + 2] == 255);
        if (g < VGX_DRIFT_LOW) nlo += v3; else nhi += v3;
    }
    if (sat) {      // a byte of 255 stands for "255 or more": the sums again from the 4-byte counts
        nlo = 0; nhi = 0;
        for (int g = 0; g < c.sites; ++g) {
            const int v3 = d8s_count(c, h ^ (1 << (2 * g))) + d8s_count(c, h ^ (2 << (2 * g))) + d8s_count(c, h ^ (3 << (2 * g)));
            if (g < VGX_DRIFT_LOW) nlo += v3; else nhi += v3;
        }
    }
    double drift = c.kI * (double)Icell;
    if (c.one_rate) drift = __builtin_fma(c.rate_lo, (double)(nlo + nhi), drift);
    else drift = __builtin_fma(c.rate_hi, (double)nhi, __builtin_fma(c.rate_lo, (double)nlo, drift));
    if (c.use_col) {
        const double T = c.cT[h];
        const double mg = c.use_tw ? __builtin_fma(c.c2, c.cTW[h], c.c1 * T) : c.c1 * T;
        drift = __builtin_fma(c.Bsum, mg, drift);
    }
    return drift;
}
static __device__ __forceinline__ void d8s_ctx(const VgxTauArgs &a, int rep, int pn, D8S &c) {
    const VgxDevParams &p = a.p;
    const int64_t rowoff = ((int64_t)rep * p.P + pn) * p.H;
    const double *pk = a.d8s_pk + ((int64_t)rep * p.P + pn) * 8;
    c.I8row = a.I8 + rowoff; c.Irow = a.I + rowoff;
    c.cT = a.colT + (int64_t)rep * p.H; c.cTW = a.colTW + (int64_t)rep * p.H;
    c.sites = p.sites;
    c.use_col = a.has_mig && a.mig_uniform;
    c.use_tw = pk[7] != 0.0;
    c.rate_lo = a.mutp[p.sites - VGX_DRIFT_LOW][0]; c.rate_hi = a.mutHi_rate;
    c.one_rate = c.rate_lo == c.rate_hi;
    c.Bsum = pk[0]; c.c1 = pk[1]; c.c2 = pk[2]; c.kI = pk[3];
}
// a candidate of ChooseTau (pyx:2440-2444) into the lane's running values, as vgx_tau_drift8_kernel keeps them
static __device__ __forceinline__ void d8s_candidate(int Icell, double drift, double &cand_min, double &ad_max) {
    const double ad = fabs(drift);
    const bool large = Icell > 66;
    ad_max = fmax(ad_max, large ? 0.0 : ad);
    if (large && ad >= 1e-8) cand_min = fmin(cand_min, ((double)(0.03f * (float)Icell) / 2.0) / ad);
}
static __device__ __forceinline__ void d8s_commit(const VgxTauArgs &a, int rep, double cand_min, double ad_max) {
    if (ad_max >= 1e-8 && 1.0 / ad_max < cand_min) cand_min = 1.0 / ad_max;
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_down(cand_min, o);
        if (other < cand_min) cand_min = other;
    }
    // (a look before the atomic: tens of thousands of wavefronts on one address, and all but a few have nothing smaller to offer)
    if ((threadIdx.x & 63) == 0 && cand_min < __longlong_as_double((long long)*(volatile unsigned long long *)&a.tau_bits[rep]))
        atomic_min_pos_double(&a.tau_bits[rep], cand_min);
}
// what an empty compartment's arrivals must reach to matter: 1 / (the smallest candidate so far)
static __device__ __forceinline__ double d8s_need(const VgxTauArgs &a, int rep) { return 1.0 / __longlong_as_double((long long)a.tau_bits[rep]); }

// grid of the passes over the regions = (8 * ceil(P / 8) * ceil(occ_nreg / 4), R), flattened as vgx_tau_drift8_kernel's: the blocks of one
// population on one XCD (workgroups go to the XCDs round-robin), so that a row's bytes are fetched into ONE L2
static __device__ __forceinline__ void d8s_block(const VgxTauArgs &a, int &pn, int &bx) {
    const unsigned nb = (unsigned)((a.occ_nreg + 3) / 4), f = blockIdx.x, sq = f >> 3;
    pn = (int)(sq / nb) * 8 + (int)(f & 7u);
    bx = (int)(sq % nb);
}
// the coefficients of every population (block = D8_TB threads as vgx_tau_drift8_kernel: tau_migu_setup's sum runs in the block's order).
// grid = (P, R); d8s_bc zeroed before.
extern "C" __global__ void __launch_bounds__(D8_TB) vgx_tau_drift8s_prep_kernel(VgxTauArgs a) {
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, pn = blockIdx.x, rep = blockIdx.y;
    if (!a.active[rep]) return;
    __shared__ double l_base[64];
    __shared__ double s_wu[16];
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    for (int i = threadIdx.x; i < S; i += D8_TB) l_base[i] = p.cb_b[0] * p.cb_sigma[i] * (double)Sus[i];
    __syncthreads();
    const bool use_col = a.has_mig && a.mig_uniform;
    MigU mu = {0.0, 0.0, 0.0, true};
    if (use_col) mu = tau_migu_setup(a, rep, pn, s_wu);
    double Bsum = 0.0;
    for (int sn = 0; sn < S; ++sn) Bsum += l_base[sn];
    const double F = a.F[(int64_t)rep * P + pn];
    const double cd0 = p.c_d[0], cs0 = p.c_s[0] * p.sampMult[pn], ctm0 = p.c_tm[0];
    const bool use_tw = use_col && !mu.weq;
    double c1, c2, kmig, kI;
    d8_coeffs(mu, use_tw, Bsum, F, cd0 + cs0 + ctm0, c1, c2, kmig, kI);
    if (threadIdx.x == 0) {
        double *pk = a.d8s_pk + ((int64_t)rep * P + pn) * 8;
        pk[0] = Bsum; pk[1] = c1; pk[2] = c2; pk[3] = kI; pk[4] = F; pk[5] = mu.wt; pk[6] = kmig; pk[7] = use_tw ? 1.0 : 0.0;
        // upper bounds of |Bsum|, |c1|, |c2| over the populations (non-negative doubles order as their bit patterns)
        atomicMax(&a.d8s_bc[(int64_t)rep * 8 + 0], (unsigned long long)__double_as_longlong(fabs(Bsum)));
        atomicMax(&a.d8s_bc[(int64_t)rep * 8 + 1], (unsigned long long)__double_as_longlong(fabs(c1)));
        atomicMax(&a.d8s_bc[(int64_t)rep * 8 + 2], (unsigned long long)__double_as_longlong(fabs(c2)));
    }
}

// (2a') the same for the regions whose lists are incomplete (vgx_tau_drift8_kernel puts them on the launch's list of such regions): a unit
// = the 256 compartments one wavefront of that kernel takes in a turn, four units of one region per block and turn of the grid-stride loop.
// grid = (VGX_D8S_OVF_BLOCKS, R), block = 256.
#define VGX_D8S_OVF_BLOCKS 2048
extern "C" __global__ void __launch_bounds__(256) vgx_tau_drift8s_ovf_kernel(VgxTauArgs a) {
    const VgxDevParams &p = a.p;
    const int P = p.P, rep = blockIdx.y;
    if (!a.active[rep]) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int novf = (int)a.d8s_bc[(int64_t)rep * 8 + 4];
    if (novf == 0) return;
    const int low = p.sites < VGX_D8_LOW ? p.sites : VGX_D8_LOW, TSd = 1 << (2 * low - 2), turns = TSd / D8_TB;   // (a multiple of four from seven sites on)
    const double need = d8s_need(a, rep);
    for (int g = blockIdx.x; g < novf * (turns / 4); g += gridDim.x) {
        const int ridx = a.d8s_ovf[(int64_t)rep * P * a.occ_nreg + g / (turns / 4)];
        const int pn = ridx / a.occ_nreg, reg = ridx % a.occ_nreg, k = (g % (turns / 4)) * 4 + w;
        const int64_t region = ((int64_t)rep * P + pn) * a.occ_nreg + reg;
        D8S c;
        d8s_ctx(a, rep, pn, c);
        const double rate = fmax(fabs(c.rate_lo), fabs(c.rate_hi));
        const double xpush = 0.49 * need / (rate * 3.0 * (double)p.sites);
        if (!(rate > 0.0) || !((double)a.d8s_regmax[region] >= xpush)) continue;
        const int tl = reg / VGX_D8_WAVES, wv = reg % VGX_D8_WAVES;
        const int h0 = (tl << (2 * low)) + 4 * (wv * 64 + lane + D8_TB * k);
        const uint32_t own = *(const uint32_t *)(c.I8row + h0);
        double cand_min = 1.0, ad_max = 0.0;
        int nheavy = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (((own >> (8 * j)) & 255u) == 0u) continue;
            const int h = h0 + j;
            if (!((double)d8s_count(c, h) >= xpush)) continue;
            nheavy += 1;
            for (int q = 0; q < 3 * p.sites; ++q) {
                const int nb = h ^ ((q % 3 + 1) << (2 * (q / 3)));
                if (c.I8row[nb] == 0) d8s_candidate(0, d8s_drift(c, nb, 0), cand_min, ad_max);
            }
        }
        d8s_commit(a, rep, cand_min, ad_max);
        for (int o = 32; o > 0; o >>= 1) nheavy += __shfl_down(nheavy, o);
        if (lane == 0 && nheavy != 0) atomicAdd(&a.d8s_bc[(int64_t)rep * 8 + 3], (unsigned long long)nheavy);
    }
}

// (2a) the empty neighbours of the larger compartments of the regions with complete lists: those whose largest count (kept by pass 1)
// reaches the bound.  Same grid as pass 1.
extern "C" __global__ void __launch_bounds__(256) vgx_tau_drift8s_heavy_kernel(VgxTauArgs a) {
    const VgxDevParams &p = a.p;
    const int rep = blockIdx.y;
    int pn, bx;
    d8s_block(a, pn, bx);
    if (pn >= p.P || !a.active[rep]) return;
    const int lane = threadIdx.x & 63, reg = bx * 4 + (threadIdx.x >> 6);
    if (reg >= a.occ_nreg) return;
    const int64_t region = ((int64_t)rep * p.P + pn) * a.occ_nreg + reg;
    const int n = (int)a.occ_n[region];
    if (n > VGX_OCC_CAP) return;
    const double rate = fmax(fabs(a.mutp[p.sites - VGX_DRIFT_LOW][0]), fabs(a.mutHi_rate));
    if (!(rate > 0.0)) return;
    const double xpush = 0.49 * d8s_need(a, rep) / (rate * 3.0 * (double)p.sites);
    if (!((double)a.d8s_regmax[region] >= xpush)) return;
    D8S c;
    d8s_ctx(a, rep, pn, c);
    double cand_min = 1.0, ad_max = 0.0;
    int nheavy = 0;
    const int32_t *src = a.occ + region * VGX_OCC_CAP;
    const int nnb = 3 * p.sites;
    for (int i0 = 0; i0 < n; i0 += 64) {
        // the turn's large compartments, then one (compartment, neighbour) pair per lane
        const int h = i0 + lane < n ? src[i0 + lane] : 0;
        const bool big = i0 + lane < n && (double)d8s_count(c, h) >= xpush;
        unsigned long long m = __ballot(big);
        nheavy += (int)__popcll(m);
        while (m != 0ull) {
            const int l = (int)__builtin_ctzll(m);
            m &= m - 1ull;
            const int hh = __shfl(h, l);
            if (lane < nnb) {
                const int nb = hh ^ ((lane % 3 + 1) << (2 * (lane / 3)));
                if (c.I8row[nb] == 0) d8s_candidate(0, d8s_drift(c, nb, 0), cand_min, ad_max);
            }
        }
    }
    d8s_commit(a, rep, cand_min, ad_max);
    // how many compartments took this path: the host goes back to the dense pass where it is no longer the few (vgx_api.hip)
    if (lane == 0 && nheavy != 0) atomicAdd(&a.d8s_bc[(int64_t)rep * 8 + 3], (unsigned long long)nheavy);
}

// (2b) the empty compartments of the columns with the larger sums.  grid = (H / 256, R), block = 256: one haplotype per thread.
extern "C" __global__ void __launch_bounds__(256) vgx_tau_drift8s_col_kernel(VgxTauArgs a) {
    const VgxDevParams &p = a.p;
    const int P = p.P, H = p.H, rep = blockIdx.y;
    if (!a.active[rep]) return;
    const int h = blockIdx.x * 256 + threadIdx.x;
    const double Bm = __longlong_as_double((long long)a.d8s_bc[(int64_t)rep * 8 + 0]), c1m = __longlong_as_double((long long)a.d8s_bc[(int64_t)rep * 8 + 1]),
                 c2m = __longlong_as_double((long long)a.d8s_bc[(int64_t)rep * 8 + 2]);
    const double need = 0.49 * d8s_need(a, rep);
    double cand_min = 1.0, ad_max = 0.0;
    if (h < H) {
        const double T = a.colT[(int64_t)rep * H + h];
        // (the weighted sum is read only where the weights differ: d8s_pk[.][7] of every population then; its bound by the largest weight
        // is not kept, so the column's own TW stands for it)
        const bool tw = a.d8s_pk[((int64_t)rep * P) * 8 + 7] != 0.0;
        const double TW = tw ? fabs(a.colTW[(int64_t)rep * H + h]) : 0.0;
        if (T > 0.0 && Bm * (c1m * T + c2m * TW) >= need) {
            for (int q = 0; q < P; ++q) {
                if (a.I8[((int64_t)rep * P + q) * H + h] != 0) continue;
                D8S c;
                d8s_ctx(a, rep, q, c);
                d8s_candidate(0, d8s_drift(c, h, 0), cand_min, ad_max);
            }
        }
    }
    d8s_commit(a, rep, cand_min, ad_max);
}

// per tile of the rows: the hosts of all populations and their weighted sum (the populations in a fixed order).  grid = (nt8, R), block = 256.
extern "C" __global__ void __launch_bounds__(256) vgx_tau_drift8s_tiles_kernel(VgxTauArgs a) {
    const int P = a.p.P, t = blockIdx.x, rep = blockIdx.y, nt = a.nt8;
    if (!a.active[rep]) return;
    __shared__ double sT[4], sW[4];
    double T = 0.0, TW = 0.0;
    for (int q = threadIdx.x; q < P; q += 256) {
        const double v = (double)a.tI_pt[((int64_t)rep * P + q) * nt + t];
        T += v;
        TW += a.d8s_pk[((int64_t)rep * P + q) * 8 + 5] * v;
    }
    for (int o = 32; o > 0; o >>= 1) { T += __shfl_down(T, o); TW += __shfl_down(TW, o); }
    if ((threadIdx.x & 63) == 0) { sT[threadIdx.x >> 6] = T; sW[threadIdx.x >> 6] = TW; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.d8s_tile[(int64_t)rep * 2 * nt + t] = (sT[0] + sT[1]) + (sT[2] + sT[3]);
        a.d8s_tile[(int64_t)rep * 2 * nt + nt + t] = (sW[0] + sW[1]) + (sW[2] + sW[3]);
    }
}

// the susceptible compartments' drift of (population, tile) from the integer sums, where vgx_tau_drift8_kernel's blocks leave theirs.
// grid = (P, R), block = 64.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_drift8s_sus_kernel(VgxTauArgs a) {
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, pn = blockIdx.x, rep = blockIdx.y, nt = a.nt8;
    if (!a.active[rep]) return;
    const double *s_T = a.d8s_tile + (int64_t)rep * 2 * nt, *s_TW = s_T + nt;     // per tile: hosts of all populations, and weighted
    const double *pk = a.d8s_pk + ((int64_t)rep * P + pn) * 8;
    const double Bsum = pk[0], c1 = pk[1], c2 = pk[2], F = pk[4], kmig = pk[6];
    const bool use_tw = pk[7] != 0.0;
    (void)Bsum;
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    const double cd0 = p.c_d[0], cs0 = p.c_s[0] * p.sampMult[pn];
    const int st0 = p.c_stype[0];
    for (int i = threadIdx.x; i < nt * S; i += 64) {
        const int t = i / S, sn = i % S;
        const double tI = (double)a.tI_pt[((int64_t)rep * P + pn) * nt + t];
        const double tM = use_tw ? __builtin_fma(c2, s_TW[t], c1 * s_T[t]) : c1 * s_T[t];
        const double mgsum = tM - kmig * tI;
        const double base = p.cb_b[0] * p.cb_sigma[sn] * (double)Sus[sn];
        double v = -base * (F * tI + mgsum);
        if (sn == st0) v += (cd0 + cs0) * tI;
        a.dS_part[(((int64_t)rep * P + pn) * a.ds_nb + t) * S + sn] = v;
    }
}

// Susceptible compartments: immunity-transition drift (pyx:2374-2381), tau candidates (pyx:2445-2450),
// final tau_l; clears the per-step accumulators.  grid = R, block = 64.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_choose_kernel(VgxTauArgs a) {
    const int rep = blockIdx.x;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S;
    const int64_t *Sus = a.S + (int64_t)rep * P * S;
    double best = 1.0;
    for (int idx = threadIdx.x; idx < P * S; idx += 64) {
        int pn = idx / S, sn = idx % S;
        double d = 0.0;   // the blocks' parts in block order
        for (int b = 0; b < a.ds_nb; ++b) d += a.dS_part[(((int64_t)rep * P + pn) * a.ds_nb + b) * S + sn];
        for (int o = 0; o < S; ++o) {
            if (o == sn) continue;
            d += p.suscepTransition[o * S + sn] * (double)Sus[pn * S + o];
            d -= p.suscepTransition[sn * S + o] * (double)Sus[pn * S + sn];
        }
        if (fabs(d) >= 1e-8) {
            float eps = 0.03f;
            double v = (double)(eps * (float)Sus[idx]) / 2.0;
            double cand = (v > 1.0 ? v : 1.0) / fabs(d);
            if (cand < best) best = cand;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_down(best, o);
        if (other < best) best = other;
    }
    if (threadIdx.x == 0) {
        double t = __longlong_as_double((long long)a.tau_bits[rep]);
        a.tau[rep] = t < best ? t : best;
        a.retry[rep] = 0;
        a.accepted[rep] = 0;
    }
}


// ---- sieve of the halving loop -----------------------------------------------------------------------------------------
// The reference redraws the whole step with tau/2 whenever one compartment fails the bounds check (pyx:2316-2321).  With
// many sparsely filled compartments the first tries are certain to fail: at config 4 (2.7e8 compartments of 3 hosts) the
// expected number of failing compartments is 1.6e6 at the chosen tau and still 500 three halvings later.  A try's random
// streams are keyed by the try index (Philox counter word `retry`), so leaving out a try that would have been rejected
// changes nothing at all in what is finally accepted.  This kernel computes, for tau * 2^-k (k < VGX_SIEVE_K), a LOWER
// bound E_k on the expected number of failing compartments; vgx_tau_sieve_pick_kernel then starts the loop at the first k
// with E_k <= VGX_SIEVE_MIN_FAILS (treating compartments as independent, a skipped try would have been accepted with
// probability < exp(-VGX_SIEVE_MIN_FAILS)).
//
// Bound for compartment c with X hosts (all channels are independent Poisson variables):
//   fails if   no transmission and no out-migration of c (both are booked as + on c by the check, pyx:2473, 2517),
//              no mutant arrives, and recoveries + samplings >= X + 1.
//   P(no transmission, no out-migration) = exp(-(r_tr + r_mig) X tau)
//   P(no mutant arrives) >= exp(-(|drift_c| + r_dec X) tau) >= exp(-m_c - r_dec X tau),  m_c = max(eps X / 2, 1):
//              transmission + incoming mutation <= drift_c + decrements, and ChooseTau made |drift_c| tau <= m_c
//   P(Poisson(mu) >= n) >= exp(-mu) mu^n / n!,   mu = (r_rec + r_samp) X tau,  n = X + 1
// Evaluated in single precision (the threshold has orders of magnitude of slack).  grid = (gx, P, R).
#define VGX_SIEVE_MIN_FAILS 32.0
#define VGX_SIEVE_XMAX 64
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_sieve_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, C = p.C, H = p.H;
    if (C > 256) return;   // no table space: E stays 0 and nothing is skipped
    __shared__ float s_all[256], s_dec[256], s_lf[VGX_SIEVE_XMAX + 2];
    __shared__ double s_E[VGX_SIEVE_K];
    const double F = a.F[(int64_t)rep * P + pn];
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    for (int i = threadIdx.x; i < C; i += TB) {
        const int cb = p.c_bidx[i];
        double rtr = 0.0;
        for (int sn = 0; sn < S; ++sn) rtr += p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F;
        const double rmig = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * p.CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
        const double dec = p.c_d[i] + p.c_s[i] * p.sampMult[pn];
        s_dec[i] = (float)dec;
        s_all[i] = (float)((rmig + dec + (a.mut_uniform ? a.mut_total : p.c_tm[i]) + rtr) * 1.000001);   // rounded up
    }
    for (int i = threadIdx.x; i < VGX_SIEVE_XMAX + 2; i += TB) s_lf[i] = (float)(lgamma((double)i + 1.0) * 1.000001);
    if (threadIdx.x < VGX_SIEVE_K) s_E[threadIdx.x] = 0.0;
    __syncthreads();
    const float tau0 = (float)a.tau[rep];
    float acc[VGX_SIEVE_K];
#pragma unroll
    for (int k = 0; k < VGX_SIEVE_K; ++k) acc[k] = 0.f;
    const int32_t *Irow = a.I + ((int64_t)rep * P + pn) * H;
    for (int h0 = (blockIdx.x * TB + threadIdx.x) * 4; h0 < H; h0 += gridDim.x * TB * 4) {
        int Iv[4] = {0, 0, 0, 0};
        if (h0 + 3 < H && (H & 3) == 0) { const int4 x = *(const int4 *)(Irow + h0); Iv[0] = x.x; Iv[1] = x.y; Iv[2] = x.z; Iv[3] = x.w; }
        else for (int j = 0; j < 4; ++j) if (h0 + j < H) Iv[j] = Irow[h0 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int X = Iv[j];
            if (X <= 0 || X > VGX_SIEVE_XMAX) continue;
            const int c = (C == 1) ? 0 : p.cls[h0 + j];
            const float Xf = (float)X, n = Xf + 1.0f;
            const float mu0 = s_dec[c] * Xf * tau0;
            if (!(mu0 > 0.f)) continue;
            const float m = fmaxf(0.0151f * Xf, 1.0f);
            const float A = n * (__logf(mu0) - 1e-6f) - s_lf[X + 1];
            const float B = (s_all[c] * Xf * tau0 + mu0) + m;      // (the arrivals' bound |drift| tau <= m halves with tau like the rest)
            const float nl2 = n * 0.69314724f;   // ln 2 rounded up
            float sc = 1.0f, Ak = A;
#pragma unroll
            for (int k = 0; k < VGX_SIEVE_K; ++k) {
                acc[k] += __expf(Ak - B * sc);
                Ak -= nl2;
                sc *= 0.5f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VGX_SIEVE_K; ++k) {
        float v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0 && v > 0.f) atomicAdd(&s_E[k], (double)v);
    }
    __syncthreads();
    if (threadIdx.x < VGX_SIEVE_K && s_E[threadIdx.x] > 0.0) atomicAdd(&a.sieve[(int64_t)rep * VGX_SIEVE_K + threadIdx.x], s_E[threadIdx.x] * 0.99);
}

// The same bound from the histogram of compartment sizes vgx_tau_drift_fast_kernel filled (the term of a compartment depends
// on its size, class and population only): no pass over the compartments.  Double precision; the histogram is cleared for
// the next step.  grid = (P, R), block = 64: lane <-> compartment size 1..64.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_sieve_hist_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y, pn = blockIdx.x;
    if (!a.active[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, C = p.C;
    const double F = a.F[(int64_t)rep * P + pn];
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    const double tau0 = a.tau[rep];
    const int X = threadIdx.x + 1;
    const double Xd = (double)X, n = Xd + 1.0;
    double acc[VGX_SIEVE_K];
#pragma unroll
    for (int k = 0; k < VGX_SIEVE_K; ++k) acc[k] = 0.0;
    for (int c = 0; c < C; ++c) {
        unsigned int *hp = a.hist + ((int64_t)rep * P + pn) * C * VGX_HIST_X + c * VGX_HIST_X + threadIdx.x;
        const unsigned int cnt = *hp;
        *hp = 0;
        if (cnt == 0) continue;
        const int cb = p.c_bidx[c];
        double rtr = 0.0;
        for (int sn = 0; sn < S; ++sn) rtr += p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F;
        const double rmig = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * p.CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
        const double dec = p.c_d[c] + p.c_s[c] * p.sampMult[pn];
        const double all = (rmig + dec + (a.mut_uniform ? a.mut_total : p.c_tm[c]) + rtr) * 1.000001;   // rounded up
        const double mu0 = dec * Xd * tau0;
        if (!(mu0 > 0.0)) continue;
        const double m = fmax(0.0151 * Xd, 1.0);
        double Ak = n * log(mu0) - lgamma(n + 1.0);
        const double B = all * Xd * tau0 + mu0 + m;     // (the arrivals' bound |drift| tau <= m halves with tau like the rest)
        const double nl2 = n * 0.6931471805599454;   // ln 2 rounded up
        double sc = 1.0;
#pragma unroll
        for (int k = 0; k < VGX_SIEVE_K; ++k) {
            acc[k] += (double)cnt * exp(Ak - B * sc);
            Ak -= nl2;
            sc *= 0.5;
        }
    }
#pragma unroll
    for (int k = 0; k < VGX_SIEVE_K; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        // the population's part: summed over the populations by vgx_tau_sieve_pick_kernel in a fixed order (256 blocks adding to the
        // same 16 doubles were worked off one after the other: 40 us per step, and the sum depended on the order)
        if (threadIdx.x == 0) a.sieve_pop[((int64_t)rep * P + pn) * VGX_SIEVE_K + k] = v * 0.99;
    }
}

// Starts the halving loop of the step at the first try that is not certain to fail.  grid = R, one thread.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_sieve_pick_kernel(VgxTauArgs a) {
    const int rep = blockIdx.x;
    if (!a.active[rep]) return;
    double *E = a.sieve + (int64_t)rep * VGX_SIEVE_K;
    if (a.hist) {   // the populations' parts (vgx_tau_sieve_hist_kernel): lane-strided partial sums, then a tree — one order, always
        const int P = a.p.P;
        double acc[VGX_SIEVE_K];
#pragma unroll
        for (int k = 0; k < VGX_SIEVE_K; ++k) acc[k] = 0.0;
        for (int pn = threadIdx.x; pn < P; pn += 64) {      // (a population's sixteen values: two cache lines, the loads independent)
            const double *sp = a.sieve_pop + ((int64_t)rep * P + pn) * VGX_SIEVE_K;
#pragma unroll
            for (int k = 0; k < VGX_SIEVE_K; ++k) acc[k] += sp[k];
        }
#pragma unroll
        for (int k = 0; k < VGX_SIEVE_K; ++k) {
            double v = acc[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (threadIdx.x == 0) E[k] += v;
        }
    }
    if (threadIdx.x != 0) return;
    int k0 = 0;
    while (k0 < VGX_SIEVE_K && E[k0] > VGX_SIEVE_MIN_FAILS) k0 += 1;
    for (int k = 0; k < VGX_SIEVE_K; ++k) E[k] = 0.0;
    if (k0 > 0) {
        double t = a.tau[rep];
        for (int k = 0; k < k0; ++k) t *= 0.5;   // the halvings the skipped tries would have made (pyx:2319)
        a.tau[rep] = t;
        a.retry[rep] = k0;
        a.sieve_skipped[rep] += k0;
    }
}

// A multievent row (events.pxi:116-125) of the step being drawn; rows of a rejected retry are discarded by
// rewinding mev_n (vgx_tau_decide_kernel).
static __device__ __forceinline__ void tau_row(const VgxTauArgs &a, int rep, int64_t num, int type, int hap, int pop,
                                               int nh, int np) {
    if (a.mev_cap <= 0) return;
    unsigned long long slot = atomicAdd(&a.mev_n[rep], 1ull);
    if ((int64_t)slot < a.mev_cap) {
        int64_t *row = a.mev + ((int64_t)rep * a.mev_cap + (int64_t)slot) * 6;
        row[0] = num; row[1] = type; row[2] = hap; row[3] = pop; row[4] = nh; row[5] = np;
    }
}

// ---- the list of individual moves of a try ---------------------------------------------------------------------------------
// entry = compartment (bits 0-37) | magnitude (bits 38-60) | sign (bit 61) | applied-only (bit 62)
//   a mutant entering a compartment:  +k, booked in both sets of deltas (pyx:2473 and pyx:2548)
//   a migrant entering a compartment: +k, applied only (pyx:2548; the reference's check books it on its SOURCE, pyx:2473)
//   sparse mode only, a compartment's own net change: signed, applied only (its check is made where it is drawn)
// The list is sharded by thread block (VGX_INC_SHARDS counters) so that appends do not serialise on one address; a
// wavefront stages its entries in LDS and reserves room for them with one atomic.
#define VGX_INC_CELL_BITS 38
#define VGX_INC_MAXMULT ((int64_t)((1 << 23) - 1))
#define VGX_INC_NEG ((int64_t)1 << 61)
#define VGX_INC_APPLIED ((int64_t)1 << 62)
static __device__ __forceinline__ int64_t tau_entry_cell(int64_t e) { return e & (((int64_t)1 << VGX_INC_CELL_BITS) - 1); }
static __device__ __forceinline__ int tau_entry_delta(int64_t e) {
    const int k = (int)((e >> VGX_INC_CELL_BITS) & VGX_INC_MAXMULT);
    return (e & VGX_INC_NEG) ? -k : k;
}
#define VGX_WSTAGE 192      // entries a wavefront stages before one reservation in the global list
struct WaveStage { int n; int pad; int64_t e[VGX_WSTAGE]; };

static __device__ __forceinline__ void tau_list_global(const VgxTauArgs &a, int rep, int64_t entry) {
    const int shard = (int)((blockIdx.x + gridDim.x * blockIdx.y) & (a.inc_shards - 1));
    const int64_t scap = a.inc_cap / a.inc_shards;
    unsigned long long slot = atomicAdd(&a.inc_n[(int64_t)rep * VGX_INC_SHARDS + shard], 1ull);
    if ((int64_t)slot < scap) a.inc[(int64_t)rep * a.inc_cap + (int64_t)shard * scap + (int64_t)slot] = entry;   // a full shard is detected later
}
// st == nullptr: straight to the global list
static __device__ __forceinline__ void tau_list_add(const VgxTauArgs &a, WaveStage *st, int rep, int64_t cell, int64_t delta, bool applied_only) {
    const int64_t flags = (delta < 0 ? VGX_INC_NEG : 0) | (applied_only ? VGX_INC_APPLIED : 0);
    int64_t left = delta < 0 ? -delta : delta;
    while (left > 0) {
        const int64_t k = left < VGX_INC_MAXMULT ? left : VGX_INC_MAXMULT;
        const int64_t entry = cell | (k << VGX_INC_CELL_BITS) | flags;
        int slot = st ? atomicAdd(&st->n, 1) : VGX_WSTAGE;
        if (slot < VGX_WSTAGE) st->e[slot] = entry;
        else tau_list_global(a, rep, entry);
        left -= k;
    }
}
// wave-uniform call: moves the wavefront's stage to the global list
static __device__ __forceinline__ void tau_stage_flush(const VgxTauArgs &a, WaveStage *st, int rep) {
    WSYNC();
    int n = st->n;
    n = __builtin_amdgcn_readfirstlane(n < VGX_WSTAGE ? n : VGX_WSTAGE);
    if (n > 0) {
        const int lane = threadIdx.x & 63;
        const int shard = (int)((blockIdx.x + gridDim.x * blockIdx.y) & (a.inc_shards - 1));
        const int64_t scap = a.inc_cap / a.inc_shards;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&a.inc_n[(int64_t)rep * VGX_INC_SHARDS + shard], (unsigned long long)n);
        base = (unsigned long long)bcast_i64((int64_t)base, 0);
        int64_t *dst = a.inc + (int64_t)rep * a.inc_cap + (int64_t)shard * scap;
        for (int i = lane; i < n; i += 64)
            if ((int64_t)(base + i) < scap) dst[base + i] = st->e[i];
        WSYNC();
    }
    if ((threadIdx.x & 63) == 0) st->n = 0;
    WSYNC();
}

// ---- compartments found below zero on their own ------------------------------------------------------------------------
// Hash table of the sparse mode (open addressing, linear probing).  A key carries the try counter `gen` of the call in its
// upper bits: entries of earlier tries read as empty, nothing is ever cleared.
static __device__ __forceinline__ uint32_t tau_hash(int64_t cell) {
    return (uint32_t)(((uint64_t)cell * 0x9E3779B97F4A7C15ull) >> 32);
}
static __device__ void tau_st_insert(const VgxTauArgs &a, int rep, int64_t cell, int64_t v) {
    unsigned long long *keys = a.st_key + (int64_t)rep * a.st_size;
    const unsigned long long mine = (unsigned long long)cell | ((unsigned long long)a.gen << VGX_INC_CELL_BITS);
    const uint32_t mask = (uint32_t)(a.st_size - 1);
    uint32_t slot = tau_hash(cell) & mask;
    for (int64_t probe = 0; probe < 2 * a.st_size; ++probe) {
        const unsigned long long k = __atomic_load_n(&keys[slot], __ATOMIC_RELAXED);
        if ((k >> VGX_INC_CELL_BITS) != (unsigned long long)a.gen) {   // empty or of an earlier try: claim it
            if (atomicCAS(&keys[slot], k, mine) == k) { a.st_val[(int64_t)rep * a.st_size + slot] = v; return; }
            continue;                                                   // somebody else was faster: look at the slot again
        }
        slot = (slot + 1) & mask;
    }
}
// slot of `cell` in this try's table, or -1
static __device__ __forceinline__ int64_t tau_st_find(const VgxTauArgs &a, int rep, int64_t cell) {
    const unsigned long long *keys = a.st_key + (int64_t)rep * a.st_size;
    const uint32_t mask = (uint32_t)(a.st_size - 1);
    uint32_t slot = tau_hash(cell) & mask;
    for (int64_t probe = 0; probe < a.st_size; ++probe) {
        const unsigned long long k = keys[slot];
        if ((k >> VGX_INC_CELL_BITS) != (unsigned long long)a.gen) return -1;
        if ((int64_t)(k & (((unsigned long long)1 << VGX_INC_CELL_BITS) - 1)) == cell) return (int64_t)slot;
        slot = (slot + 1) & mask;
    }
    return -1;
}

// Bounds check of GenerateEvents_tau (pyx:2522-2528) for a compartment's OWN deltas, v = infectious + delta as booked by the
// reference's check.  Incoming mutants only add to it afterwards, so v > sizes is final (the try is rejected); v < 0 may still
// be rescued: the compartment is listed and looked at again once the arrivals are known.
static __device__ __forceinline__ void tau_own_check(const VgxTauArgs &a, int rep, int pn, int hn, int64_t v) {
    if (v > a.p.sizes[pn]) atomicAnd(&a.ok[rep], 0);
    else if (v < 0) {
        const unsigned long long slot = atomicAdd(&a.suspect_n[rep], 1ull);
        if ((int64_t)slot < a.suspect_cap) {
            const int64_t cell = (int64_t)pn * a.p.H + hn;
            a.suspect[((int64_t)rep * a.suspect_cap + (int64_t)slot) * 2] = cell;
            a.suspect[((int64_t)rep * a.suspect_cap + (int64_t)slot) * 2 + 1] = v;
            if (a.sparse) tau_st_insert(a, rep, cell, v);
        }
    }
}

// Per-block tables of the draw kernel (a block works on ONE population): class parameters, this population's
// transmission / migration weights per birth class and the bisection tables, staged in LDS when they fit
// (else the pointers refer to the global arrays).
// TABS says where they are: 0 = global memory (generic pointers), 1 = LDS except the migration table, 2 = all in LDS.  The LDS
// forms carry address-space-3 pointers: a generic pointer that may be LDS or global compiles to FLAT loads, and every flat
// load waits for ALL outstanding memory operations of the wavefront (the loads issued ahead, the atomics of the round).
#define VGX_AS3 __attribute__((address_space(3)))
template <int TABS> struct TauTabSel { typedef const double *DP; typedef const int32_t *IP; typedef const double *CP; };
template <> struct TauTabSel<1> { typedef const VGX_AS3 double *DP; typedef const VGX_AS3 int32_t *IP; typedef const double *CP; };
template <> struct TauTabSel<2> { typedef const VGX_AS3 double *DP; typedef const VGX_AS3 int32_t *IP; typedef const VGX_AS3 double *CP; };
template <int TABS> struct TauTabT {
    typename TauTabSel<TABS>::DP c_d, c_s, c_tm;      // [C]
    typename TauTabSel<TABS>::IP c_bidx, c_stype;     // [C]
    typename TauTabSel<TABS>::DP rtr;                 // [CB]   transmission rate per infected of the class: sum_sn wtr
    typename TauTabSel<TABS>::DP wtr;                 // [CB][S] b * sigma[sn] * S[pn][sn] * F[pn]
    typename TauTabSel<TABS>::DP rmig;                // [CB]   out-migration rate per infected: Gout * b * m[pn][pn]
    typename TauTabSel<TABS>::DP mutcum;              // [3*sites] running sums of the uniform mutation model
    typename TauTabSel<TABS>::CP cdf;                 // [CB][P*S] running sums of the out-migration channel weights of pn
};

// ---- random numbers of a try -----------------------------------------------------------------------------------------
// Every compartment owns a Philox stream keyed by (seed, attempt) with the counter (compartment, step, try).  Its FIRST
// uniform is built in two parts: the top 8 bits (`bucket`) come from a Philox block shared by the 16 compartments a
// lane of the draw kernel looks at (counter = (group, step, try | 0xFFFFF)), the remaining 52 bits from the compartment's
// own stream and only when they are needed.  A compartment with a small mean draws no event whenever
// (bucket + 1) / 256 <= 1 - lam <= exp(-lam), which the draw kernel tests in single precision with lam rounded up: one
// Philox block and a handful of instructions for 16 compartments, and the law of the draw is exactly that of inversion
// with a 60-bit uniform.
// group of compartment hn of population pn: 16 consecutive haplotypes (what a lane of the scan kernel looks at); the bucket
// of haplotype 16 g + 4 k + j is byte j of word k of the group's block.
// What a block of the events kernel reads from global memory about its (replicate, population) ONCE: inside the round loop the
// compiler cannot keep such values in registers across the loop's stores and atomics, and every reload is a dependent scalar
// load (pointer from the kernel arguments, then the value).
struct TauEnv { uint64_t seed; uint32_t att, step, retry; double sampMult; };
static __device__ __forceinline__ TauEnv tau_env(const VgxTauArgs &a, int rep, int pn) {
    TauEnv e;
    e.seed = (uint64_t)a.seeds[rep]; e.att = (uint32_t)a.attempt[rep]; e.step = (uint32_t)a.step[rep]; e.retry = (uint32_t)a.retry[rep];
    e.sampMult = a.p.sampMult[pn];
    return e;
}
static __device__ __forceinline__ uint32_t tau_bucket(const VgxTauArgs &a, const TauEnv &E, int pn, int hn) {
    const int H = a.p.H;
    const uint64_t groups = (uint64_t)((H + 15) >> 4);
    const uint64_t gidx = (uint64_t)pn * groups + (uint64_t)(hn >> 4);
    const uint32_t key[2] = {(uint32_t)E.seed ^ (E.att * 0x9E3779B9u), (uint32_t)(E.seed >> 32) ^ 0x85EBCA6Bu};
    const uint32_t ctr[4] = {(uint32_t)gidx, (uint32_t)(gidx >> 32), E.step, (E.retry << 20) | 0xFFFFFu};
    uint32_t w[4];
    vgx_philox4x32(ctr, key, w);
    return (w[(hn >> 2) & 3] >> (8 * (hn & 3))) & 255u;
}

// GenerateEvents_tau for one compartment (pn, hn).  All channels out of a compartment are independent Poisson
// variables, so their sum is Poisson with the summed rate and, given the sum, the channel of each event is
// multinomial: ONE draw per compartment, then a split (same joint law as pyx:2464-2520, ~5x fewer draws).  The
// split first only counts the frequent kinds (recovery, sampling, transmission) and the number of mutants and
// migrants; their targets are then chosen by bisection in a second, short loop, so the rare, expensive branches
// are not executed by the whole wavefront on every event.  Books
//   ownChk : the compartment's own infectious delta as the reference's bounds check sees it (pyx:2473: a migrant is booked
//            on its SOURCE compartment there),
//   ownApp : the own delta UpdateCompartmentCounts_tau applies (pyx:2548: the migrant infects the TARGET population),
//   the susceptible deltas (identical in both), the tentative counters, multievent rows and the list of arrivals.
// The compartment arrays themselves are not touched, so every thread sees the pre-step state.
// Returns 2 when the compartment expects a.big_lam (VGX_TAU_BIG / VGX_TAU_BIG_SMALL) events or more (nothing drawn: vgx_tau_draw_big_kernel's case), else 0/1.
// MODE 1 (DRY): no bookkeeping at all; ownChk returns the number of mutants that go to haplotype `target` (the same random
// numbers in the same order, so the count is the one the compartment's real draw produces).
// MODE 2 (CHECK): no bookkeeping either, but ownChk / ownApp are the real draw's (births, migrants and mutants are counted, their
// targets drawn and dropped): what the front pass of a try needs to know whether the compartment falls below zero on its own.
template <int MODE, int TABS>
static __device__ __forceinline__ int tau_cell_events(const VgxTauArgs &a, const TauTabT<TABS> &T, const TauEnv &E, int rep, int pn, int hn, double tau,
                                                      int64_t Icell, uint32_t bucket, int64_t &ownChk, int64_t &ownApp, int64_t *cnt,
                                                      WaveStage *stage, unsigned long long *sS /* LDS [S]: this population's susceptible deltas */,
                                                      int target, int cls = -1 /* the compartment's rate class if the caller has it */,
                                                      long long *prof = nullptr /* diagnostic build: [0] last stamp, [1..5] phases */) {
#define CEPROF(i) do { if (prof) { const long long t_ = clock64(); prof[i] += t_ - prof[0]; prof[0] = t_; } } while (0)
    constexpr bool DRY = MODE == 1, CHECK = MODE == 2;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H, sites = p.sites;
    ownChk = 0;
    ownApp = 0;
    if (Icell <= 0) return 0;
    const double Ih = (double)Icell;
    const int c = cls >= 0 ? cls : ((p.C == 1) ? 0 : p.cls[hn]);
    const int cb = T.c_bidx[c];
    const int st = T.c_stype[c];
    // ---- channel rates per unit time ----
    const double r_rec = T.c_d[c] * Ih;                                              // pyx:2386
    const double r_samp = T.c_s[c] * Ih * E.sampMult;                                // pyx:2392
    const double r_tr = T.rtr[cb] * Ih;                                              // pyx:2412-2414 summed over sn
    const double r_mut = (a.mut_uniform ? a.mut_total : T.c_tm[c]) * Ih;             // pyx:2400-2401 summed over (s, i)
    const double r_mig = T.rmig[cb] * Ih;                                            // pyx:2366-2367 summed over (tpn, sn)
    const double r_all = r_mig + r_rec + r_samp + r_mut + r_tr;
    const double lam = r_all * tau;
    if (!(lam > 0.0)) return 0;
    if (lam >= a.big_lam) return 2;
    TauRng g;
    g.init(E.seed, E.att, (uint64_t)pn * (uint64_t)H + (uint64_t)hn, E.step, E.retry);
    int64_t N;
    int64_t rec = 0, samp = 0, births = 0, n_mut = 0, n_mig = 0;
    int bsn[4] = {0, 0, 0, 0};   // births per susceptibility group (first four): ONE multievent row per channel, as upstream's
    const bool by_kind = lam >= VGX_TAU_KINDS;
    if (by_kind) {
        // A mean of several events: one Poisson draw per KIND of event (recoveries, samples, mutants, migrants, births per
        // susceptibility group) instead of the total and a walk through the events one by one (half a Philox block each, and
        // the lane with the most events holds its wavefront): Poisson splitting, the same joint law.  (Below a mean of 1 the
        // scan kernel's bucket test relies on the FIRST uniform deciding "no event": that form stays for small means.)
        N = 0;
        for (int kd = 0; kd < 4 + S; ++kd) {
            double lk;
            if (kd == 0) lk = r_rec; else if (kd == 1) lk = r_samp; else if (kd == 2) lk = r_mut; else if (kd == 3) lk = r_mig;
            else lk = T.wtr[cb * S + (kd - 4)] * Ih;
            const int64_t x = tau_poisson(g, lk * tau);
            N += x;
            if (kd == 0) rec = x; else if (kd == 1) samp = x; else if (kd == 2) n_mut = x; else if (kd == 3) n_mig = x;
            else if (x != 0 && !DRY) {   // transmission to susceptibility group sn (pyx:2515-2520 / 2589-2593)
                const int sn = kd - 4;
                births += x;
                if (CHECK) continue;
                if (sn < 4) { cnt[8 + sn] -= x; bsn[sn] += (int)x; }
                else { atomicAdd(&sS[sn], (unsigned long long)(-x)); tau_row(a, rep, x, 0, hn, pn, sn, 0); }
            }
        }
    } else {   // inversion by sequential search (same law as numpy's sampler below 10)
        const double u = ((double)bucket + g.uniform()) * (1.0 / 256.0);
        double pk = exp(-lam), F = pk;
        N = 0;
        while (u > F && N < 200) {
            N += 1;
            pk *= lam * __builtin_amdgcn_rcp((double)N);   // (v_rcp_f64: within an ulp of lam / N at a third of a division's cost)
            F += pk;
        }
    }
    CEPROF(1);
    if (N == 0) return 0;
    const double t1 = r_rec, t2 = t1 + r_samp, t3 = t2 + r_tr, t4 = t3 + r_mut;
    for (int64_t ev = 0; ev < (by_kind ? 0 : N); ++ev) {
        double u = g.uniform() * r_all;
        if (u < t1) rec += 1;
        else if (u < t2) samp += 1;
        else if (u < t3 || (r_mut == 0.0 && r_mig == 0.0)) {
            if (DRY) continue;
            // transmission to susceptibility group sn (pyx:2515-2520 / 2589-2593)
            int sn_hit = -1;
            if (S == 1) {                          // one group: nothing to choose (and no division for the whole wavefront)
                if (T.wtr[cb] > 0.0) sn_hit = 0;
            } else {
                double uu = (u - t2) / Ih, acc = 0.0;
                for (int sn = 0; sn < S; ++sn) {
                    double w = T.wtr[cb * S + sn];
                    acc += w;
                    if (w > 0.0) sn_hit = sn;      // last positive channel so far: fallback at the upper end
                    if (uu < acc) break;
                }
            }
            if (sn_hit < 0) continue;
            births += 1;
            if (CHECK) continue;
            if (sn_hit < 4) { cnt[8 + sn_hit] -= 1; bsn[sn_hit] += 1; }
            else { atomicAdd(&sS[sn_hit], (unsigned long long)(-1ll)); tau_row(a, rep, 1, 0, hn, pn, sn_hit, 0); }
        } else if (u < t4 || r_mig == 0.0) n_mut += 1;
        else n_mig += 1;
    }
    CEPROF(2);
    // ---- mutants (pyx:2506-2512 / 2579-2586): site and derived state ----
    int64_t mut_done = 0, to_target = 0;
    for (int64_t k = 0; k < n_mut; ++k) {
        int ss = -1, ii = 0;
        if (a.mut_uniform) {
            const int nch = 3 * sites;
            double uu = g.uniform() * T.mutcum[nch - 1];
            int lo = 0, hi = nch - 1;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (T.mutcum[mid] > uu) hi = mid; else lo = mid + 1;
            }
            while (lo > 0 && T.mutcum[lo] == T.mutcum[lo - 1]) lo -= 1;   // a zero-weight channel can only be hit by rounding
            if (T.mutcum[nch - 1] > 0.0) { ss = lo / 3; ii = lo % 3; }
        } else {
            double uu = g.uniform() * T.c_tm[c], acc = 0.0;
            for (int s = 0; s < sites && ss < 0; ++s) {
                const double *hm = p.hapMutType + ((int64_t)hn * sites + s) * 3;
                double wsum = hm[0] + hm[1] + hm[2], mr = p.mRate[(int64_t)hn * sites + s];
                for (int i = 0; i < 3; ++i) {
                    acc += mr * hm[i] / wsum;
                    if (uu < acc) { ss = s; ii = i; break; }
                }
            }
            if (ss < 0)
                for (int s = sites - 1; s >= 0 && ss < 0; --s) {
                    const double *hm = p.hapMutType + ((int64_t)hn * sites + s) * 3;
                    for (int i = 2; i >= 0; --i)
                        if (p.mRate[(int64_t)hn * sites + s] * hm[i] > 0.0) { ss = s; ii = i; break; }
                }
        }
        if (ss < 0) continue;
        int nh = tau_mutate(sites, hn, ss, ii);
        mut_done += 1;
        if (DRY) { if (nh == target) to_target += 1; continue; }
        if (CHECK) continue;
        tau_list_add(a, stage, rep, (int64_t)pn * H + nh, 1, false);
        tau_row(a, rep, 1, 3, hn, pn, nh, 0);
    }
    if (DRY) { ownChk = to_target; return 1; }
    CEPROF(3);
    // ---- migrants (pyx:2464-2474 / 2541-2550): target population and susceptibility group, by bisection in the
    // cumulative channel weights of this source population and birth class ----
    int64_t *dS = a.dSi + (int64_t)rep * P * S;
    int64_t migrants = 0;
    for (int64_t k = 0; k < n_mig; ++k) {
        const typename TauTabSel<TABS>::CP cdf = T.cdf + (int64_t)cb * P * S;
        const int nch = P * S;
        double uu = g.uniform() * cdf[nch - 1];
        int lo = 0, hi = nch - 1;
        while (lo < hi) {               // first channel with cdf > uu
            int mid = (lo + hi) >> 1;
            if (cdf[mid] > uu) hi = mid; else lo = mid + 1;
        }
        while (lo > 0 && cdf[lo] == cdf[lo - 1]) lo -= 1;
        int tp = lo / S, ts = lo % S;
        if (tp == pn || !(cdf[nch - 1] > 0.0)) continue;
        migrants += 1;
        if (CHECK) continue;
        tau_list_add(a, stage, rep, (int64_t)tp * H + hn, 1, true);
        atomicAdd((unsigned long long *)&dS[tp * S + ts], (unsigned long long)(-1ll));
        atomicAdd((unsigned long long *)&a.dTot[(int64_t)rep * P + tp], 1ull);
        tau_row(a, rep, 1, 5, hn, pn, ts, tp);
    }
    CEPROF(4);
    if (CHECK) {
        ownChk = births - rec - samp - mut_done + migrants;
        ownApp = births - rec - samp - mut_done;
        return 1;
    }
    cnt[0] += births; cnt[1] += rec; cnt[2] += samp; cnt[3] += mut_done; cnt[5] += migrants;
    for (int sn = 0; sn < 4; ++sn)
        if (bsn[sn]) tau_row(a, rep, bsn[sn], 0, hn, pn, sn, 0);
    if (rec) tau_row(a, rep, rec, 1, hn, pn, st, 0);
    if (samp) tau_row(a, rep, samp, 2, hn, pn, st, 0);
    const int64_t own = births - rec - samp - mut_done;
    ownChk = own + migrants;
    ownApp = own;
    if (rec + samp != 0) { if (st < 4) cnt[8 + st] += rec + samp; else atomicAdd(&sS[st], (unsigned long long)(rec + samp)); }
    const int64_t dt = births - rec - samp;
    cnt[6] += dt;             // delta of totalInfectious[pn], added to the block's total at the end
    cnt[7] += dt + migrants;  // ... and the same sum as the check books it (mutants cancel inside the population)
    CEPROF(5);
    return 1;
#undef CEPROF
}

// LDS budget of the events kernel's tables (doubles, then int32); 0 = tables stay in global memory
static __host__ __device__ inline size_t tau_tab_lds_bytes(int C, int CB, int S, int P, bool &cdf_in_lds) {
    cdf_in_lds = false;
    if (C > 256 || CB > 16) return 0;
    size_t dbl = 4 * (size_t)C + 2 * (size_t)CB + (size_t)CB * S + 48;
    if ((size_t)CB * P * S <= 4096) { cdf_in_lds = true; dbl += (size_t)CB * P * S; }
    return dbl * 8 + 2 * (size_t)C * 4;
}

// A try is drawn by two kernels.  vgx_tau_scan_kernel streams over all compartments (light: many wavefronts per CU, HBM-bound)
// and queues the few that may draw events; vgx_tau_events_kernel (heavy: the whole event logic, few wavefronts per CU) works
// the queue off with all lanes busy.  The queue is sharded by (population, block of the scan kernel): block (bx, pn) of the
// events kernel takes the entries block (bx, pn) of the scan kernel queued, so its tables are those of ONE population.
#define VGX_QW 512          // entries a wavefront of the scan kernel stages in LDS: a quarter tile adds at most 256, they are moved out from 256 on
#define EB 64               // threads per block of the events kernel
#ifndef VGX_EV_WAVES
#define VGX_EV_WAVES 2      // wavefronts per SIMD the events kernel's register allocation aims at
#endif
#ifndef VGX_EV_CHUNK
#define VGX_EV_CHUNK 8      // rounds whose queue entries and counts a wavefront of the events kernel loads in one go
#endif
#ifndef VGX_EV_SHORT_CHUNKS
#define VGX_EV_SHORT_CHUNKS 1   // the first chunks of a wavefront of the events kernel are 1, 1, 2, 4 rounds long (a rejected try is noticed early)
#endif
#ifndef VGX_EV_BLOCKS
#define VGX_EV_BLOCKS 2048  // wavefronts of the events kernel per launch: what the chip holds at VGX_EV_WAVES per SIMD
#endif
static __host__ __device__ inline unsigned tau_draw_gx(int64_t H) {   // blocks of the scan kernel per (population, replicate)
    const unsigned tiles = (unsigned)((H + 1023) >> 10);              // wave tiles of 1024 haplotypes
    const unsigned blocks = (tiles + (TB / 64) - 1) / (TB / 64);
    return blocks < VGX_DRAW_GX ? blocks : VGX_DRAW_GX;
}
// most compartments one block of the scan kernel looks at (= the most entries its shard of the queue can get)
static __host__ __device__ inline int64_t tau_queue_shard_max(int64_t H) {
    const int64_t tiles = (H + 1023) >> 10, waves = (int64_t)tau_draw_gx(H) * (TB / 64);
    return (tiles + waves - 1) / waves * (TB / 64) * 1024;
}

// grid = (tau_draw_gx(H), P, R)
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_scan_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, C = p.C, CB = p.CB, H = p.H;
    __shared__ float s_rt[256];
    __shared__ double s_rtr[16], s_rmig[16];
    __shared__ int64_t q[TB / 64][VGX_QW];
    const double tau = a.tau[rep];
    const double F = a.F[(int64_t)rep * P + pn];
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    const bool tab = C <= 256 && CB <= 16;   // (CB <= 16 whenever C > 256: checked by the host)
    for (int cb = threadIdx.x; cb < CB && cb < 16; cb += TB) {
        double r = 0.0;
        for (int sn = 0; sn < S; ++sn) r += p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F;
        s_rtr[cb] = r;
        s_rmig[cb] = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
    }
    __syncthreads();
    if (tab)
        for (int i = threadIdx.x; i < C; i += TB) {   // the terms of r_all in tau_cell_events, per infected
            const int cb = p.c_bidx[i];
            const double r1 = s_rmig[cb] + p.c_d[i] + p.c_s[i] * p.sampMult[pn] + (a.mut_uniform ? a.mut_total : p.c_tm[i]) + s_rtr[cb];
            // rounded up: covers the roundings of the single-precision test and the other summation order of r_all
            s_rt[i] = (float)(r1 * tau * (1.0 + 1.0 / 1048576.0)) * (1.0f + 1.0f / 1048576.0f);
        }
    __syncthreads();
    const int L = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t *Irow = a.I + ((int64_t)rep * P + pn) * H;
    const uint8_t *I8row = a.I8 + ((int64_t)rep * P + pn) * H;   // min(count, 255), written by this step's drift pass
    int32_t *dCrow = a.dChk + ((int64_t)rep * P + pn) * H, *dArow = a.dApp + ((int64_t)rep * P + pn) * H;
    const bool dense = !a.sparse;
    const uint32_t key[2] = {(uint32_t)a.seeds[rep] ^ ((uint32_t)a.attempt[rep] * 0x9E3779B9u),
                             (uint32_t)((uint64_t)a.seeds[rep] >> 32) ^ 0x85EBCA6Bu};
    const uint32_t ctr_step = (uint32_t)a.step[rep], ctr_retry = ((uint32_t)a.retry[rep] << 20) | 0xFFFFFu;   // loop invariants
    const int tiles = (H + 1023) >> 10;
    const uint64_t groups = (uint64_t)((H + 15) >> 4);
    const bool vec = (H & 15) == 0;   // rows of the one-byte copy are 16-byte aligned
    const int64_t scap = a.q_cap / a.q_shards;
    const int64_t shard = (int64_t)pn * gridDim.x + blockIdx.x;
    unsigned long long *qn = a.q_n + (int64_t)rep * a.q_shards + shard;
    int64_t *qdst = a.q + (int64_t)rep * a.q_cap + shard * scap;
    int nq = 0;                      // wave-uniform: entries in this wavefront's stage
    auto flush = [&]() {             // wave-uniform call
        WSYNC();
        unsigned long long base = 0;
        if (L == 0) base = atomicAdd(qn, (unsigned long long)nq);
        base = (unsigned long long)bcast_i64((int64_t)base, 0);
        for (int i = L; i < nq; i += 64)
            if ((int64_t)(base + i) < scap) qdst[base + i] = q[wave][i];   // a full shard is detected by the events kernel
        WSYNC();
        nq = 0;
    };
    // A lane looks at 16 consecutive compartments of a wave tile of 1024: ONE 16-byte load of the one-byte copy of the counts
    // (the instruction of the wave covers one contiguous KiB) and the 16 bytes of one Philox block.  Only a lane that meets a
    // saturated byte (255 hosts or more) reads its counts proper.
    const int wstride = gridDim.x * (TB / 64);
    auto load_tile = [&](int t) -> uint4 {
        const int h0 = (t << 10) + 16 * L;
        if (vec && h0 + 15 < H) return *(const uint4 *)(I8row + h0);
        uint32_t x[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; ++j)
            if (h0 + j < H) x[j >> 2] |= (uint32_t)I8row[h0 + j] << (8 * (j & 3));
        return make_uint4(x[0], x[1], x[2], x[3]);
    };
    const int wt0 = blockIdx.x * (TB / 64) + wave;
    uint4 nx = make_uint4(0, 0, 0, 0);
    if (wt0 < tiles) nx = load_tile(wt0);
    for (int wt = wt0; wt < tiles; wt += wstride) {
        const uint4 cur = nx;
        if (wt + wstride < tiles) nx = load_tile(wt + wstride);   // the next tile's load is in flight while this one is worked on
        const int hl = (wt << 10) + 16 * L;                       // the lane's first haplotype
        const uint64_t gidx = (uint64_t)pn * groups + (uint64_t)(hl >> 4);
        const uint32_t ctr[4] = {(uint32_t)gidx, (uint32_t)(gidx >> 32), ctr_step, ctr_retry};
        uint32_t w[4];
        vgx_philox4x32(ctr, key, w);
        const uint32_t b8[4] = {cur.x, cur.y, cur.z, cur.w};
        // a byte is 255 iff its low seven bits + 1 carry into its top bit and that bit is set
        auto has255 = [](uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u) != 0u; };
        const bool sat = has255(cur.x) || has255(cur.y) || has255(cur.z) || has255(cur.w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h0 = hl + 4 * k;
            int Iv[4] = {(int)(b8[k] & 255u), (int)((b8[k] >> 8) & 255u), (int)((b8[k] >> 16) & 255u), (int)(b8[k] >> 24)};
            if (sat) {   // some count of this lane does not fit a byte: the four counts proper
                for (int j = 0; j < 4; ++j)
                    if (Iv[j] == 255 && h0 + j < H) Iv[j] = Irow[h0 + j];
            }
            int cl[4] = {0, 0, 0, 0};
            if (C != 1) {
                if ((H & 3) == 0 && h0 + 3 < H) { const int4 cc = *(const int4 *)(p.cls + h0); cl[0] = cc.x; cl[1] = cc.y; cl[2] = cc.z; cl[3] = cc.w; }
                else for (int j = 0; j < 4; ++j) if (h0 + j < H) cl[j] = p.cls[h0 + j];
            }
            int qm = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t b = (w[k] >> (8 * j)) & 255u;
                bool qd;
                if (tab) {
                    const float thr = fmaf(-256.0f, s_rt[cl[j]] * (float)Iv[j], 255.999f);
                    qd = Iv[j] > 0 && (float)(b + 1u) > thr;
                } else {
                    const int c = cl[j];
                    const int cb = p.c_bidx[c];
                    const double rate = s_rmig[cb] + p.c_d[c] + p.c_s[c] * p.sampMult[pn] + (a.mut_uniform ? a.mut_total : p.c_tm[c]) + s_rtr[cb];
                    const double lam = rate * (double)Iv[j] * tau * (1.0 + 1e-9);
                    qd = Iv[j] > 0 && (double)(b + 1u) > 256.0 - 256.0 * lam;
                }
                if (qd && h0 + j < H) qm |= 1 << j;
            }
            if (dense && h0 < H) {
                // zero deltas of the compartments that draw nothing here (the queued ones are written by the events kernel, large
                // ones by vgx_tau_draw_big_kernel): every entry of both arrays is written in every try
                if (qm == 0 && (H & 3) == 0 && h0 + 3 < H) {
                    *(int4 *)(dCrow + h0) = make_int4(0, 0, 0, 0);
                    *(int4 *)(dArow + h0) = make_int4(0, 0, 0, 0);
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (h0 + j < H && !((qm >> j) & 1)) { dCrow[h0 + j] = 0; dArow[h0 + j] = 0; }
                }
            }
            if (__any(qm != 0)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool qd = (qm >> j) & 1;
                    const unsigned long long m = __ballot(qd);
                    if (qd) {
                        const int slot = nq + __popcll(m & ((1ull << L) - 1ull));
                        q[wave][slot] = (int64_t)(h0 + j) | ((int64_t)((w[k] >> (8 * j)) & 255u) << 32);
                    }
                    nq += __popcll(m);
                }
                if (nq >= VGX_QW - 256) flush();
            }
        }
    }
    if (nq > 0) flush();
}

// The scan for the usual shapes: at most 16 rate classes (thresholds tabulated in LDS), haplotype count a multiple of 16.
// C1 = one rate class, DENSE = the dense delta arrays are written (validation modes).  Same decisions as the general kernel:
// a compartment of X < 255 hosts is queued iff bucket >= T[class][X], T = the smallest bucket the single-precision test of the
// general kernel queues (X = 0: never); counts of 255 or more take that test itself on the count proper.  Queued compartments
// are collected as one bit per compartment and written with one wave-wide prefix sum per half tile.
#define VGX_QWF 768          // stage of a wavefront: a half tile adds at most 512 entries, moved out from 256 on
template <bool C1, bool DENSE>
__global__ void __launch_bounds__(TB) vgx_tau_scan_fast_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    if (a.front_on && a.ok[rep] == 0) return;   // the front pass found a failure: the try is lost, nothing of it is needed
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, C = p.C, CB = p.CB, H = p.H;
    __shared__ float s_rt[16];
    __shared__ double s_rtr[16], s_rmig[16];
    __shared__ uint16_t s_thr[C1 ? 256 : 16 * 256];
    __shared__ int64_t q[TB / 64][VGX_QWF];
    const double tau = a.tau[rep];
    const double F = a.F[(int64_t)rep * P + pn];
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    for (int cb = threadIdx.x; cb < CB && cb < 16; cb += TB) {
        double r = 0.0;
        for (int sn = 0; sn < S; ++sn) r += p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F;
        s_rtr[cb] = r;
        s_rmig[cb] = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += TB) {   // as in vgx_tau_scan_kernel
        const int cb = p.c_bidx[i];
        const double r1 = s_rmig[cb] + p.c_d[i] + p.c_s[i] * p.sampMult[pn] + (a.mut_uniform ? a.mut_total : p.c_tm[i]) + s_rtr[cb];
        s_rt[i] = (float)(r1 * tau * (1.0 + 1.0 / 1048576.0)) * (1.0f + 1.0f / 1048576.0f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 256; i += TB) {
        const int X = i & 255;
        const float thr = fmaf(-256.0f, s_rt[i >> 8] * (float)X, 255.999f);
        // queued iff (float)(bucket + 1) > thr  <=>  bucket >= floor(thr) (thr >= 0), always (thr < 0), never (X = 0)
        s_thr[i] = X == 0 ? (uint16_t)256 : (uint16_t)(thr < 0.0f ? 0 : (int)floorf(thr));
    }
    __syncthreads();
    const int L = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t *Irow = a.I + ((int64_t)rep * P + pn) * H;
    const uint8_t *I8row = a.I8 + ((int64_t)rep * P + pn) * H;
    int32_t *dCrow = a.dChk + ((int64_t)rep * P + pn) * H, *dArow = a.dApp + ((int64_t)rep * P + pn) * H;
    const uint32_t key[2] = {(uint32_t)a.seeds[rep] ^ ((uint32_t)a.attempt[rep] * 0x9E3779B9u),
                             (uint32_t)((uint64_t)a.seeds[rep] >> 32) ^ 0x85EBCA6Bu};
    const uint32_t ctr_step = (uint32_t)a.step[rep], ctr_retry = ((uint32_t)a.retry[rep] << 20) | 0xFFFFFu;
    const int tiles = (H + 1023) >> 10;
    const uint64_t groups = (uint64_t)(H >> 4);
    const int64_t scap = a.q_cap / a.q_shards;
    const int64_t shard = (int64_t)pn * gridDim.x + blockIdx.x;
    unsigned long long *qn = a.q_n + (int64_t)rep * a.q_shards + shard;
    int64_t *qdst = a.q + (int64_t)rep * a.q_cap + shard * scap;
    int nq = 0;                      // wave-uniform: entries in this wavefront's stage
    const int wstride = gridDim.x * (TB / 64);
    const int wt0 = blockIdx.x * (TB / 64) + wave;
    uint4 nx = make_uint4(0, 0, 0, 0);
    if (wt0 < tiles && (wt0 << 10) + 16 * L < H) nx = *(const uint4 *)(I8row + (wt0 << 10) + 16 * L);
    for (int wt = wt0; wt < tiles; wt += wstride) {
        const uint4 cur = nx;
        {
            const int hn = ((wt + wstride) << 10) + 16 * L;
            if (wt + wstride < tiles && hn < H) nx = *(const uint4 *)(I8row + hn);   // in flight while this tile is worked on
        }
        const int hl = (wt << 10) + 16 * L;                       // the lane's first haplotype (a lane beyond H sees zeros)
        const uint64_t gidx = (uint64_t)pn * groups + (uint64_t)(hl >> 4);
        const uint32_t ctr[4] = {(uint32_t)gidx, (uint32_t)(gidx >> 32), ctr_step, ctr_retry};
        uint32_t w[4];
        vgx_philox4x32(ctr, key, w);
        const uint32_t b8[4] = {cur.x, cur.y, cur.z, cur.w};
        auto has255 = [](uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u) != 0u; };
        const bool sat = has255(cur.x) || has255(cur.y) || has255(cur.z) || has255(cur.w);
        uint32_t qbits = 0;          // bit 4k + j: compartment hl + 4k + j is queued
        int4 clv[4] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
        if (!C1 && hl < H) {
#pragma unroll
            for (int k = 0; k < 4; ++k) clv[k] = *(const int4 *)(p.cls + hl + 4 * k);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cl[4] = {C1 ? 0 : clv[k].x, C1 ? 0 : clv[k].y, C1 ? 0 : clv[k].z, C1 ? 0 : clv[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t X = (b8[k] >> (8 * j)) & 255u, b = (w[k] >> (8 * j)) & 255u;
                const uint32_t T = s_thr[(C1 ? 0 : (cl[j] << 8)) + X];
                if (b >= T) qbits |= 1u << (4 * k + j);
            }
        }
        if (hl >= H) qbits = 0;   // (a lane beyond the row: its counts read as zero, nothing is queued)
        if (sat && hl < H) {   // counts of 255 or more: the general kernel's test on the count proper
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (int j = 0; j < 4; ++j)
                    if (((b8[k] >> (8 * j)) & 255u) == 255u) {
                        const int c = C1 ? 0 : (j == 0 ? clv[k].x : j == 1 ? clv[k].y : j == 2 ? clv[k].z : clv[k].w);
                        const float thr = fmaf(-256.0f, s_rt[c] * (float)Irow[hl + 4 * k + j], 255.999f);
                        const uint32_t b = (w[k] >> (8 * j)) & 255u;
                        if ((float)(b + 1u) > thr) qbits |= 1u << (4 * k + j); else qbits &= ~(1u << (4 * k + j));
                    }
        }
        if (DENSE && hl < H) {
            // zero deltas of the compartments that draw nothing here: every entry of both arrays is written in every try
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t qm = (qbits >> (4 * k)) & 15u;
                if (qm == 0) {
                    *(int4 *)(dCrow + hl + 4 * k) = make_int4(0, 0, 0, 0);
                    *(int4 *)(dArow + hl + 4 * k) = make_int4(0, 0, 0, 0);
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (!((qm >> j) & 1u)) { dCrow[hl + 4 * k + j] = 0; dArow[hl + 4 * k + j] = 0; }
                }
            }
        }
        // the queued compartments of the tile, half a tile at a time: a wave-wide prefix sum of the lanes' counts, then
        // every lane writes its own entries
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t hb = (qbits >> (8 * half)) & 255u;
            if (!__any(hb != 0)) continue;
            const int cnt = __popc(hb);
            int pre = cnt;   // inclusive prefix over the lanes
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(pre, o);
                if (L >= o) pre += v;
            }
            const int total = __builtin_amdgcn_readlane(pre, 63);
            int slot = nq + pre - cnt;
            uint32_t m = hb;
            while (m) {
                const int i = __ffs((int)m) - 1;
                m &= m - 1;
                const int cell = 8 * half + i;
                q[wave][slot++] = (int64_t)(hl + cell) | ((int64_t)((w[cell >> 2] >> (8 * (cell & 3))) & 255u) << 32);
            }
            nq += total;
            if (nq >= VGX_QWF - 512) {
                WSYNC();
                unsigned long long base = 0;
                if (L == 0) base = atomicAdd(qn, (unsigned long long)nq);
                base = (unsigned long long)bcast_i64((int64_t)base, 0);
                for (int i = L; i < nq; i += 64)
                    if ((int64_t)(base + i) < scap) qdst[base + i] = q[wave][i];   // a full shard is detected by the events kernel
                WSYNC();
                nq = 0;
            }
        }
    }
    if (nq > 0) {
        WSYNC();
        unsigned long long base = 0;
        if (L == 0) base = atomicAdd(qn, (unsigned long long)nq);
        base = (unsigned long long)bcast_i64((int64_t)base, 0);
        for (int i = L; i < nq; i += 64)
            if ((int64_t)(base + i) < scap) qdst[base + i] = q[wave][i];
    }
}

// ---- the front pass of a try: the likely failures first -------------------------------------------------------------------
// A try of the halving loop is rejected as soon as ONE compartment falls below zero on its own and no mutant arrives to rescue it
// (pyx:2522-2528), and most tries are rejected (three of four at config 4).  A compartment of X hosts can fall below zero only with
// N >= X + 1 events, i.e. with its first uniform u in the far upper tail: 1 - u < P(N >= X + 1) <= lam^(X+1) / (X+1)!.  The top
// eight bits of u are the bucket byte of the group's Philox block (see above), so this kernel — the scan's Philox block per 16
// compartments and nothing else for the 15 of 16 lanes that see no bucket of 255 — finds every compartment that CAN fail that way
// (lam rounded up, the bound evaluated in single precision with slack: a superset), a few dozen to a few thousand per try, and
// lists them per population.  vgx_tau_events_kernel<TABS, true> then draws exactly those compartments with the code of the try
// proper (no bookkeeping) and clears `ok` on a definite failure; the scan and the events kernel of a try whose `ok` is already
// clear return at once.  What the front pass does not see (a failure with a smaller bucket: large means; compartments drawn kind by
// kind or channel by channel; the upper bounds) the try finds as before: the pass only ever ends a try early that would have been
// rejected anyway, results are bit for bit those without it (VGX_TAU_NO_FRONT=1).  grid = (tau_draw_gx(H), P, R); the tabulated scan's shapes
// (at most 16 rate classes).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_front_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H;
    const int C = p.C, CB = p.CB;
    __shared__ float s_rt[16];
    __shared__ double s_rtr[16], s_rmig[16];
    {   // as in vgx_tau_scan_fast_kernel: the terms of r_all per infected and class, times tau, rounded up
        const double F = a.F[(int64_t)rep * P + pn];
        const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
        for (int cb = threadIdx.x; cb < CB && cb < 16; cb += TB) {
            double r = 0.0;
            for (int sn = 0; sn < S; ++sn) r += p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F;
            s_rtr[cb] = r;
            s_rmig[cb] = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
        }
        __syncthreads();
        const double tau = a.tau[rep];
        for (int i = threadIdx.x; i < C && i < 16; i += TB) {
            const int cb = p.c_bidx[i];
            const double r1 = s_rmig[cb] + p.c_d[i] + p.c_s[i] * p.sampMult[pn] + (a.mut_uniform ? a.mut_total : p.c_tm[i]) + s_rtr[cb];
            s_rt[i] = (float)(r1 * tau * (1.0 + 1.0 / 1048576.0)) * (1.0f + 1.0f / 1048576.0f);
        }
    }
    const int L = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t *Irow = a.I + ((int64_t)rep * P + pn) * H;
    const uint8_t *I8row = a.I8 + ((int64_t)rep * P + pn) * H;
    const uint64_t seed = (uint64_t)a.seeds[rep];
    const uint32_t att = (uint32_t)a.attempt[rep], step = (uint32_t)a.step[rep], retry = (uint32_t)a.retry[rep];
    const uint32_t key[2] = {(uint32_t)seed ^ (att * 0x9E3779B9u), (uint32_t)(seed >> 32) ^ 0x85EBCA6Bu};
    const uint32_t ctr_retry = (retry << 20) | 0xFFFFFu;
    const int tiles = (H + 1023) >> 10;
    const uint64_t groups = (uint64_t)(H >> 4);
    unsigned int *fn = a.front_n + (int64_t)rep * P + pn;
    int64_t *fdst = a.front + ((int64_t)rep * P + pn) * a.front_cap;
    const int wstride = gridDim.x * (TB / 64);
    // The compartments with bucket 255 (one in 256) are collected per wavefront in LDS and looked at 64 at a time, one per lane: the
    // second Philox block and the bound are then paid by full wavefronts, not by the one lane in sixteen that holds such a bucket.
    __shared__ float s_lf[257];                     // log(n!)
    __shared__ int32_t s_c[TB / 64][64 + 1024];     // (a tile adds at most 1024)
    for (int i = threadIdx.x; i < 257; i += TB) s_lf[i] = vgx_tau_logfact_f[i];
    __syncthreads();
    int nq = 0;                                     // wave-uniform
    auto look = [&](int base, int cnt) {            // `cnt` (<= 64) collected compartments from `base` on, one per lane
        if (L < cnt) {
            const int hn = s_c[wave][base + L];
            int64_t X = I8row[hn];
            if (X == 255) X = Irow[hn];
            if (X > 0) {
                const float lam = s_rt[C == 1 ? 0 : p.cls[hn]] * (float)X, n = (float)X + 1.0f;
                bool cand = true;
                if (lam < 8.0f && X < 256) {   // (beyond: drawn kind by kind or close to it, always listed)
                    TauRng g;
                    g.init(seed, att, (uint64_t)pn * (uint64_t)H + (uint64_t)hn, step, retry);
                    const double u52 = g.uniform();
                    const float lhs = __logf((float)(1.0 - u52)) - 5.5451775f;          // log(1 - u), rounded down (log 256 = 5.545177444)
                    const float rhs = n * __logf(lam) - s_lf[(int)X + 1] + 0.01f * n + 0.05f;   // log(lam^n / n!) with slack
                    // (1 - u within rounding of 0: the inversion's running sum reaches 1.0 in double precision before the exact one would)
                    cand = lhs <= rhs || (1.0 - u52) < 1e-11;
                }
                if (cand) {
                    const unsigned int slot = atomicAdd(fn, 1u);
                    if ((int)slot < a.front_cap) fdst[slot] = (int64_t)hn | ((int64_t)255 << 32);
                }
            }
        }
    };
    for (int wt = blockIdx.x * (TB / 64) + wave; wt < tiles; wt += wstride) {
        const int hl = (wt << 10) + 16 * L;
        uint32_t z[4] = {0, 0, 0, 0};               // bit 8 j + 7 of z[k]: the bucket of compartment hl + 4 k + j is 255
        if (hl < H) {
            const uint64_t gidx = (uint64_t)pn * groups + (uint64_t)(hl >> 4);
            const uint32_t ctr[4] = {(uint32_t)gidx, (uint32_t)(gidx >> 32), step, ctr_retry};
            uint32_t w[4];
            vgx_philox4x32(ctr, key, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t y = ~w[k];           // a zero byte of y <-> a byte 255 of w (exact per byte)
                z[k] = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
            }
        }
        const int cnt = __popc(z[0]) + __popc(z[1]) + __popc(z[2]) + __popc(z[3]);
        if (!__any(cnt != 0)) continue;
        int pre = cnt;                              // inclusive prefix over the lanes
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(pre, o);
            if (L >= o) pre += v;
        }
        const int total = __builtin_amdgcn_readlane(pre, 63);
        int slot = nq + pre - cnt;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t m = z[k];
            while (m) {
                const int bit = __ffs((int)m) - 1;
                m &= m - 1;
                s_c[wave][slot++] = hl + 4 * k + (bit >> 3);
            }
        }
        nq += total;
        WSYNC();
        while (nq >= 64) {   // (the last 64 of the stage: nothing has to move)
            look(nq - 64, 64);
            nq -= 64;
        }
        WSYNC();
    }
    if (nq > 0) look(0, nq);
}

// ---- sparse states: the scan and the front pass of a try over the LIST of occupied compartments ------------------------------------
// At natural occupancy (an epidemic grown from an index case: 0.65 % of config 4's compartments hold a host after SURVEY 8(d)'s
// warm-up) the two passes above draw a Philox block for every group of 16 compartments to find out that almost all of them are empty.
// vgx_tau_drift8_kernel, which streams every byte once per step anyway, lists the occupied compartments (one region per population,
// tile and wavefront: no atomics), and this kernel makes the same decisions for the listed compartments only: the compartment's
// bucket (byte hn & 15 of its group's block — the block of vgx_tau_scan_fast_kernel), queued iff bucket >= T[class][X] (the same
// table), and for a bucket of 255 the front pass's bound (vgx_tau_front_kernel).  Same queue entries, same front lists, another
// order.  A region with more occupied compartments than its list holds (VGX_OCC_CAP of 8192: the neighbourhood of a large lineage)
// is swept compartment by compartment instead.
// grid = (tau_draw_gx(H), P, R); the blocks' shards of the queue as in the scan.
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_listscan_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, C = p.C, CB = p.CB, H = p.H;
    __shared__ float s_rt[16], s_lf[257];
    __shared__ double s_rtr[16], s_rmig[16];
    __shared__ uint16_t s_thr[16 * 256];
    __shared__ int64_t q[TB / 64][256];
    {
        const double tau = a.tau[rep];
        const double F = a.F[(int64_t)rep * P + pn];
        const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
        for (int cb = threadIdx.x; cb < CB && cb < 16; cb += TB) {
            double r = 0.0;
            for (int sn = 0; sn < S; ++sn) r += p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F;
            s_rtr[cb] = r;
            s_rmig[cb] = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C && i < 16; i += TB) {   // as in vgx_tau_scan_fast_kernel
            const int cb = p.c_bidx[i];
            const double r1 = s_rmig[cb] + p.c_d[i] + p.c_s[i] * p.sampMult[pn] + (a.mut_uniform ? a.mut_total : p.c_tm[i]) + s_rtr[cb];
            s_rt[i] = (float)(r1 * tau * (1.0 + 1.0 / 1048576.0)) * (1.0f + 1.0f / 1048576.0f);
        }
        for (int i = threadIdx.x; i < 257; i += TB) s_lf[i] = vgx_tau_logfact_f[i];
        __syncthreads();
        for (int i = threadIdx.x; i < C * 256 && i < 16 * 256; i += TB) {
            const int X = i & 255;
            const float thr = fmaf(-256.0f, s_rt[i >> 8] * (float)X, 255.999f);
            s_thr[i] = X == 0 ? (uint16_t)256 : (uint16_t)(thr < 0.0f ? 0 : (int)floorf(thr));
        }
        __syncthreads();
    }
    const int L = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t *Irow = a.I + ((int64_t)rep * P + pn) * H;
    const uint8_t *I8row = a.I8 + ((int64_t)rep * P + pn) * H;
    const uint64_t seed = (uint64_t)a.seeds[rep];
    const uint32_t att = (uint32_t)a.attempt[rep], step = (uint32_t)a.step[rep], retry = (uint32_t)a.retry[rep];
    const uint32_t key[2] = {(uint32_t)seed ^ (att * 0x9E3779B9u), (uint32_t)(seed >> 32) ^ 0x85EBCA6Bu};
    const uint32_t ctr_retry = (retry << 20) | 0xFFFFFu;
    const uint64_t groups = (uint64_t)(H >> 4);
    const int64_t scap = a.q_cap / a.q_shards;
    const int64_t shard = (int64_t)pn * gridDim.x + blockIdx.x;
    unsigned long long *qn = a.q_n + (int64_t)rep * a.q_shards + shard;
    int64_t *qdst = a.q + (int64_t)rep * a.q_cap + shard * scap;
    unsigned int *fn = a.front_n + (int64_t)rep * P + pn;
    int64_t *fdst = a.front + ((int64_t)rep * P + pn) * a.front_cap;
    int nq = 0;                      // wave-uniform: entries in this wavefront's stage
    auto flush = [&]() {
        WSYNC();
        unsigned long long base = 0;
        if (L == 0) base = atomicAdd(qn, (unsigned long long)nq);
        base = (unsigned long long)bcast_i64((int64_t)base, 0);
        for (int i = L; i < nq; i += 64)
            if ((int64_t)(base + i) < scap) qdst[base + i] = q[wave][i];   // a full shard is detected by the events kernel
        WSYNC();
        nq = 0;
    };
    // one compartment per lane: queued iff its bucket reaches the threshold of its count; a bucket of 255 also takes the front pass's test
    auto cell = [&](bool valid, int hn, int64_t X, uint32_t b) {
        const int cl = (C == 1 || !valid) ? 0 : p.cls[hn];
        bool queued = false;
        if (valid && X > 0) {
            if (X < 255) queued = b >= (uint32_t)s_thr[(cl << 8) + (int)X];
            else {               // 255 or more: the general kernel's test on the count proper
                X = Irow[hn];
                queued = (float)(b + 1u) > fmaf(-256.0f, s_rt[cl] * (float)X, 255.999f);
            }
        }
        if (queued && b == 255u) {      // the front pass's bound (vgx_tau_front_kernel)
            const float lam = s_rt[cl] * (float)X, nn = (float)X + 1.0f;
            bool cand = true;
            if (lam < 8.0f && X < 256) {
                TauRng g;
                g.init(seed, att, (uint64_t)pn * (uint64_t)H + (uint64_t)hn, step, retry);
                const double u52 = g.uniform();
                const float lhs = __logf((float)(1.0 - u52)) - 5.5451775f;
                const float rhs = nn * __logf(lam) - s_lf[(int)X + 1] + 0.01f * nn + 0.05f;
                cand = lhs <= rhs || (1.0 - u52) < 1e-11;
            }
            if (cand) {
                const unsigned int slot = atomicAdd(fn, 1u);
                if ((int)slot < a.front_cap) fdst[slot] = (int64_t)hn | ((int64_t)255 << 32);
            }
        }
        const unsigned long long m = __ballot(queued);
        if (m != 0ull) {
            if (queued) q[wave][nq + (int)__popcll(m & ((1ull << L) - 1ull))] = (int64_t)hn | ((int64_t)b << 32);
            nq += (int)__popcll(m);
            if (nq > 256 - 64) flush();
        }
    };
    const int low = p.sites < VGX_D8_LOW ? p.sites : VGX_D8_LOW, TSd = 1 << (2 * low - 2);   // the drift pass's tile: its dwords
    const int nwaves = gridDim.x * (TB / 64);
    for (int reg = blockIdx.x * (TB / 64) + wave; reg < a.occ_nreg; reg += nwaves) {
        const int64_t region = ((int64_t)rep * P + pn) * a.occ_nreg + reg;
        const int n = (int)a.occ_n[region];
        if (n > VGX_OCC_CAP) {
            // a densely occupied region (the neighbourhood of a large lineage): its list is incomplete, so its compartments are
            // swept — the cells the drift pass's wavefront `reg % 8` of tile `reg / 8` looked at, four per lane and turn
            const int tl = reg / VGX_D8_WAVES, wv = reg % VGX_D8_WAVES;
            for (int qd = wv * 64 + L; qd < TSd; qd += D8_TB) {
                const int h0 = (tl << (2 * low)) + 4 * qd;
                const uint32_t own = *(const uint32_t *)(I8row + h0);
                const uint64_t gidx = (uint64_t)pn * groups + (uint64_t)(h0 >> 4);
                const uint32_t ctr[4] = {(uint32_t)gidx, (uint32_t)(gidx >> 32), step, ctr_retry};
                uint32_t w[4];
                vgx_philox4x32(ctr, key, w);
                const uint32_t word = w[(h0 >> 2) & 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) cell(true, h0 + j, (int64_t)((own >> (8 * j)) & 255u), (word >> (8 * j)) & 255u);
            }
            continue;
        }
        const int32_t *src = a.occ + region * VGX_OCC_CAP;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const bool valid = i0 + L < n;
            const int hn = valid ? src[i0 + L] : 0;
            const int64_t X = valid ? (int64_t)I8row[hn] : 0;
            const uint64_t gidx = (uint64_t)pn * groups + (uint64_t)(hn >> 4);
            const uint32_t ctr[4] = {(uint32_t)gidx, (uint32_t)(gidx >> 32), step, ctr_retry};
            uint32_t w[4];
            vgx_philox4x32(ctr, key, w);
            cell(valid, hn, X, (w[(hn >> 2) & 3] >> (8 * (hn & 3))) & 255u);
        }
    }
    if (nq > 0) flush();
}

// In-kernel stamps of the events kernel (diagnostic build only, -DVGX_PROFILE; tools/profile_tau_events.py): shader cycles per
// phase summed over all wavefronts.  [0] prologue, [1] wait for the round's count, [2] draws + bookkeeping, [3] rescue tests,
// [4] staged list -> global, [5] epilogue, [7..11] inside tau_cell_events (lane 0's stamps: rates + count, split, mutants,
// migrants, tallies), [12] rounds, [13] wavefronts.
#ifdef VGX_PROFILE
__device__ unsigned long long vgx_tau_big_prof[8];   // vgx_tau_draw_big_kernel: setup loads, means, draws, books, sums + leader, flush, end, iterations
extern "C" int vgx_tau_get_big_profile(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vgx_tau_big_prof), sizeof(unsigned long long) * 8) != hipSuccess;
}
__device__ unsigned long long vgx_tau_ev_prof[16];
extern "C" int vgx_tau_get_profile(unsigned long long *out, int clear) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vgx_tau_ev_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (clear) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vgx_tau_ev_prof), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#define EVPROF(i) do { const long long t_ = clock64(); pacc[i] += t_ - pt; pt = t_; } while (0)
#else
#define EVPROF(i)
#endif
// grid = (tau_events_gx * ev_split, P, R), one wavefront per block: ev_split blocks share a shard of the queue and take its
// rounds of 64 entries in turn (few shards with many entries each — mid-size models — still fill the chip), and a block works
// through several shards of its population one after the other when there are more shards than the chip holds wavefronts;
// dSi / dTot / dChkTot are zero on entry.  The shards' counters are cleared by vgx_tau_decide_kernel.
// FRONT = true: the front pass of a try (see vgx_tau_front_kernel): the same code on the population's short list of likely
// failures, no bookkeeping (tau_cell_events MODE 2); all it may do is clear `ok`.  grid = (1, P, R).
template <int TABS, bool FRONT = false>
__global__ void __launch_bounds__(EB, VGX_EV_WAVES) vgx_tau_events_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
#ifdef VGX_PROFILE
    long long pacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prounds = 0;
    long long &pt = pacc[6];   // (tau_cell_events stamps through the same array: [6] = last stamp, [7..11] its phases)
    pt = clock64();
#endif
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, C = p.C, CB = p.CB, H = p.H;
    const int64_t scap = FRONT ? (int64_t)a.front_cap : a.q_cap / a.q_shards;
    const int split = FRONT ? 1 : a.ev_split, sub = FRONT ? 0 : (int)(blockIdx.x % split);
    // the block's shards of its population's part of the queue: sb0, sb0 + sbstep, ... (its tables, the state of the try and
    // the final sums are set up / added once per block: with one shard per block they were a quarter of a wavefront's life)
    const int shards_pop = FRONT ? 1 : (int)(a.q_shards / P), sb0 = FRONT ? 0 : (int)(blockIdx.x / split), sbstep = FRONT ? 1 : (int)(gridDim.x / split);
    const unsigned long long *qn_pop = a.q_n + (int64_t)rep * a.q_shards + (int64_t)pn * shards_pop;
    const int64_t front_n = FRONT ? min((int64_t)a.front_n[(int64_t)rep * P + pn], scap) : 0;   // (what did not fit is drawn with the rest)
    if (FRONT) {
        if (front_n == 0 || blockIdx.x != 0) return;
    } else {
        bool any = false;
        for (int sb = sb0; sb < shards_pop; sb += sbstep) any = any || (int64_t)sub * EB < (int64_t)qn_pop[sb];
        if (!any) return;
    }
    if (__hip_atomic_load(&a.ok[rep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // the try is already lost (see below)
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    __shared__ unsigned long long sS[64];
    __shared__ double g_rtr[16], g_rmig[16], g_wtr[TABS == 0 ? 16 * 64 : 1];   // fallback storage when the class tables stay global
    __shared__ WaveStage stage_s;
    __shared__ int64_t s_q[VGX_EV_CHUNK * EB];   // queue entries and counts of the rounds of a chunk, one column per lane
    __shared__ int32_t s_I[VGX_EV_CHUNK * EB], s_c[VGX_EV_CHUNK * EB];   // (s_c: rate classes, when there are several)
    const bool many_cls = C != 1;
    sS[threadIdx.x] = 0;
    const bool cdfL = TABS == 2;   // (the launcher picks the instantiation from tau_tab_lds_bytes)
    TauTabT<TABS> T;
    const double tau = a.tau[rep];
    const double F = a.F[(int64_t)rep * P + pn];
    const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
    const double *gcdf = a.migcdf + ((int64_t)rep * P + pn) * CB * (int64_t)P * S;
    if constexpr (TABS != 0) {
        VGX_AS3 double *d = (VGX_AS3 double *)dsm;
        VGX_AS3 double *l_cd = d; d += C;
        VGX_AS3 double *l_cs = d; d += C;
        VGX_AS3 double *l_ctm = d; d += C;
        VGX_AS3 double *l_rtr = d; d += CB;
        VGX_AS3 double *l_rmig = d; d += CB;
        VGX_AS3 double *l_wtr = d; d += CB * S;
        VGX_AS3 double *l_mut = d; d += 48;
        VGX_AS3 double *l_r1 = d; d += C;
        VGX_AS3 double *l_cdf = d; if (cdfL) d += CB * P * S;
        VGX_AS3 int32_t *l_bidx = (VGX_AS3 int32_t *)d;
        VGX_AS3 int32_t *l_stype = l_bidx + C;
        for (int i = threadIdx.x; i < C; i += EB) {
            l_cd[i] = p.c_d[i]; l_cs[i] = p.c_s[i]; l_ctm[i] = p.c_tm[i]; l_bidx[i] = p.c_bidx[i]; l_stype[i] = p.c_stype[i];
        }
        for (int i = threadIdx.x; i < CB * S; i += EB) l_wtr[i] = p.cb_b[i / S] * p.cb_sigma[i] * (double)Sus[i % S] * F;
        for (int i = threadIdx.x; i < 3 * p.sites && i < 48; i += EB) l_mut[i] = a.mutcum[i];
        if (cdfL)
            for (int i = threadIdx.x; i < CB * P * S; i += EB) l_cdf[i] = gcdf[i];
        __syncthreads();
        for (int cb = threadIdx.x; cb < CB; cb += EB) {
            double r = 0.0;
            for (int sn = 0; sn < S; ++sn) r += l_wtr[cb * S + sn];
            l_rtr[cb] = r;
            l_rmig[cb] = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
        }
        __syncthreads();
        (void)l_r1;
        T.c_d = l_cd; T.c_s = l_cs; T.c_tm = l_ctm; T.c_bidx = l_bidx; T.c_stype = l_stype;
        T.rtr = l_rtr; T.wtr = l_wtr; T.rmig = l_rmig; T.mutcum = l_mut;
        if constexpr (TABS == 2) T.cdf = l_cdf; else T.cdf = gcdf;
    } else {
        // many classes: parameters stay in global memory; the per-population weights of up to 16 birth classes
        // are still prepared once per block (more birth classes are rejected by the host)
        for (int i = threadIdx.x; i < CB * S && i < 16 * 64; i += EB) g_wtr[i] = p.cb_b[i / S] * p.cb_sigma[i] * (double)Sus[i % S] * F;
        __syncthreads();
        for (int cb = threadIdx.x; cb < CB && cb < 16; cb += EB) {
            double r = 0.0;
            for (int sn = 0; sn < S; ++sn) r += g_wtr[cb * S + sn];
            g_rtr[cb] = r;
            g_rmig[cb] = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] : 0.0;
        }
        T.c_d = p.c_d; T.c_s = p.c_s; T.c_tm = p.c_tm; T.c_bidx = p.c_bidx; T.c_stype = p.c_stype;
        T.rtr = g_rtr; T.wtr = g_wtr; T.rmig = g_rmig; T.mutcum = a.mutcum; T.cdf = gcdf;
    }
    WaveStage *stage = &stage_s;
    if (threadIdx.x == 0) stage->n = 0;
    __syncthreads();
    // per-thread tallies: [0..5] event counters, [6] delta of totalInfectious[pn], [7] the same as the check books it,
    // [8..11] susceptible deltas of the first four groups (all compartments of a population add to the same few addresses:
    // no atomics per event)
    int64_t cnt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int L = threadIdx.x;
    const int32_t *Irow = a.I + ((int64_t)rep * P + pn) * H;
    int32_t *dCrow = a.dChk + ((int64_t)rep * P + pn) * H, *dArow = a.dApp + ((int64_t)rep * P + pn) * H;
    const bool dense = !a.sparse;
    const TauEnv E = tau_env(a, rep, pn);
    // `ok` is read and cleared at device scope: the XCDs' L2 caches are not coherent with each other for plain accesses, a
    // wavefront on another XCD would never see the flag.  The load is issued one round ahead of its use.
    int okv = __hip_atomic_load(&a.ok[rep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t kstep = (int64_t)split * EB, kfirst = (int64_t)sub * EB;
    int round = 0;
    EVPROF(0);
    // The rounds are staged in chunks (below); the first chunks are short (1, 1, 2, 4 rounds, then VGX_EV_CHUNK): a try that is
    // going to be rejected usually meets its first failing compartment within the first rounds of SOME wavefront, and the others
    // notice at their next chunk boundary — with full chunks from the start a rejected try cost 0.2 ms of this kernel, two
    // thirds of an accepted one.
    int sb = sb0 - sbstep;
    int64_t n = 0, kc = 0;
    const int64_t *qsrc = nullptr;
    bool shard_open = false;
    int chunk_no = 0;
    for (;;) {
    int nr = 0;                     // rounds staged in s_q / s_I / s_c
    if (okv == 0) break;
    {
        if (!shard_open || kc >= n) {   // the next shard of the block with entries for it
            shard_open = false;
            for (sb += sbstep; sb < shards_pop; sb += sbstep) {
                n = FRONT ? front_n : (int64_t)qn_pop[sb];
                if (kfirst >= n) continue;
                if (n > scap) {   // the shard overflowed: compartments were lost, the host enlarges the queue and the same try runs again
                    if (threadIdx.x == 0) atomicOr(&a.grow[rep], 8);
                    continue;
                }
                shard_open = true;
                break;
            }
            if (!shard_open) break;
            qsrc = FRONT ? a.front + ((int64_t)rep * P + pn) * scap : a.q + (int64_t)rep * a.q_cap + ((int64_t)pn * shards_pop + sb) * scap;
            kc = kfirst;
        }
        // The queue entries of the next VGX_EV_CHUNK rounds and, from them, the compartments' counts (dependent, scattered): two
        // bursts of unconditional loads (indices clamped), parked in the wavefront's LDS stage.  One load per round issued
        // "ahead" does not work here: vmcnt counts loads and stores together and the round's body has stores and atomics, so
        // the compiler has to wait for everything (vmcnt(0)) at each use — every round paid a full trip to HBM.
        // `ok` rides along (a device-scope load goes past the L2: a few microseconds): a lost try is noticed within a chunk.
        okv = __hip_atomic_load(&a.ok[rep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int want = VGX_EV_SHORT_CHUNKS ? (chunk_no < 2 ? 1 : chunk_no == 2 ? 2 : chunk_no == 3 ? 4 : VGX_EV_CHUNK) : VGX_EV_CHUNK;
        chunk_no += 1;
        int64_t qv[VGX_EV_CHUNK];
        int32_t Iv[VGX_EV_CHUNK], Cv[VGX_EV_CHUNK];
#pragma unroll
        for (int j = 0; j < VGX_EV_CHUNK; ++j) {
            const int64_t kk = kc + (int64_t)j * kstep + L;
            qv[j] = -1;
            if (j < want) qv[j] = qsrc[kk < n ? kk : n - 1];
            if (kk >= n) qv[j] = -1;   // (nothing there)
        }
#ifdef VGX_PROFILE
        { int t_; asm volatile("v_mov_b32 %0, %1" : "=v"(t_) : "v"((int)qv[VGX_EV_CHUNK - 1])); asm volatile("" :: "v"(t_)); }
        { const long long t_ = clock64(); pacc[12] += t_ - pt; pt = t_; }
#endif
#pragma unroll
        for (int j = 0; j < VGX_EV_CHUNK; ++j) {
            const int hh = min((int)(qv[j] & 0x7FFFFFFFll), H - 1);   // (a bad entry must not fault)
            Iv[j] = 0; Cv[j] = 0;
            if (j < want) {
                Iv[j] = Irow[hh];
                Cv[j] = many_cls ? p.cls[hh] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < VGX_EV_CHUNK; ++j) {
            s_q[j * EB + L] = qv[j]; s_I[j * EB + L] = Iv[j];
            if (many_cls) s_c[j * EB + L] = Cv[j];
        }
        const int64_t left = (n - kc + kstep - 1) / kstep;     // rounds of this shard from kc on
        nr = left < want ? (int)left : want;
        kc += (int64_t)nr * kstep;
    }
#ifdef VGX_PROFILE
    EVPROF(1);
#endif
    for (int j = 0; j < nr; ++j, ++round) {
        if (okv == 0) break;   // (the try is already lost, nothing of it counts: vgx_tau_decide_kernel)
        const int64_t qe = s_q[j * EB + L];
        const int32_t I_now = s_I[j * EB + L];
        const int cls_now = many_cls ? s_c[j * EB + L] : 0;
#ifdef VGX_PROFILE
        prounds += 1;
#endif
        int h = 0;
        int64_t v = 0;
        bool below = false;   // below zero on its own (sparse mode): looked at by the whole wavefront, see below
        bool isbig = false;
        if (qe >= 0 && (int)(qe & 0x7FFFFFFFll) < H) {
            h = (int)(qe & 0x7FFFFFFFll);
            const int64_t Ih = (int64_t)I_now;
            int64_t oc = 0, oa = 0;
#ifdef VGX_PROFILE
            const int r = tau_cell_events<FRONT ? 2 : 0, TABS>(a, T, E, rep, pn, h, tau, Ih, (uint32_t)(qe >> 32) & 255u, oc, oa, cnt, stage, sS, -1, cls_now, pacc + 6);
#else
            const int r = tau_cell_events<FRONT ? 2 : 0, TABS>(a, T, E, rep, pn, h, tau, Ih, (uint32_t)(qe >> 32) & 255u, oc, oa, cnt, stage, sS, -1, cls_now);
#endif
            if (FRONT) {
                if (r == 1) { v = Ih + oc; below = v < 0; }
            } else {
            isbig = r == 2;   // many events: a group of lanes draws its channels one by one (vgx_tau_draw_big_kernel)
            if (dense) { dCrow[h] = (int32_t)oc; dArow[h] = (int32_t)oa; }
            else if (oa != 0) tau_list_add(a, stage, rep, (int64_t)pn * H + h, oa, true);
            if (r == 1) {
                v = Ih + oc;
                if (v < 0 && !dense) below = true;
                else tau_own_check(a, rep, pn, h, v);
            }
            }
        }
        // A compartment below zero on its own: do the mutants of its neighbours rescue it (pyx:2522-2528 look at the sum)?  Their
        // streams depend on nothing but the compartment, the step and the try, so the wavefront draws them here and now, one
        // single-site neighbour per lane (no bookkeeping: tau_cell_events<true>), and a definite failure ends the try for
        // everybody at once: the other wavefronts stop before their next round.  A neighbour that is drawn channel by channel
        // (vgx_tau_draw_big_kernel) leaves the question to the list of arrivals (vgx_tau_arrivals_kernel).
        // the compartments for vgx_tau_draw_big_kernel: ONE reservation per wavefront in their list (an atomic per compartment
        // on the one counter is worked off one after the other by the memory side: in a large epidemic on a small model, where
        // every compartment is of this kind, that was the whole kernel)
        const unsigned long long bigm = __ballot(isbig);
        if (__builtin_expect(bigm != 0, 0)) {
            unsigned long long base = 0;
            if (L == __ffsll((long long)bigm) - 1) base = atomicAdd(&a.big_n[rep], (unsigned long long)__popcll(bigm));
            base = (unsigned long long)bcast_i64((int64_t)base, __ffsll((long long)bigm) - 1);
            if (isbig) {
                const unsigned long long slot = base + (unsigned long long)__popcll(bigm & ((1ull << L) - 1ull));
                if ((int64_t)slot < a.big_cap) a.big[(int64_t)rep * a.big_cap + (int64_t)slot] = (int64_t)pn * H + h;
                else atomicOr(&a.grow[rep], 4);   // list full: the host enlarges it and the same try runs again
            }
        }
        unsigned long long todo = __ballot(below);
        EVPROF(2);
        while (__builtin_expect(todo != 0, 0)) {   // (rare: keeps this code out of the round loop's way in the instruction cache)
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int hs = __builtin_amdgcn_readlane(h, src);
            int64_t arr = 0;
            bool open = false;
            for (int i = L; i < 3 * p.sites; i += EB) {
                const int nb = hs ^ ((i % 3 + 1) << (2 * (p.sites - i / 3 - 1)));
                const int64_t In = (int64_t)Irow[nb];
                if (In <= 0) continue;
                int64_t kk = 0, dummy = 0;
                if (tau_cell_events<1, TABS>(a, T, E, rep, pn, nb, tau, In, tau_bucket(a, E, pn, nb), kk, dummy, cnt, nullptr, nullptr, hs) == 2) open = true;
                arr += kk;
            }
            for (int o = 32; o > 0; o >>= 1) arr += __shfl_down(arr, o);
            arr = bcast_i64(arr, 0);
            open = __any(open);
            if (L == src) {
                if (open) { if (!FRONT) tau_own_check(a, rep, pn, h, v); }   // (front pass: left to the try proper)
                else if (v + arr < 0) atomicAnd(&a.ok[rep], 0);
            }
        }
        // the staged moves go out when the next round might not fit (a round adds at most one own change per lane plus its
        // mutants and migrants; what does not fit goes to the list entry by entry)
        // (LDS only: a full fence would also wait for the loads issued ahead and for this round's global atomics)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        EVPROF(3);
        if (stage->n > VGX_WSTAGE - 96) tau_stage_flush(a, stage, rep);
        EVPROF(4);
    }   // rounds
    }   // chunks
    // a lost try: nothing of it counts (vgx_tau_decide_kernel discards the tallies, the list and the deltas), so a wavefront that
    // leaves early adds nothing — its atomics on the few shared addresses would be worked off one after the other
    if (okv == 0 || FRONT) return;
    tau_stage_flush(a, stage, rep);   // what is left
    // the event counters go to the population's own slots (vgx_tau_decide_kernel folds them): all wavefronts of a launch adding to
    // the same six addresses were worked off one after the other by the memory side, a quarter of an accepted try's kernel time
    unsigned long long *ct = a.cnt_pop + ((int64_t)rep * P + pn) * 8;
    for (int i = 0; i < 12; ++i) {   // wave-level sums, then one global atomic per tally
        long long v = cnt[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (threadIdx.x == 0 && v != 0) {
            if (i < 6) atomicAdd(&ct[i], (unsigned long long)v);
            else if (i == 6) atomicAdd((unsigned long long *)&a.dTot[(int64_t)rep * P + pn], (unsigned long long)v);
            else if (i == 7) atomicAdd((unsigned long long *)&a.dChkTot[(int64_t)rep * P + pn], (unsigned long long)v);
            else atomicAdd(&sS[i - 8], (unsigned long long)v);
        }
    }
    WSYNC();   // (one wavefront per block: no barrier instruction, and above all no wait for the global atomics just issued —
               //  all blocks add to the same few addresses, their acknowledgements were a fifth of a wavefront's life)
    if (threadIdx.x < S && sS[threadIdx.x])
        atomicAdd((unsigned long long *)&a.dSi[((int64_t)rep * P + pn) * S + threadIdx.x], sS[threadIdx.x]);
#ifdef VGX_PROFILE
    EVPROF(5);
    if (threadIdx.x == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&vgx_tau_ev_prof[i], (unsigned long long)pacc[i]);
        for (int i = 7; i < 12; ++i) atomicAdd(&vgx_tau_ev_prof[i], (unsigned long long)pacc[i]);
        atomicAdd(&vgx_tau_ev_prof[14], (unsigned long long)pacc[12]);   // first burst (queue entries + ok) of "wait for the count"
        atomicAdd(&vgx_tau_ev_prof[12], (unsigned long long)prounds);
        atomicAdd(&vgx_tau_ev_prof[13], 1ull);
    }
#endif
}


// Compartments that expect many events in this leap (>= a.big_lam; the events kernel lists them): every channel
// gets its own Poisson draw, as in the reference (pyx:2464-2520), with the channels of one compartment spread over the
// lanes of a wavefront.  The single draw + split of tau_cell_events walks through the events one by one, which a
// compartment with 10^5 hosts would do for tens of thousands of events on one lane (natural epidemics: most hosts carry a
// few haplotypes); the joint law of the channel counts is the same (Poisson splitting).  A channel's stream is keyed by
// (compartment, channel), so the result does not depend on the lane mapping.  grid = (VGX_BIG_BLOCKS, R), 4 waves a block.
// Immunity transitions (pyx:2479-2487 / 2554-2562): P*S*S slots per replicate, one thread each: the blocks of
// vgx_tau_draw_big_kernel beyond VGX_BIG_BLOCKS (a launch of its own was 3 % of a small model's step).
static __device__ __forceinline__ void tau_suscep_draw(const VgxTauArgs &a, int rep, int idx) {
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S;
    if (idx >= P * S * S) return;
    int pn = idx / (S * S), rest = idx % (S * S), ssn = rest / S, tsn = rest % S;
    if (ssn == tsn) return;
    const int64_t *Sus = a.S + (int64_t)rep * P * S;
    int64_t *dS = a.dSi + (int64_t)rep * P * S;
    double lam = p.suscepTransition[ssn * S + tsn] * (double)Sus[pn * S + ssn] * a.tau[rep];
    if (!(lam > 0.0)) return;
    TauRng g;
    g.init((uint64_t)a.seeds[rep], (uint32_t)a.attempt[rep], ((uint64_t)1 << 62) + (uint64_t)idx, (uint32_t)a.step[rep], (uint32_t)a.retry[rep]);
    int64_t k = tau_poisson(g, lam);
    if (k == 0) return;
    atomicAdd((unsigned long long *)&dS[pn * S + tsn], (unsigned long long)k);
    atomicAdd((unsigned long long *)&dS[pn * S + ssn], (unsigned long long)(-k));
    atomicAdd((unsigned long long *)&a.cnt_try[(int64_t)rep * 8 + 4], (unsigned long long)k);
    tau_row(a, rep, k, 4, ssn, pn, tsn, 0);
}

#define VGX_BIG_BLOCKS 512    // four wavefronts each: what the chip holds at two wavefronts per SIMD (blocks without work leave at once)
#define VGX_BIG_LT 2048       // block-local sums kept in LDS: P * (S + 2) + 5 entries (else straight to global memory)
// A compartment is worked on by a GROUP of 16, 32 or 64 lanes (the smallest that holds its channels, or 64): with few channels
// — small models, where such compartments are the whole epidemic — a wavefront draws four or two compartments at a time.  The
// lanes first work out their channel's mean, then ALL draw through one call of the sampler (five inlined copies, one per kind
// of channel, ran one after the other), then book the result.
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_draw_big_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    if (blockIdx.x >= VGX_BIG_BLOCKS) {   // the susceptible compartments' immunity transitions
        tau_suscep_draw(a, rep, (int)(blockIdx.x - VGX_BIG_BLOCKS) * TB + (int)threadIdx.x);
        return;
    }
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H, sites = p.sites, CB = p.CB;
    const int lane = threadIdx.x & 63;
    unsigned long long n = a.big_n[rep];
    if ((int64_t)n > a.big_cap) n = (unsigned long long)a.big_cap;
    if (n == 0) return;
#ifdef VGX_PROFILE
    long long bp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt = clock64();
#define BPROF(i) do { const long long t_ = clock64(); bp[i] += t_ - bt; bt = t_; } while (0)
#else
#define BPROF(i)
#endif
    const double tau = a.tau[rep];
    const uint64_t seed = (uint64_t)a.seeds[rep];
    const uint32_t att = (uint32_t)a.attempt[rep], step = (uint32_t)a.step[rep], retry = (uint32_t)a.retry[rep];
    int64_t *dS = a.dSi + (int64_t)rep * P * S;
    const int n_mut = 3 * sites, nchan_max = 2 + S + n_mut + (a.has_mig ? P * S : 0);
    const int G = nchan_max <= 16 ? 16 : (nchan_max <= 32 ? 32 : 64), per_wave = 64 / G, gl = lane & (G - 1);
    long long tot[5] = {0, 0, 0, 0, 0};   // group leaders: their tallies (one atomic each at the end, not one per compartment)
    // the moves are staged per wavefront (one reservation in the global list per ~128 entries instead of one per entry: every
    // channel of a large compartment has events, and all wavefronts of a small model append to the same few shards)
    // Sums per population (susceptible deltas, the two infectious totals) and the counters are collected per BLOCK in LDS and
    // added to global memory once at the end: every compartment of a population adds to the same few addresses, and atomics
    // on one address are worked off one after the other by the memory side — with eight populations that was half the kernel.
    __shared__ unsigned long long lt[VGX_BIG_LT];
    const int lt_n = P * (S + 2) + 5;
    const bool use_lt = lt_n <= VGX_BIG_LT;
    if (use_lt) {
        for (int i = threadIdx.x; i < lt_n; i += TB) lt[i] = 0;
        __syncthreads();
    }
    unsigned long long *ct = (unsigned long long *)&a.cnt_try[(int64_t)rep * 8];
    auto add_dS = [&](int idx, long long v) {
        if (use_lt) atomicAdd(&lt[idx], (unsigned long long)v); else atomicAdd((unsigned long long *)&dS[idx], (unsigned long long)v);
    };
    auto add_tot = [&](int pn_, long long v) {
        if (use_lt) atomicAdd(&lt[P * S + pn_], (unsigned long long)v); else atomicAdd((unsigned long long *)&a.dTot[(int64_t)rep * P + pn_], (unsigned long long)v);
    };
    auto add_chk = [&](int pn_, long long v) {
        if (use_lt) atomicAdd(&lt[P * S + P + pn_], (unsigned long long)v); else atomicAdd((unsigned long long *)&a.dChkTot[(int64_t)rep * P + pn_], (unsigned long long)v);
    };
    __shared__ WaveStage stages[TB / 64];
    WaveStage *stage = &stages[threadIdx.x >> 6];
    if (lane == 0) stage->n = 0;
    WSYNC();
    const unsigned long long wave = (unsigned long long)blockIdx.x * (TB / 64) + (threadIdx.x >> 6), nwaves = (unsigned long long)VGX_BIG_BLOCKS * (TB / 64);   // (the blocks beyond are the immunity transitions')
    for (unsigned long long e0 = wave * per_wave; e0 < n; e0 += nwaves * per_wave) {
        const unsigned long long e = e0 + (unsigned long long)(lane / G);
        const bool have = e < n;
        const int64_t cell = have ? a.big[(int64_t)rep * a.big_cap + (int64_t)e] : 0;
        const int pn = (int)(cell / H), hn = (int)(cell - (int64_t)pn * H);
        const double Ih = (double)a.I[(int64_t)rep * P * H + cell];
        const int c = (p.C == 1) ? 0 : p.cls[hn];
        const int cb = p.c_bidx[c], st = p.c_stype[c];
        const double F = a.F[(int64_t)rep * P + pn];
        const int64_t *Sus = a.S + (int64_t)rep * P * S + (int64_t)pn * S;
        const double *cdf = a.migcdf + (((int64_t)rep * P + pn) * CB + cb) * (int64_t)P * S;
        const double r_mig = a.has_mig ? a.Gout[((int64_t)rep * P + pn) * CB + cb] * p.cb_b[cb] * p.mig[(int64_t)pn * P + pn] * Ih : 0.0;
        const int n_mig = (r_mig > 0.0) ? P * S : 0, nchan = have ? 2 + S + n_mut + n_mig : 0;
        int64_t births = 0, rec = 0, samp = 0, mut_done = 0, migrants = 0;
#ifdef VGX_PROFILE
        { int t_; asm volatile("v_mov_b32 %0, %1" : "=v"(t_) : "v"((int)(r_mig != 0.0) + cb + st + (int)F)); asm volatile("" :: "v"(t_)); }
#endif
        BPROF(0);
        for (int ch = gl; ch < nchan; ch += G) {
            // the channel's mean number of events in this leap ...
            double lam;
            if (ch == 0) lam = p.c_d[c] * Ih * tau;                                          // recovery, pyx:2386
            else if (ch == 1) lam = p.c_s[c] * Ih * p.sampMult[pn] * tau;                    // sampling, pyx:2392
            else if (ch < 2 + S) {                                                           // transmission to group sn, pyx:2412-2414
                const int sn = ch - 2;
                lam = p.cb_b[cb] * p.cb_sigma[cb * S + sn] * (double)Sus[sn] * F * Ih * tau;
            } else if (ch < 2 + S + n_mut) {                                                 // mutation, pyx:2400-2401
                const int m = ch - 2 - S, ss = m / 3, ii = m % 3;
                double rate;
                if (a.mut_uniform) rate = a.mutp[ss][ii];
                else {
                    const double *hm = p.hapMutType + ((int64_t)hn * sites + ss) * 3;
                    rate = p.mRate[(int64_t)hn * sites + ss] * hm[ii] / (hm[0] + hm[1] + hm[2]);
                }
                lam = rate * Ih * tau;
            } else {                                                                         // migration, pyx:2366-2367
                const int j = ch - 2 - S - n_mut, tp = j / S;
                const double wj = cdf[j] - (j > 0 ? cdf[j - 1] : 0.0);
                lam = (wj > 0.0 && tp != pn) ? wj / cdf[P * S - 1] * r_mig * tau : 0.0;
            }
#ifdef VGX_PROFILE
            { int t_; asm volatile("v_mov_b32 %0, %1" : "=v"(t_) : "v"((int)lam)); asm volatile("" :: "v"(t_)); }
#endif
            BPROF(1);
            // ... its draw (a mean of zero draws nothing and uses no random number) ...
            TauRng g;
            g.init(seed, att, (uint64_t)cell | ((uint64_t)(ch + 1) << 40), step, retry);
            const int64_t k = tau_poisson(g, lam);
            BPROF(2);
            if (k == 0) continue;
            // ... and its books
            if (ch == 0) { rec = k; tau_row(a, rep, k, 1, hn, pn, st, 0); }
            else if (ch == 1) { samp = k; tau_row(a, rep, k, 2, hn, pn, st, 0); }
            else if (ch < 2 + S) {
                const int sn = ch - 2;
                births += k;
                add_dS(pn * S + sn, -k);
                tau_row(a, rep, k, 0, hn, pn, sn, 0);
            } else if (ch < 2 + S + n_mut) {
                const int m = ch - 2 - S;
                const int nh = tau_mutate(sites, hn, m / 3, m % 3);
                mut_done += k;
                tau_list_add(a, stage, rep, (int64_t)pn * H + nh, k, false);
                tau_row(a, rep, k, 3, hn, pn, nh, 0);
            } else {
                const int j = ch - 2 - S - n_mut, tp = j / S, ts = j % S;
                migrants += k;
                tau_list_add(a, stage, rep, (int64_t)tp * H + hn, k, true);
                add_dS(tp * S + ts, -k);
                add_tot(tp, k);
                tau_row(a, rep, k, 5, hn, pn, ts, tp);
            }
        }
        BPROF(3);
        long long v[5] = {births, rec, samp, mut_done, migrants};
        for (int i = 0; i < 5; ++i)
            for (int o = G >> 1; o > 0; o >>= 1) v[i] += __shfl_down(v[i], o, G);
        if (gl == 0 && have) {
            const int64_t own = v[0] - v[1] - v[2] - v[3];
            const int64_t off = (int64_t)rep * P * H + cell;
            if (!a.sparse) {
                a.dChk[off] = (int32_t)(own + v[4]);   // pyx:2473: migrants are booked on their source here
                a.dApp[off] = (int32_t)own;
            } else if (own != 0) {
                tau_list_add(a, stage, rep, cell, own, true);
            }
            tau_own_check(a, rep, pn, hn, (int64_t)a.I[off] + own + v[4]);
            if (v[0] - v[1] - v[2] + v[4]) add_chk(pn, v[0] - v[1] - v[2] + v[4]);
            for (int i = 0; i < 5; ++i) tot[i] += v[i];
            if (v[1] + v[2]) add_dS(pn * S + st, v[1] + v[2]);
            if (v[0] - v[1] - v[2]) add_tot(pn, v[0] - v[1] - v[2]);
        }
        WSYNC();
        BPROF(4);
        if (stage->n > VGX_WSTAGE - 96) tau_stage_flush(a, stage, rep);   // (what does not fit goes to the list entry by entry)
        BPROF(5);
#ifdef VGX_PROFILE
        bp[7] += 1;
#endif
    }
    tau_stage_flush(a, stage, rep);
#ifdef VGX_PROFILE
    BPROF(6);
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&vgx_tau_big_prof[i], (unsigned long long)bp[i]);
#endif
#undef BPROF
    const int slot[5] = {0, 1, 2, 3, 5};
    if (gl == 0) {
        for (int i = 0; i < 5; ++i)
            if (tot[i]) { if (use_lt) atomicAdd(&lt[P * (S + 2) + i], (unsigned long long)tot[i]); else atomicAdd(&ct[slot[i]], (unsigned long long)tot[i]); }
    }
    if (use_lt) {
        __syncthreads();
        for (int i = threadIdx.x; i < lt_n; i += TB) {
            const unsigned long long v = lt[i];
            if (v == 0) continue;
            if (i < P * S) atomicAdd((unsigned long long *)&dS[i], v);
            else if (i < P * S + P) atomicAdd((unsigned long long *)&a.dTot[(int64_t)rep * P + (i - P * S)], v);
            else if (i < P * S + 2 * P) atomicAdd((unsigned long long *)&a.dChkTot[(int64_t)rep * P + (i - P * S - P)], v);
            else atomicAdd(&ct[slot[i - P * (S + 2)]], v);
        }
    }
}

// Adds the appended incoming individuals to the delta arrays.  grid = (inc_shards, R, VGX_INC_SHARDS / inc_shards):
// the blocks of one shard take interleaved 64-entry slices of it, so a few long shards (small models with large
// epidemics) are worked off by as many wavefronts as many short ones.  vgx_tau_decide_kernel empties the list.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_scatter_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y, shard = blockIdx.x;
    if (!a.active[rep] || a.accepted[rep]) return;
    const int64_t PH = (int64_t)a.p.P * a.p.H;
    const int64_t scap = a.inc_cap / a.inc_shards;
    unsigned long long n = a.inc_n[(int64_t)rep * VGX_INC_SHARDS + shard];
    if ((int64_t)n > scap) {  // shard overflow: the step cannot be validated
        if (threadIdx.x == 0 && blockIdx.z == 0) atomicOr(&a.grow[rep], 1);
        n = (unsigned long long)scap;
    }
    const int64_t *lst = a.inc + (int64_t)rep * a.inc_cap + (int64_t)shard * scap;
    for (unsigned long long i = (unsigned long long)blockIdx.z * 64 + threadIdx.x; i < n; i += (unsigned long long)gridDim.z * 64) {
        int64_t e = lst[i];
        int64_t cell = tau_entry_cell(e);
        int k = tau_entry_delta(e);
        atomicAdd(&a.dApp[(int64_t)rep * PH + cell], k);
        if (!(e & VGX_INC_APPLIED)) {
            // upper bound of the check (pyx:2522-2528): arrivals only increase the compartment's delta, so whoever adds
            // last sees the final value (the compartment's own delta was stored by the draw kernels before this one)
            const int old = atomicAdd(&a.dChk[(int64_t)rep * PH + cell], k);
            if ((int64_t)a.I[(int64_t)rep * PH + cell] + (int64_t)old + (int64_t)k > a.p.sizes[cell / a.p.H]) a.ok[rep] = 0;
        }
    }
}

// The listed compartments after the arrivals (vgx_tau_scatter_kernel): still below zero -> the try is rejected.  Also the
// susceptible compartments' bounds.  grid = (32, R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_suspect_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S;
    const int64_t PH = (int64_t)P * p.H;
    unsigned long long n = a.suspect_n[rep];
    bool bad = false;
    if ((int64_t)n > a.suspect_cap) n = (unsigned long long)a.suspect_cap;   // overflow: the host runs the dense check instead
    for (unsigned long long i = (unsigned long long)blockIdx.x * TB + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * TB) {
        const int64_t cell = a.suspect[((int64_t)rep * a.suspect_cap + (int64_t)i) * 2];
        if ((int64_t)a.I[(int64_t)rep * PH + cell] + (int64_t)a.dChk[(int64_t)rep * PH + cell] < 0) bad = true;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < P * S; i += TB) {
            const int64_t v = a.dSi[(int64_t)rep * P * S + i] + a.S[(int64_t)rep * P * S + i];
            if (v < 0 || v > p.sizes[i / S]) bad = true;
        }
    if (__any(bad) && (threadIdx.x & 63) == 0) a.ok[rep] = 0;
}

// ---- sparse mode: the two kernels that stand in for vgx_tau_scatter_kernel / vgx_tau_suspect_kernel --------------------
// Adds the mutants that arrive in a listed compartment (found below zero on its own) to its sum in the hash table.
// grid = (inc_shards, R, VGX_INC_SHARDS / inc_shards) as for the scatter kernel.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_arrivals_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y, shard = blockIdx.x;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    const int64_t scap = a.inc_cap / a.inc_shards;
    unsigned long long n = a.inc_n[(int64_t)rep * VGX_INC_SHARDS + shard];
    if ((int64_t)n > scap) {  // shard overflow: entries were lost, nothing of this try counts
        if (threadIdx.x == 0 && blockIdx.z == 0) atomicOr(&a.grow[rep], 1);
        n = (unsigned long long)scap;
    }
    if (a.suspect_n[rep] == 0) return;   // nobody to rescue
    const int64_t *lst = a.inc + (int64_t)rep * a.inc_cap + (int64_t)shard * scap;
    for (unsigned long long i = (unsigned long long)blockIdx.z * 64 + threadIdx.x; i < n; i += (unsigned long long)gridDim.z * 64) {
        const int64_t e = lst[i];
        if (e & VGX_INC_APPLIED) continue;   // migrants and own changes are not booked on the compartment by the check (pyx:2473)
        const int64_t slot = tau_st_find(a, rep, tau_entry_cell(e));
        if (slot >= 0) atomicAdd((unsigned long long *)&a.st_val[(int64_t)rep * a.st_size + slot], (unsigned long long)(long long)tau_entry_delta(e));
    }
}

// The listed compartments with their arrivals: still below zero -> the try is rejected (pyx:2522-2528).  The susceptible
// compartments' bounds.  The upper bound of the infectious compartments, per population: the compartments of a population
// that passed the lower bound sum to totalInfectious + (sum of the deltas as the check books them); if that is within the
// population size so is each of them.  Otherwise the same try is run again with dense delta arrays (grow bit 1 << 1).
// grid = (32, R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_verdict_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep] || tau_gate_closed(a, rep)) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S;
    unsigned long long n = a.suspect_n[rep];
    bool bad = false, redo = false;
    if ((int64_t)n > a.suspect_cap) { n = (unsigned long long)a.suspect_cap; redo = true; }   // more than the table holds
    for (unsigned long long i = (unsigned long long)blockIdx.x * TB + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * TB) {
        const int64_t cell = a.suspect[((int64_t)rep * a.suspect_cap + (int64_t)i) * 2];
        const int64_t slot = tau_st_find(a, rep, cell);
        if (slot < 0 || a.st_val[(int64_t)rep * a.st_size + slot] < 0) bad = true;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < P * S; i += TB) {
            const int64_t v = a.dSi[(int64_t)rep * P * S + i] + a.S[(int64_t)rep * P * S + i];
            if (v < 0 || v > p.sizes[i / S]) bad = true;
        }
        for (int pn = threadIdx.x; pn < P; pn += TB)
            if (a.totInf[(int64_t)rep * P + pn] + a.dChkTot[(int64_t)rep * P + pn] > p.sizes[pn]) redo = true;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) a.ok[rep] = 0;
    if (__any(redo) && (threadIdx.x & 63) == 0) atomicOr(&a.grow[rep], 2);
}

// UpdateCompartmentCounts_tau of the sparse mode: the list of the accepted try is added to the compartments (one atomic per
// entry; a rejected try leaves nothing behind), the small per-population accumulators are applied / cleared as in
// vgx_tau_commit_kernel.  grid = (inc_shards, R, VGX_INC_SHARDS / inc_shards).
extern "C" __global__ void __launch_bounds__(64) vgx_tau_apply_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y, shard = blockIdx.x;
    if (!a.deciding[rep] || (a.gate == 2 && a.spec[rep] < 2)) return;   // (a round whose front passes found no try to run: nothing to apply or clear)
    const int P = a.p.P, S = a.p.S;
    const bool acc = a.accepted[rep] && !a.error[rep];
    if (acc) {
        const int64_t PH = (int64_t)P * a.p.H;
        const int64_t scap = a.inc_cap / a.inc_shards;
        unsigned long long n = a.inc_n[(int64_t)rep * VGX_INC_SHARDS + shard];
        if ((int64_t)n > scap) n = (unsigned long long)scap;   // (an overflowing try is never accepted)
        const int64_t *lst = a.inc + (int64_t)rep * a.inc_cap + (int64_t)shard * scap;
        for (unsigned long long i = (unsigned long long)blockIdx.z * 64 + threadIdx.x; i < n; i += (unsigned long long)gridDim.z * 64) {
            const int64_t e = lst[i];
            atomicAdd(&a.I[(int64_t)rep * PH + tau_entry_cell(e)], tau_entry_delta(e));
        }
    }
    if (shard == 0 && blockIdx.z == 0) {
        for (int i = threadIdx.x; i < P * S; i += 64) {
            const int64_t off = (int64_t)rep * P * S + i;
            if (acc) a.S[off] += a.dSi[off];
            a.dSi[off] = 0;
        }
        for (int pn = threadIdx.x; pn < P; pn += 64) {
            const int64_t off = (int64_t)rep * P + pn;
            if (acc) a.totInf[off] += a.dTot[off];
            a.dTot[off] = 0;
            a.dChkTot[off] = 0;
        }
    }
}

// After vgx_tau_apply_kernel: the one-byte copy of every compartment the accepted try's list touched (same grid as the apply kernel;
// a compartment listed twice is written twice with the same, final, value).
extern "C" __global__ void __launch_bounds__(64) vgx_tau_sync8_kernel(VgxTauArgs a) {
    const int rep = blockIdx.y, shard = blockIdx.x;
    if (!a.deciding[rep] || !(a.accepted[rep] && !a.error[rep]) || (a.gate == 2 && a.spec[rep] < 2)) return;
    const int P = a.p.P, H = a.p.H;
    const int64_t PH = (int64_t)P * H;
    const int64_t scap = a.inc_cap / a.inc_shards;
    unsigned long long n = a.inc_n[(int64_t)rep * VGX_INC_SHARDS + shard];
    if ((int64_t)n > scap) n = (unsigned long long)scap;
    const int64_t *lst = a.inc + (int64_t)rep * a.inc_cap + (int64_t)shard * scap;
    for (unsigned long long i = (unsigned long long)blockIdx.z * 64 + threadIdx.x; i < n; i += (unsigned long long)gridDim.z * 64) {
        const int64_t cell = tau_entry_cell(lst[i]);
        const int32_t v = __hip_atomic_load(&a.I[(int64_t)rep * PH + cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.I8[(int64_t)rep * PH + cell] = (uint8_t)(v < 255 ? v : 255);
        unsigned int *tm = a.tmax8 + ((int64_t)rep * P + cell / H) * a.nt8 + ((cell % H) >> (2 * VGX_D8_LOW));
        const unsigned int b = (unsigned int)(v < 255 ? v : 255);
        if (b > *tm) atomicMax(tm, b);     // (never lowered between two conversions: an upper bound is all the drift pass needs)
    }
}

// Bounds check of GenerateEvents_tau (pyx:2522-2528) as one dense pass over all compartments: the fallback when the list of
// vgx_tau_suspect_kernel overflowed, and the validation mode (vgx_run_opts.reserved[1] = 1).  grid = (ceil(H/TB), P, R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_check_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.active[rep] || a.accepted[rep]) return;
    const VgxDevParams &p = a.p;
    const int P = p.P, S = p.S, H = p.H;
    const int h0 = (blockIdx.x * TB + threadIdx.x) * 4;   // four compartments per thread (16-byte loads)
    bool bad = false;
    if (h0 < H) {
        const int64_t off = (int64_t)rep * P * H + (int64_t)pn * H + h0;
        int d[4] = {0, 0, 0, 0};
        const bool full = h0 + 3 < H && (H & 3) == 0;
        if (full) { int4 x = *(const int4 *)(a.dChk + off); d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; }
        else for (int j = 0; j < 4; ++j) if (h0 + j < H) d[j] = a.dChk[off + j];
        if (d[0] | d[1] | d[2] | d[3]) {
            if (full) {   // one 16-byte load of the counts and one 16-byte clearing store for the four compartments
                const int4 x = *(const int4 *)(a.I + off);
                const int Iv[4] = {x.x, x.y, x.z, x.w};
                for (int j = 0; j < 4; ++j)
                    if (d[j] != 0) {
                        int64_t v = (int64_t)d[j] + (int64_t)Iv[j];
                        bad = bad || v < 0 || v > p.sizes[pn];
                    }
            } else {
                for (int j = 0; j < 4; ++j)
                    if (d[j] != 0) {
                        int64_t v = (int64_t)d[j] + (int64_t)a.I[off + j];
                        bad = bad || v < 0 || v > p.sizes[pn];
                    }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < S) {
        int64_t v = a.dSi[((int64_t)rep * P + pn) * S + threadIdx.x] + a.S[((int64_t)rep * P + pn) * S + threadIdx.x];
        if (v < 0 || v > p.sizes[pn]) bad = true;
    }
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) a.ok[rep] = 0;
    }
}

// After the check of one retry: accept, or halve tau and discard the tentative tallies (pyx:2316-2321).
// grid = R, block = 64.  `deciding` marks the replicates whose deltas the commit kernel must now handle.
// grow[rep] on entry: what the kernels of the try found out (1 = list of moves, 4 = list of large compartments, 8 = queue of
// drawing compartments overflowed: data were lost, nothing of the try counts; 2 = the sparse check cannot decide the upper
// bound); on exit: what the host has to do before the SAME try runs again (tau and the try index stay, the streams are keyed
// by the try index): 1 / 4 / 8 / 16 = enlarge that list (16: the multievent rows), 2 = run it with dense delta arrays; 0 = the
// try was accepted or rejected.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_decide_kernel(VgxTauArgs a) {
    const int rep = blockIdx.x;
    if (tau_gate_closed(a, rep)) return;     // (a round: this try's turn has passed; the flags the host reads stay the deciding try's)
    const bool live = a.active[rep] && !a.accepted[rep];
    int g = live ? a.grow[rep] : 0;
    if (live && a.mev_cap > 0 && (int64_t)a.mev_n[rep] > a.mev_cap) g |= 16;   // rows were lost: the host enlarges the buffer
    const bool ok = live && a.ok[rep];
    if (a.phase == 1 && ok) {   // the front pass alone, and it found no failure: nothing is touched, the try proper comes next
        if (threadIdx.x == 0 && a.host_flags) { a.host_flags[rep] = a.accepted[rep]; a.host_flags[a.R + rep] = 0; a.host_flags[2 * a.R + rep] = 1; }
        if (threadIdx.x == 0) a.spec[rep] = 1;
        return;
    }
    const int again = (g & 29) ? (g & 29) : (ok ? (g & 2) : 0);
    const bool accept = live && ok && again == 0;
    const bool front_alone = a.phase == 1;   // (rejected by its front pass alone: no tallies, and without the lists no queue either)
    if (live && !front_alone) {   // the events kernel's tallies, kept per population (vgx_tau_events_kernel's epilogue)
        const int P = a.p.P;
        unsigned long long part[6] = {0, 0, 0, 0, 0, 0};
        for (int pn = threadIdx.x; pn < P; pn += 64) {
            unsigned long long *cp = a.cnt_pop + ((int64_t)rep * P + pn) * 8;
            for (int i = 0; i < 6; ++i) { part[i] += cp[i]; cp[i] = 0; }
        }
        for (int i = 0; i < 6; ++i) {
            long long v = (long long)part[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (threadIdx.x == 0) a.cnt_try[(int64_t)rep * 8 + i] += v;
        }
    }
    __syncthreads();
    // the list of moves: applied already (dense mode: vgx_tau_scatter_kernel) or discarded; the sparse mode's accepted list
    // is applied after this kernel (vgx_tau_apply_kernel) and emptied by vgx_tau_finish_kernel
    if (live && (!accept || !a.sparse))
        for (int i = threadIdx.x; i < a.inc_shards; i += 64) a.inc_n[(int64_t)rep * VGX_INC_SHARDS + i] = 0;
    if (live) {  // the queue of the try has been worked off
        if (!front_alone || a.use_list)
            for (int64_t i = threadIdx.x; i < a.q_shards; i += 64) a.q_n[(int64_t)rep * a.q_shards + i] = 0;
        if (a.front_on)
            for (int i = threadIdx.x; i < a.p.P; i += 64) a.front_n[(int64_t)rep * a.p.P + i] = 0;
    }
    if (threadIdx.x != 0) return;
    // what the host looks at after every try, written to its (pinned) memory as well: no copy to wait for
    struct Mirror {
        const VgxTauArgs &a; int rep;
        __device__ ~Mirror() { if (a.host_flags) { a.host_flags[rep] = a.accepted[rep]; a.host_flags[a.R + rep] = a.grow[rep]; a.host_flags[2 * a.R + rep] = 0; } }
    } mirror{a, rep};
    a.deciding[rep] = 0;
    if (!live) return;
    a.big_n[rep] = 0;
    a.suspect_n[rep] = 0;
    a.deciding[rep] = 1;
    a.grow[rep] = again;
    if (again) {   // discarded like a rejected try, but tau and the try index stay
        a.spec[rep] = 2;
        a.ok[rep] = 1;
        for (int i = 0; i < 6; ++i) a.cnt_try[(int64_t)rep * 8 + i] = 0;
        a.mev_n[rep] = a.mev_base[rep];
        return;
    }
    if (accept) {
        a.accepted[rep] = 1;
        a.spec[rep] = 2;
        for (int i = 0; i < 6; ++i) { a.counters[(int64_t)rep * 8 + i] += a.cnt_try[(int64_t)rep * 8 + i]; }
        int64_t drawn = 0;
        for (int i = 0; i < 6; ++i) { drawn += a.cnt_try[(int64_t)rep * 8 + i]; a.cnt_try[(int64_t)rep * 8 + i] = 0; }
        a.counters[(int64_t)rep * 8 + 7] += drawn;  // total events drawn (throughput metric)
    } else {
        a.tau[rep] *= 0.5;
        a.retry[rep] += 1;
        a.ok[rep] = 1;
        for (int i = 0; i < 6; ++i) a.cnt_try[(int64_t)rep * 8 + i] = 0;
        a.mev_n[rep] = a.mev_base[rep];
        if (a.gate == 2) a.spec[rep] = 3;     // (a round: its try proper was rejected; the next round starts with front passes again)
        if (a.retry[rep] > 200) { a.accepted[rep] = 1; a.error[rep] = 5; a.spec[rep] = 2; }  // loop guard (tau underflow)
    }
}

// UpdateCompartmentCounts_tau for accepted replicates (I += dApp, S += dS, totals); the small per-population
// accumulators are cleared for accepted and rejected ones alike, and so are the entries of dApp that the try set.  grid = (ceil(H/TB), P, R).
extern "C" __global__ void __launch_bounds__(TB) vgx_tau_commit_kernel(VgxTauArgs a) {
    const int rep = blockIdx.z, pn = blockIdx.y;
    if (!a.deciding[rep]) return;
    const int P = a.p.P, S = a.p.S, H = a.p.H;
    const bool acc = a.accepted[rep] && !a.error[rep];
    const int h0 = (blockIdx.x * TB + threadIdx.x) * 4;   // four compartments per thread
    if (h0 < H && acc) {   // the accepted try only: a rejected one leaves nothing behind (the next try overwrites the deltas)
        const int64_t off = (int64_t)rep * P * H + (int64_t)pn * H + h0;
        if (h0 + 3 < H && (H & 3) == 0) {
            const int4 x = *(const int4 *)(a.dApp + off);
            if (x.x | x.y | x.z | x.w) {
                int4 v = *(const int4 *)(a.I + off);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                *(int4 *)(a.I + off) = v;
            }
        } else {
            for (int j = 0; j < 4; ++j)
                if (h0 + j < H) { const int32_t dA = a.dApp[off + j]; if (dA != 0) a.I[off + j] += dA; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < S) {
        int64_t off = ((int64_t)rep * P + pn) * S + threadIdx.x;
        if (acc) a.S[off] += a.dSi[off];
        a.dSi[off] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t off = (int64_t)rep * P + pn;
        if (acc) a.totInf[off] += a.dTot[off];
        a.dTot[off] = 0;
        a.dChkTot[off] = 0;
    }
}

// End of a step (pyx:2322-2329): globalInfectious, CheckLockdown for every population (contact density
// only: the step kernels rebuild what they need from it).  grid = R, block = 64.
extern "C" __global__ void __launch_bounds__(64) vgx_tau_finish_kernel(VgxTauArgs a) {
    const int rep = blockIdx.x;
    if (!a.active[rep]) return;
    if (a.gate == 3 && !a.accepted[rep]) return;     // (a round that did not end the step)
    const VgxDevParams &p = a.p;
    const int P = p.P;
    const int lane = threadIdx.x;
    long long part = 0;
    for (int pn = lane; pn < P; pn += 64) part += a.totInf[(int64_t)rep * P + pn];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    long long g = __shfl(part, 0);
    if (g != 0) {  // pyx:2326-2329: no lockdown check after extinction
        for (int pn = lane; pn < P; pn += 64) {
            int64_t ti = a.totInf[(int64_t)rep * P + pn];
            int32_t *lock = a.lockON + (int64_t)rep * P + pn;
            double *cd = a.cd + (int64_t)rep * P + pn;
            for (int pass = 0; pass < 2; ++pass) {
                bool flip = pass == 0 ? ((double)ti > p.startLD[pn] * (double)p.sizes[pn] && *lock == 0)
                                      : ((double)ti < p.endLD[pn] * (double)p.sizes[pn] && *lock == 1);
                if (flip) {
                    *cd = pass == 0 ? p.cdAfter[pn] : p.cdBefore[pn];
                    *lock = pass == 0 ? 1 : 0;
                    unsigned long long slot = atomicAdd(&a.loc_n[rep], 1ull);
                    if (slot < VGX_LOC_CAP) {
                        a.loc_rec[((int64_t)rep * VGX_LOC_CAP + slot) * 2 + 0] = pass == 0 ? 1 : 0;
                        a.loc_rec[((int64_t)rep * VGX_LOC_CAP + slot) * 2 + 1] = pn;
                        a.loc_time[(int64_t)rep * VGX_LOC_CAP + slot] = a.time_now[rep] + a.tau[rep];
                    } else {
                        a.error[rep] = 7;   // lockdown log full: the call fails like the direct path's (no silent truncation)
                    }
                    atomicAdd((unsigned long long *)&a.counters[(int64_t)rep * 8 + 6], 1ull);  // swapLockdown
                    a.eff_dirty[rep] = 1;
                }
            }
        }
    }
    __syncthreads();   // the lockdown records above use time_now + tau
    for (int i = lane; i < a.inc_shards; i += 64) a.inc_n[(int64_t)rep * VGX_INC_SHARDS + i] = 0;   // (sparse mode: applied by now)
    long long occ_sum = 0;
    if (a.occ_pop) {
        for (int pn = lane; pn < P; pn += 64) { occ_sum += (long long)a.occ_pop[(int64_t)rep * P + pn]; a.occ_pop[(int64_t)rep * P + pn] = 0; }
        for (int o = 32; o > 0; o >>= 1) occ_sum += __shfl_down(occ_sum, o);
    }
    if (lane == 0) {
        a.gI[rep] = g;
        // one packed record per replicate for the host, and the bookkeeping it used to upload before every step
        int64_t *o = a.res + (int64_t)rep * 16;
        o[0] = __double_as_longlong(a.tau[rep]);
        o[1] = g;
        for (int i = 0; i < 8; ++i) o[2 + i] = a.counters[(int64_t)rep * 8 + i];
        o[10] = (int64_t)a.mev_base[rep];        // multievent rows of this step: [o[10], o[11])
        o[11] = (int64_t)a.mev_n[rep];
        o[12] = a.error[rep];
        o[13] = (a.occ_pop && a.use8) ? occ_sum : -1;   // occupied compartments at the start of this step (drift pass on the bytes), else -1
        o[14] = a.retry[rep];                            // rejected tries of this step
        o[15] = (a.drift_sparse && a.d8s_bc) ? (int64_t)a.d8s_bc[(int64_t)rep * 8 + 3] : -1;   // compartments whose empty neighbours the sparse drift pass formed
        if (a.host_res)
            for (int i = 0; i < 15; ++i) a.host_res[(int64_t)rep * 16 + i] = o[i];   // the host's (pinned) copy
        a.mev_base[rep] = a.mev_n[rep];          // rows of the accepted step stay (pyx:2325)
        a.time_now[rep] += a.tau[rep];           // pyx:2322
        a.step[rep] += 1;
    }
}

// ---- launchers -----------------------------------------------------------------------------------
#define TAU_LAUNCH(name, grid, block)                                                            \
    extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_##name(const VgxTauArgs *a, \
                                                                           hipStream_t s) {      \
        hipLaunchKernelGGL(vgx_##name##_kernel, grid, block, 0, s, *a);                          \
        return hipGetLastError();                                                                \
    }
#define CELL_GRID dim3((unsigned)((a->p.H + 4 * TB - 1) / (4 * TB)), (unsigned)a->p.P, (unsigned)a->R)   /* 4 compartments per thread */
TAU_LAUNCH(tau_eff, dim3((unsigned)a->p.P, (unsigned)a->R), dim3(TB))
TAU_LAUNCH(tau_prep, dim3((unsigned)a->p.P, (unsigned)a->R), dim3(TB))
// blocks per (population, replicate) of the drift kernel vgxi_tau_drift launches last (= slots of VgxTauArgs.dS_part in use)
extern "C" __attribute__((visibility("hidden"))) int vgxi_tau_drift_blocks(const VgxTauArgs *a) {
    if (a->use8) return a->nt8;
    if (tau_drift_tiled_ok(a->p.sites, a->mut_uniform) && a->mutHi) {
        const int sites = a->p.sites, low = sites < VGX_DRIFT_LOW ? sites : VGX_DRIFT_LOW;
        const int64_t ntiles = a->p.H >> (2 * low);
        if (a->mutlow_fast) return (int)std::min<int64_t>(ntiles, std::max<int64_t>(1, 4096 / std::max<int64_t>(1, (int64_t)a->p.P * a->R)));
        return (int)ntiles;
    }
    const unsigned tiles = (unsigned)((a->p.H + TB - 1) / TB);
    return (int)(tiles < 32u ? tiles : 32u);
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_tau_conv8(const VgxTauArgs *a, hipStream_t s) {
    hipLaunchKernelGGL(vgx_tau_conv8_kernel, dim3((unsigned)((a->p.H + 4 * TB - 1) / (4 * TB)), (unsigned)a->p.P, (unsigned)a->R), dim3(TB), 0, s, *a);
    return hipGetLastError();
}
TAU_LAUNCH(tau_sync8, dim3((unsigned)a->inc_shards, (unsigned)a->R, (unsigned)(VGX_INC_SHARDS / a->inc_shards)), dim3(64))
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_tau_drift(const VgxTauArgs *a, hipStream_t s) {
    if (a->use8) {
        if (a->drift_sparse) {     // (the host: lists wanted, uniform migration — the column sums' pass writes the lists)
            hipError_t err = hipMemsetAsync(a->d8s_bc, 0, (size_t)a->R * 64, s);
            if (err != hipSuccess) return err;
        }
        if (a->has_mig) {     // (uniform: the condition of use8)
            hipLaunchKernelGGL(vgx_tau_colsum8_kernel, dim3((unsigned)((a->p.H + 4 * TB - 1) / (4 * TB)), (unsigned)a->R), dim3(TB), 0, s, *a);
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) return err;
        }
        if (a->drift_sparse) {
            hipError_t err;
            const dim3 greg((unsigned)(8 * ((a->p.P + 7) / 8) * ((a->occ_nreg + 3) / 4)), (unsigned)a->R);
            hipLaunchKernelGGL(vgx_tau_drift8s_prep_kernel, dim3((unsigned)a->p.P, (unsigned)a->R), dim3(D8_TB), 0, s, *a);
            const int low = a->p.sites < VGX_D8_LOW ? a->p.sites : VGX_D8_LOW;
            const size_t lds = (size_t)1 << (2 * low);
            err = hipFuncSetAttribute((const void *)vgx_tau_drift8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (err != hipSuccess) return err;
            hipLaunchKernelGGL(vgx_tau_drift8_kernel, dim3((unsigned)(8 * ((a->p.P + 7) / 8) * a->nt8), (unsigned)a->R), dim3(D8_TB), lds, s, *a);
            hipLaunchKernelGGL(vgx_tau_drift8s_heavy_kernel, greg, dim3(256), 0, s, *a);
            hipLaunchKernelGGL(vgx_tau_drift8s_ovf_kernel, dim3(VGX_D8S_OVF_BLOCKS, (unsigned)a->R), dim3(256), 0, s, *a);
            hipLaunchKernelGGL(vgx_tau_drift8s_col_kernel, dim3((unsigned)((a->p.H + 255) / 256), (unsigned)a->R), dim3(256), 0, s, *a);
            hipLaunchKernelGGL(vgx_tau_drift8s_tiles_kernel, dim3((unsigned)a->nt8, (unsigned)a->R), dim3(256), 0, s, *a);
            hipLaunchKernelGGL(vgx_tau_drift8s_sus_kernel, dim3((unsigned)a->p.P, (unsigned)a->R), dim3(64), 0, s, *a);
            return hipGetLastError();
        }
        const int low = a->p.sites < VGX_D8_LOW ? a->p.sites : VGX_D8_LOW;
        const size_t lds = (size_t)1 << (2 * low);
        hipError_t err = hipFuncSetAttribute((const void *)vgx_tau_drift8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        const dim3 grid((unsigned)(8 * ((a->p.P + 7) / 8) * a->nt8), (unsigned)a->R);
        hipLaunchKernelGGL(vgx_tau_drift8_kernel, grid, dim3(D8_TB), lds, s, *a);
#ifdef VGX_D8_STATS
        {
            unsigned long long h[8];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(vgx_d8_stats), sizeof(h));
            fprintf(stderr, "d8 stats (cumulative): wave turns %llu, past level 1 %llu, past level 2 %llu; lane turns past 1 %llu, compartments exact %llu; first turns with thr > 1.5: %llu, sum thr %llu, sum U %llu\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        }
#endif
        return hipGetLastError();
    }
    if (a->has_mig && a->mig_uniform) {
        hipLaunchKernelGGL(vgx_tau_colsum_kernel, dim3((unsigned)((a->p.H + 4 * TB - 1) / (4 * TB)), (unsigned)a->R), dim3(TB), 0, s, *a);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    } else if (a->has_mig) {
        size_t lds = (size_t)a->p.P * TH * 4;
        hipError_t err = hipFuncSetAttribute((const void *)vgx_tau_migin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        int ntiles = (a->p.H + TH - 1) / TH;
        int gx = ntiles < 2048 ? ntiles : 2048;
        hipLaunchKernelGGL(vgx_tau_migin_kernel, dim3((unsigned)gx, (unsigned)a->R), dim3(MIGIN_TB), lds, s, a->Aeff, a->I, a->migIn,
                           a->active, a->p.P, a->Ppad, a->p.H);
        err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    unsigned tiles = (unsigned)((a->p.H + TB - 1) / TB);
    unsigned gx = tiles < 32u ? tiles : 32u;
    if (tau_drift_tiled_ok(a->p.sites, a->mut_uniform) && a->mutHi) {
        const int sites = a->p.sites, low = sites < VGX_DRIFT_LOW ? sites : VGX_DRIFT_LOW, nh = sites - low;
        if (nh > 0) {
            const int rows = 1 << (2 * nh), CH = VGX_MUTHIGH_CELLS / rows < 4096 ? VGX_MUTHIGH_CELLS / rows : 4096;
            const size_t lds = (size_t)rows * CH * 4;
            hipError_t err = hipFuncSetAttribute((const void *)vgx_tau_muthigh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (err != hipSuccess) return err;
            hipLaunchKernelGGL(vgx_tau_muthigh_kernel, dim3((unsigned)((1 << (2 * low)) / CH), (unsigned)a->p.P, (unsigned)a->R), dim3(TB), lds, s, *a);
        }
        if (a->mutlow_fast) {
            // blocks per (population, replicate): enough blocks to fill the chip, the rest of the tiles in each block's loop
            const dim3 grid((unsigned)vgxi_tau_drift_blocks(a), (unsigned)a->p.P, (unsigned)a->R);
            const bool c1 = a->p.C == 1, s1 = a->p.S == 1;
            if (c1 && s1) hipLaunchKernelGGL((vgx_tau_drift_fast_kernel<true, true>), grid, dim3(TB), 0, s, *a);
            else if (c1) hipLaunchKernelGGL((vgx_tau_drift_fast_kernel<true, false>), grid, dim3(TB), 0, s, *a);
            else if (s1) hipLaunchKernelGGL((vgx_tau_drift_fast_kernel<false, true>), grid, dim3(TB), 0, s, *a);
            else hipLaunchKernelGGL((vgx_tau_drift_fast_kernel<false, false>), grid, dim3(TB), 0, s, *a);
        } else
            hipLaunchKernelGGL(vgx_tau_drift_tiled_kernel, dim3((unsigned)(a->p.H >> (2 * low)), (unsigned)a->p.P, (unsigned)a->R), dim3(TB), 0, s, *a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(vgx_tau_drift_kernel, dim3(gx, (unsigned)a->p.P, (unsigned)a->R), dim3(TB), 0, s, *a);
    return hipGetLastError();
}
TAU_LAUNCH(tau_choose, dim3((unsigned)a->R), dim3(64))
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_tau_draw(const VgxTauArgs *a, hipStream_t s) {
    bool cdfL;
    size_t lds = tau_tab_lds_bytes(a->p.C, a->p.CB, a->p.S, a->p.P, cdfL);
    const int tabs = lds == 0 ? 0 : (cdfL ? 2 : 1);
    const void *evk = tabs == 2 ? (const void *)vgx_tau_events_kernel<2> : tabs == 1 ? (const void *)vgx_tau_events_kernel<1> : (const void *)vgx_tau_events_kernel<0>;
    hipError_t err = hipFuncSetAttribute(evk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 16));
    if (err != hipSuccess) return err;
    if (a->front_on) {
        const void *fk = tabs == 2 ? (const void *)vgx_tau_events_kernel<2, true> : tabs == 1 ? (const void *)vgx_tau_events_kernel<1, true> : (const void *)vgx_tau_events_kernel<0, true>;
        err = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 16));
        if (err != hipSuccess) return err;
    }
    const dim3 grid(tau_draw_gx(a->p.H), (unsigned)a->p.P, (unsigned)a->R);
    // blocks of the events kernel per (population, replicate): about VGX_EV_BLOCKS in all, each taking its share of the shards
    unsigned egx = (unsigned)((VGX_EV_BLOCKS + (int64_t)a->p.P * a->R * a->ev_split - 1) / ((int64_t)a->p.P * a->R * a->ev_split));
    egx = egx < 1u ? 1u : (egx > tau_draw_gx(a->p.H) ? tau_draw_gx(a->p.H) : egx);
    const dim3 egrid(egx * (unsigned)a->ev_split, (unsigned)a->p.P, (unsigned)a->R);
    const bool lists = a->use_list && a->front_on && a->sparse;
    if (a->phase == 2) {
        // the front pass of this try has run (and with the lists, so has its scan)
    } else if (lists) {   // a sparse state: scan and front pass over the occupied compartments only
        const dim3 fgrid(1, (unsigned)a->p.P, (unsigned)a->R);
        hipLaunchKernelGGL(vgx_tau_listscan_kernel, grid, dim3(TB), 0, s, *a);
        if (tabs == 2) hipLaunchKernelGGL((vgx_tau_events_kernel<2, true>), fgrid, dim3(EB), lds, s, *a);
        else if (tabs == 1) hipLaunchKernelGGL((vgx_tau_events_kernel<1, true>), fgrid, dim3(EB), lds, s, *a);
        else hipLaunchKernelGGL((vgx_tau_events_kernel<0, true>), fgrid, dim3(EB), 16, s, *a);
    } else if (a->front_on && a->sparse) {   // the likely failures first (vgx_tau_front_kernel)
        const dim3 fgrid(1, (unsigned)a->p.P, (unsigned)a->R);
        hipLaunchKernelGGL(vgx_tau_front_kernel, grid, dim3(TB), 0, s, *a);
        if (tabs == 2) hipLaunchKernelGGL((vgx_tau_events_kernel<2, true>), fgrid, dim3(EB), lds, s, *a);
        else if (tabs == 1) hipLaunchKernelGGL((vgx_tau_events_kernel<1, true>), fgrid, dim3(EB), lds, s, *a);
        else hipLaunchKernelGGL((vgx_tau_events_kernel<0, true>), fgrid, dim3(EB), 16, s, *a);
    }
    if (a->phase == 1) return hipGetLastError();   // the front pass alone
    if (lists) {
        // (the queue is there already)
    } else if (a->p.C <= 16 && a->p.CB <= 16 && (a->p.H & 15) == 0) {   // the usual shapes: thresholds tabulated
        const bool c1 = a->p.C == 1, dn = !a->sparse;
        if (c1 && !dn) hipLaunchKernelGGL((vgx_tau_scan_fast_kernel<true, false>), grid, dim3(TB), 0, s, *a);
        else if (c1) hipLaunchKernelGGL((vgx_tau_scan_fast_kernel<true, true>), grid, dim3(TB), 0, s, *a);
        else if (!dn) hipLaunchKernelGGL((vgx_tau_scan_fast_kernel<false, false>), grid, dim3(TB), 0, s, *a);
        else hipLaunchKernelGGL((vgx_tau_scan_fast_kernel<false, true>), grid, dim3(TB), 0, s, *a);
    } else {
        hipLaunchKernelGGL(vgx_tau_scan_kernel, grid, dim3(TB), 0, s, *a);
    }
    if (tabs == 2) hipLaunchKernelGGL(vgx_tau_events_kernel<2>, egrid, dim3(EB), lds, s, *a);
    else if (tabs == 1) hipLaunchKernelGGL(vgx_tau_events_kernel<1>, egrid, dim3(EB), lds, s, *a);
    else hipLaunchKernelGGL(vgx_tau_events_kernel<0>, egrid, dim3(EB), 16, s, *a);
    return hipGetLastError();
}
extern "C" __attribute__((visibility("hidden"))) int64_t vgxi_tau_queue_shards(int64_t H, int64_t P) { return (int64_t)tau_draw_gx(H) * P; }
extern "C" __attribute__((visibility("hidden"))) int64_t vgxi_tau_queue_shard_max(int64_t H) { return tau_queue_shard_max(H); }
TAU_LAUNCH(tau_draw_big, dim3(VGX_BIG_BLOCKS + (unsigned)((a->p.P * a->p.S * a->p.S + TB - 1) / TB), (unsigned)a->R), dim3(TB))
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_tau_sieve(const VgxTauArgs *a, hipStream_t s) {
    unsigned tiles = (unsigned)((a->p.H + 4 * TB - 1) / (4 * TB));
    unsigned gx = tiles < 64u ? tiles : 64u;
    if (a->hist) hipLaunchKernelGGL(vgx_tau_sieve_hist_kernel, dim3((unsigned)a->p.P, (unsigned)a->R), dim3(64), 0, s, *a);
    else hipLaunchKernelGGL(vgx_tau_sieve_kernel, dim3(gx, (unsigned)a->p.P, (unsigned)a->R), dim3(TB), 0, s, *a);
    hipLaunchKernelGGL(vgx_tau_sieve_pick_kernel, dim3((unsigned)a->R), dim3(64), 0, s, *a);
    return hipGetLastError();
}
TAU_LAUNCH(tau_scatter, dim3((unsigned)a->inc_shards, (unsigned)a->R, (unsigned)(VGX_INC_SHARDS / a->inc_shards)), dim3(64))
// Number of thread blocks per replicate of the draw kernel: the cross-compartment list gets one shard per block (a
// power of two, at most VGX_INC_SHARDS), so that its whole capacity is usable whatever the grid size.
extern "C" __attribute__((visibility("hidden"))) int vgxi_tau_inc_shards(int64_t H, int64_t P) {
    int64_t blocks = (int64_t)tau_draw_gx(H) * P;
    int sh = 1;
    while (sh * 2 <= blocks && sh * 2 <= VGX_INC_SHARDS) sh *= 2;
    return sh;
}
TAU_LAUNCH(tau_arrivals, dim3((unsigned)a->inc_shards, (unsigned)a->R, (unsigned)(VGX_INC_SHARDS / a->inc_shards)), dim3(64))
TAU_LAUNCH(tau_verdict, dim3(32, (unsigned)a->R), dim3(TB))
TAU_LAUNCH(tau_apply, dim3((unsigned)a->inc_shards, (unsigned)a->R, (unsigned)(VGX_INC_SHARDS / a->inc_shards)), dim3(64))
TAU_LAUNCH(tau_check, CELL_GRID, dim3(TB))
TAU_LAUNCH(tau_suspect, dim3(32, (unsigned)a->R), dim3(TB))
TAU_LAUNCH(tau_decide, dim3((unsigned)a->R), dim3(64))
TAU_LAUNCH(tau_commit, CELL_GRID, dim3(TB))
TAU_LAUNCH(tau_finish, dim3((unsigned)a->R), dim3(64))

// ---- test hooks (include/vgx.h): the device samplers on their own ----------------------------------------------
// Philox4x32-10 for one (counter, key) on the device, and n independent draws of the Poisson sampler the step kernels
// use (inversion below a mean of 10, PTRS from 10 on), draw i from the stream of compartment i.
extern "C" __global__ void vgx_test_philox_kernel(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) vgx_philox4x32(ctr, key, out);
}
extern "C" __global__ void __launch_bounds__(TB) vgx_test_poisson_kernel(double lam, int64_t n, uint64_t seed, int64_t *out) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    TauRng g;
    g.init(seed, 0u, (uint64_t)i, 0u, 0u);
    out[i] = tau_poisson(g, lam);
}
extern "C" int vgx_test_philox(int on_device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    if (!ctr || !key || !out) return VGX_ERR_ARG;
    if (!on_device) { vgx_philox4x32(ctr, key, out); return VGX_OK; }
    uint32_t *d = nullptr;
    if (hipMalloc(&d, 10 * sizeof(uint32_t)) != hipSuccess) return VGX_ERR_HIP;
    hipError_t e1 = hipMemcpy(d, ctr, 16, hipMemcpyHostToDevice), e2 = hipMemcpy(d + 4, key, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(vgx_test_philox_kernel, dim3(1), dim3(64), 0, 0, d, d + 4, d + 6);
    hipError_t e3 = hipMemcpy(out, d + 6, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) ? VGX_OK : VGX_ERR_HIP;
}
extern "C" int vgx_test_poisson(double lam, int64_t n, uint64_t seed, int64_t *out) {
    if (!out || n < 0) return VGX_ERR_ARG;
    if (n == 0) return VGX_OK;
    int64_t *d = nullptr;
    if (hipMalloc(&d, (size_t)n * 8) != hipSuccess) return VGX_ERR_HIP;
    hipLaunchKernelGGL(vgx_test_poisson_kernel, dim3((unsigned)((n + TB - 1) / TB)), dim3(TB), 0, 0, lam, n, seed, d);
    hipError_t e = hipMemcpy(out, d, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? VGX_OK : VGX_ERR_HIP;
}
