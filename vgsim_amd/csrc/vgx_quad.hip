// vgx_quad.hip — persistent direct-Gillespie kernel, FOUR replicates per wavefront (one per 16-lane DPP row).
//
// Same path as vgx_direct.hip (SimulatePopulation pyx:396-429 and everything it calls: SampleTime pyx:476,
// GenerateEvent pyx:483, UpdateRates pyx:516, Birth pyx:568, Death/Sampling pyx:616/630, Mutation pyx:640,
// GenerateMigration pyx:672, Restart pyx:714, fastChoose / fastChoose_skip fast_choose.pxi:18/36, Events.AddEvent
// events.pxi:37), same HBM layout (vgx_dev.h), same bit-exact contract: every sum the reference forms left to right is
// formed left to right, no contraction.
//
// Why four per wave.  A sequential f64 sum costs one dependent v_fmac_f64 per term whatever the number of active lanes;
// the DPP source `row_newbcast:k` broadcasts inside a ROW of 16 lanes, so one such instruction can advance four
// independent chains, one per row.  vgx_direct.hip spends a full 64-lane issue on one useful addition (its replicate's
// chain) and, at the natural occupancy of a few haplotypes per population, keeps 3 of 4 rows idle in every list
// operation.  Here each row owns a replicate: lane (row, l) holds populations l, l+16, l+32, l+48 of its replicate and
// list entry 16c + l of the list chunk c being processed, every chain instruction serves four replicates, and everything
// that is per-replicate "scalar" state is a row-uniform VGPR value.  Control flow is wave-uniform; rows that take
// different branches (birth / death / mutation / migration, finished replicates) are predicated.
//
// Scope (the host chooses this kernel only then, vgx_api.hip): popNum <= 64, one susceptibility group, one rate class
// (every haplotype has the same bRate / dRate / sRate / susceptibility / total mutation rate), no population can switch
// its lockdown state, no recombination, exact mode — BASELINE configs 2 and 3.  With one susceptibility group
// immunePopRate is +0.0 (suscepTransition is a 1x1 zero) and popRate = infectPopRate + 0.0 = infectPopRate exactly;
// without lockdown switches contactDensity, effectiveMigration and maxEffectiveBirthMigration are functions of the
// parameters only and come from vgx_quad_prep_kernel, shared by all replicates.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_rowprim.h"
#include "vgx_rowlist.h"

#ifndef VGX_QUAD_WAVES
#define VGX_QUAD_WAVES 2     // waves per SIMD the register allocation aims at (measured: 3 spills into the hot loop and is slower)
#endif

// In-kernel stamps (diagnostic build only, -DVGX_PROFILE): shader cycles per phase of the loop, summed per wavefront
// into the debug buffer r.prof of its first replicate.  No stamp executes in the product build.
#ifdef VGX_PROFILE
#define QPROF(i)                                                        \
    do {                                                                \
        unsigned long long prof_t1 = __builtin_readcyclecounter();      \
        prof_acc[i] += prof_t1 - prof_t0;                               \
        prof_t0 = prof_t1;                                              \
    } while (0)
#else
#define QPROF(i)
#endif

namespace {

enum { Q_ERR_ZERO_WEIGHT = 3, Q_ERR_CAPACITY = 4, Q_ERR_LOOP_GUARD = 5 };
enum { QEV_BIRTH = 0, QEV_DEATH, QEV_SAMPLING, QEV_MUTATION, QEV_SUSCCHANGE, QEV_MIGRATION };
enum { ST_REBUILD = 0, ST_RUN = 1, ST_DONE = 2 };

// ---- LDS layout (bytes), one wavefront per workgroup ---------------------------------------------------------
// model constants [64] f64: cd, as, smul (= sRate * samplingMultiplier), maxEBM                      2048
// PCG64 jump-ahead [16][4] u64: a^(l+1), 1 + a + ... + a^l (high, low words) for l = 0..15            512
// per replicate   infect[64], birthC[64] f64; totS[64], totI[64] i64; nocc[64] i32; cc[4] f64 (serial prefix sum of
//                 popRate at the end of each 16-population slot); counters[8] i64 (births, deaths, mutations,
//                 accepted / rejected migrations)                                                     4 x 2400
#define Q_CONST_BYTES 2048
#define Q_RNG_BYTES 512
#define Q_REP_BYTES 2912     // ... + the stage of eight event records (256 bytes)
#define Q_LDS_BYTES (Q_CONST_BYTES + Q_RNG_BYTES + 4 * Q_REP_BYTES)
enum { QC_B = 0, QC_D, QC_M, QC_MIGP, QC_MIGN };

// Long lists are streamed in TILES of 64 entries: lane l of the row holds entries 4l .. 4l+3 (32 contiguous bytes per lane,
// 512 per row and load instruction group).  List order = (k = 0..15: lane k's four entries), so the chain takes the four
// registers in turn for every broadcast lane.
#define QFM4(K) "v_fmac_f64_dpp %0, %1, %5 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                "v_fmac_f64_dpp %0, %2, %5 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                "v_fmac_f64_dpp %0, %3, %5 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                "v_fmac_f64_dpp %0, %4, %5 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
static __device__ __forceinline__ double row_sum64(double w0, double w1, double w2, double w3, double acc) {
    const double one = 1.0;
    asm volatile("s_nop 1\n\t" QFM4(0) QFM4(1) QFM4(2) QFM4(3) QFM4(4) QFM4(5) QFM4(6) QFM4(7) QFM4(8) QFM4(9) QFM4(10) QFM4(11)
                     QFM4(12) QFM4(13) QFM4(14) QFM4(15)
                 : "+v"(acc)
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(one));
    return acc;
}
// the serial prefix of every entry of a tile: p[j] of lane l = carry + (all entries before 4l + j) + its own.  The chain is the one
// of row_sum64; lane K keeps the running sum as it stands before its own four entries (one select per group of four steps) and
// forms its four prefixes from it afterwards — the same additions on the same operands in the same order as the chain made.
#define QCAP4(K)                                                                                                   \
    st = rl_ == K ? acc : st;                                                                                      \
    asm volatile(QFM4(K) : "+v"(acc) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(one));
static __device__ __forceinline__ void row_scan64(double w0, double w1, double w2, double w3, double carry, double &p0, double &p1,
                                                  double &p2, double &p3) {
    const double one = 1.0;
    const int rl_ = threadIdx.x & 15;
    double acc = carry, st = carry;
    asm volatile("s_nop 1" ::: );
    QCAP4(0) QCAP4(1) QCAP4(2) QCAP4(3) QCAP4(4) QCAP4(5) QCAP4(6) QCAP4(7) QCAP4(8) QCAP4(9) QCAP4(10) QCAP4(11)
    QCAP4(12) QCAP4(13) QCAP4(14) QCAP4(15)
    p0 = st + w0; p1 = p0 + w1; p2 = p1 + w2; p3 = p2 + w3;
}
// QT = tiles of 64 four-byte counts a row keeps in flight ahead of its summation chain: 1 in vgx_quad_kernel, 5 in
// vgx_quad_long_kernel (start states with lists longer than one tile).  The event loop is sensitive to its CODE SIZE (46 to
// 51 KB of instructions against a 64 KB instruction cache shared by two CUs): every tile of look-ahead unrolls two more
// 64-step chains, and on short lists — where that code never runs — one tile less is worth 3-4 % of throughput; the branch
// hints on the slow paths (mutation, migration, list insertions / removals, restart, the other form of the list passes) move
// their code behind the loop for another 3 %.
struct QTile { int c0, c1, c2, c3; };
static __device__ __forceinline__ QTile tile_load(const int32_t *l3, int t, int rl) {   // entries 64t + 4rl .. + 3 of the 4-byte counts
    const int4 a = *(const int4 *)(l3 + (int64_t)t * 64 + 4 * rl);
    QTile q;
    q.c0 = a.x; q.c1 = a.y; q.c2 = a.z; q.c3 = a.w;
    return q;
}

struct QSel { int k_hit, hap_hit, err; double pre_hit, w_hit; int64_t cnt_hit; };
// The haplotype choice over lists longer than 64 entries (inlined: an out-of-line call measured slower in both regimes).
// Called with all lanes active.
template <int QT>
static __device__ __forceinline__ void q_long_select(const int32_t *ln, const int32_t *lh, int n_sel, int maxn, double tE,
                                                               double r2, bool evn, int H, QSel &o, const double *ltp) {
    const int rl = threadIdx.x & 15;
    o.k_hit = -1; o.err = 0;
    // long lists: the running sum advances one tile of 64 entries per step; the tile in which it first reaches r is
    // then scanned entry by entry — same additions, same order.  Loads run QT tiles ahead (unconditional: a row
    // that is through, or has its hit, re-reads its tile 0; every list is followed by 64 entries of padding).
    QTile buf[QT];
    double carry = 0.0, carry_hit = 0.0;
    int t_hit = -1;
    const int nt = (n_sel + 63) >> 6, maxt = (maxn + 63) >> 6;
    // tiles that lie inside the list of EVERY row that has one: no bounds to look at (a row without a list adds zeros)
    const int full = -rows_max(n_sel > 0 ? -n_sel : -0x7fffffff) >> 6;
    const double tEz = n_sel > 0 ? tE : 0.0;
    if (QT >= 4) {
        // The tile by the running sums the last refresh of this population left at the end of every tile (q_long_sum: the same
        // additions in the same order as the loop below would make, on the same counts and the same tE): lane l looks at tiles
        // 4l .. 4l+3 of a group of 64.  Rows whose list fits one tile have nothing cached: their tile is scanned below.
        const int ntc = nt > 1 ? nt : 0;
        const int maxtc = rows_max(ntc);
        const int lastc = max(ntc - 1, 0);
        double gcarry = 0.0;          // the running sum at the end of the previous group
        for (int tb = 0; tb < maxtc; tb += 64) {
            const int j0 = tb + 4 * rl;
            const double q0 = ltp[min(j0 + 0, lastc)], q1 = ltp[min(j0 + 1, lastc)], q2 = ltp[min(j0 + 2, lastc)], q3 = ltp[min(j0 + 3, lastc)];
            const int kk = (j0 + 0 < ntc && !(q0 < r2)) ? 0 : (j0 + 1 < ntc && !(q1 < r2)) ? 1 : (j0 + 2 < ntc && !(q2 < r2)) ? 2
                         : (j0 + 3 < ntc && !(q3 < r2)) ? 3 : 4;
            const int q = row_min(t_hit < 0 && kk < 4 ? 4 * rl + kk : 64);
            double pl = q3;            // the lane below's last tile
            {
                const int lo = VGX_DPP_SHR(__double2loint(q3), 1), hi = VGX_DPP_SHR(__double2hiint(q3), 1);
                pl = rl == 0 ? gcarry : __hiloint2double(hi, lo);
            }
            if (t_hit < 0 && q < 64) {
                const int k = q & 3;
                t_hit = tb + q;
                carry_hit = rowget_f64(k == 0 ? pl : k == 1 ? q0 : k == 2 ? q1 : q2, q >> 2);
            }
            gcarry = rowget_f64(q3, 15);
            if (!__ballot(ntc > tb + 64 && t_hit < 0)) break;
        }
        if (ntc > 0) carry = ltp[lastc];                   // no hit: the total of the whole list
        else if (n_sel > 0) { t_hit = 0; carry_hit = 0.0; }
    } else {
#pragma unroll
    for (int d = 0; d < QT; ++d) buf[d] = tile_load(ln, d < nt ? d : 0, rl);
    for (int tb = 0; tb < maxt; tb += QT) {
#pragma unroll
        for (int d = 0; d < QT; ++d) {
            const int t = tb + d;
            const QTile c = buf[d];
            buf[d] = tile_load(ln, (t + QT < nt && t_hit < 0) ? t + QT : 0, rl);
            double w0, w1, w2, w3;
            if (QT >= 4 && t < full) {   // (the short-list instantiation keeps its code small: see VGX_QT_SHORT)
                w0 = tEz * (double)c.c0; w1 = tEz * (double)c.c1; w2 = tEz * (double)c.c2; w3 = tEz * (double)c.c3;
            } else {
                const int e0 = t * 64 + 4 * rl;
                w0 = e0 + 0 < n_sel ? tE * (double)c.c0 : 0.0; w1 = e0 + 1 < n_sel ? tE * (double)c.c1 : 0.0;
                w2 = e0 + 2 < n_sel ? tE * (double)c.c2 : 0.0; w3 = e0 + 3 < n_sel ? tE * (double)c.c3 : 0.0;
            }
            const double acc = row_sum64(w0, w1, w2, w3, carry);
            if (t_hit < 0 && t < nt && !(acc < r2)) { t_hit = t; carry_hit = carry; }
            carry = acc;
        }
        if (!__ballot(nt > tb + QT && t_hit < 0)) break;
    }
    }
    // refine inside the hit tile (rows without a hit look at their last tile for the H-1 rule)
    const int tt = t_hit >= 0 ? t_hit : max((n_sel - 1) >> 6, 0);
    const QTile c = tile_load(ln, tt, rl);
    const int4 hv = *(const int4 *)(lh + (int64_t)tt * 64 + 4 * rl);     // the haplotypes with the counts (every list is followed by a tile of padding)
    const int e0 = tt * 64 + 4 * rl;
    const double w0 = e0 + 0 < n_sel ? tE * (double)c.c0 : 0.0, w1 = e0 + 1 < n_sel ? tE * (double)c.c1 : 0.0;
    const double w2 = e0 + 2 < n_sel ? tE * (double)c.c2 : 0.0, w3 = e0 + 3 < n_sel ? tE * (double)c.c3 : 0.0;
    double p0, p1, p2, p3;
    row_scan64(w0, w1, w2, w3, t_hit >= 0 ? carry_hit : 0.0, p0, p1, p2, p3);
    int mine = 64;
    if (t_hit >= 0) {
        if (e0 + 3 < n_sel && !(p3 < r2)) mine = 4 * rl + 3;
        if (e0 + 2 < n_sel && !(p2 < r2)) mine = 4 * rl + 2;
        if (e0 + 1 < n_sel && !(p1 < r2)) mine = 4 * rl + 1;
        if (e0 + 0 < n_sel && !(p0 < r2)) mine = 4 * rl + 0;
    }
    const int q = row_min(mine);
    const int qe = q < 64 ? q : ((n_sel - 1) & 63);      // entry inside the tile: the hit, or the list's last
    const int ql = qe >> 2, qj = qe & 3;
    const double psel = qj == 0 ? p0 : qj == 1 ? p1 : qj == 2 ? p2 : p3;
    const double wsel = qj == 0 ? w0 : qj == 1 ? w1 : qj == 2 ? w2 : w3;
    const int64_t csel = (int64_t)(qj == 0 ? c.c0 : qj == 1 ? c.c1 : qj == 2 ? c.c2 : c.c3);
    // no hit: the total of the whole list (a one-tile list of the cached form: the end of this scan)
    o.pre_hit = q < 64 ? rowget_f64(psel, ql) : (QT >= 4 && nt <= 1) ? rowget_f64(p3, 15) : carry;
    o.w_hit = rowget_f64(wsel, ql);
    o.cnt_hit = rowget_i64(csel, ql);
    o.hap_hit = rowget_i32(qj == 0 ? hv.x : qj == 1 ? hv.y : qj == 2 ? hv.z : hv.w, ql);
    if (q < 64) o.k_hit = tt * 64 + q;
    else if (evn) {
        if (n_sel > 0 && o.hap_hit == H - 1 && (QT < 4 || o.cnt_hit != 0)) o.k_hit = n_sel - 1; else o.err = Q_ERR_ZERO_WEIGHT + 256 * 3;
    }
            
}

// infectPopRate over a list longer than 64 entries (pyx:519-528)
// The long-list instantiation (QT >= 4) streams a ONE-BYTE copy of the counts (l8: min(count, 255), kept next to the 4-byte
// copy by every count update of that kernel): a tile of 64 entries is one 4-byte load per lane, QB8 tiles in flight.  A byte of
// 255 stands for "255 or more": a row that meets one takes that tile's counts from the 4-byte copy.  The running sum at the end
// of every tile is left in ltp for the next haplotype choice in this population (q_long_select).
#define VGX_QB8 6      // (tiles in flight; 4: 1.88e8, 6: 2.00e8, 8: 1.92e8 events/s at 4096-entry lists: the unrolled chains are code)
template <int QT>
static __device__ __forceinline__ double q_long_sum(const int32_t *ln, const uint8_t *l8, int n, int maxn, double tE, double *ltp) {
    const int rl = threadIdx.x & 15;
    double acc = 0.0;
    const int nt = (n + 63) >> 6, maxt = (maxn + 63) >> 6;
    // tiles that lie inside the list of EVERY row that has one: no bounds to look at (a row without a list adds zeros)
    const int full = -rows_max(n > 0 ? -n : -0x7fffffff) >> 6;
    const double tEz = n > 0 ? tE : 0.0;
    if (QT >= 4) {
        uint32_t buf8[VGX_QB8];
#pragma unroll
        for (int d = 0; d < VGX_QB8; ++d) buf8[d] = *(const uint32_t *)(l8 + (int64_t)(d < nt ? d : 0) * 64 + 4 * rl);
        for (int tb = 0; tb < maxt; tb += VGX_QB8) {
#pragma unroll
            for (int d = 0; d < VGX_QB8; ++d) {
                const int t = tb + d;
                const uint32_t w = buf8[d];
                buf8[d] = *(const uint32_t *)(l8 + (int64_t)(t + VGX_QB8 < nt ? t + VGX_QB8 : 0) * 64 + 4 * rl);
                if (t >= maxt) continue;       // (the group runs past the longest list of the four rows)
                int c0 = (int)(w & 255u), c1 = (int)((w >> 8) & 255u), c2 = (int)((w >> 16) & 255u), c3 = (int)(w >> 24);
                const bool sat = t < nt && ((((w & 0x7F7F7F7Fu) + 0x01010101u) & w & 0x80808080u) != 0u);      // some byte == 255
                if (__builtin_expect(__ballot(sat) != 0, 0)) {
                    const QTile q = tile_load(ln, t < nt ? t : 0, rl);
                    if (row_max(sat ? 1 : 0)) { c0 = q.c0; c1 = q.c1; c2 = q.c2; c3 = q.c3; }
                }
                double w0, w1, w2, w3;
                if (t < full) {
                    w0 = tEz * (double)c0; w1 = tEz * (double)c1; w2 = tEz * (double)c2; w3 = tEz * (double)c3;
                } else {
                    const int e0 = t * 64 + 4 * rl;
                    w0 = e0 + 0 < n ? tE * (double)c0 : 0.0; w1 = e0 + 1 < n ? tE * (double)c1 : 0.0;
                    w2 = e0 + 2 < n ? tE * (double)c2 : 0.0; w3 = e0 + 3 < n ? tE * (double)c3 : 0.0;
                }
                acc = row_sum64(w0, w1, w2, w3, acc);
                if (t < nt && rl == (t & 15)) ltp[t] = acc;
            }
        }
        return acc;
    }
    QTile buf[QT];
#pragma unroll
    for (int d = 0; d < QT; ++d) buf[d] = tile_load(ln, d < nt ? d : 0, rl);
    for (int tb = 0; tb < maxt; tb += QT) {
#pragma unroll
        for (int d = 0; d < QT; ++d) {
            const int t = tb + d;
            const QTile c = buf[d];
            buf[d] = tile_load(ln, t + QT < nt ? t + QT : 0, rl);
            const int e0 = t * 64 + 4 * rl;
            const double w0 = e0 + 0 < n ? tE * (double)c.c0 : 0.0, w1 = e0 + 1 < n ? tE * (double)c.c1 : 0.0;
            const double w2 = e0 + 2 < n ? tE * (double)c.c2 : 0.0, w3 = e0 + 3 < n ? tE * (double)c.c3 : 0.0;
            acc = row_sum64(w0, w1, w2, w3, acc);
        }
    }
    return acc;
}

struct QArgs {          // what the prep kernel leaves for all replicates
    const double *effMig;   // [P][P]
    const double *maxEBM;   // [P]
    const int32_t *has_mig; // [1]
};

}  // namespace

// effectiveMigration, maxEffectiveBirthMigration (pyx:327-338) from the parameters and the (constant) contact densities:
// lane <-> target population, serial over sources and the inner sum, the reference's operation order.
extern "C" __global__ void __launch_bounds__(64) vgx_quad_prep_kernel(VgxDevParams p, const double *cd, double *effMig,
                                                                     double *maxEBM, int32_t *has_mig) {
    const int P = p.P;
    bool anyf = false;
    for (int base = 0; base < P; base += 64) {
        const int pn2 = base + (int)threadIdx.x;
        double mx = 0.0;
        if (pn2 < P) {
            const double *m2 = p.mig + (int64_t)pn2 * P;
            for (int pn1 = 0; pn1 < P; ++pn1) {
                if (pn1 == pn2) continue;
                const double *m1 = p.mig + (int64_t)pn1 * P;
                double e = 0.0;
                for (int pn3 = 0; pn3 < P; ++pn3) e += m1[pn3] * m2[pn3] * cd[pn3] / p.actualSizes[pn3];
                effMig[(int64_t)pn1 * P + pn2] = e;
                if (e > mx) mx = e;
            }
            maxEBM[pn2] = mx * p.maxEffectiveBirth;
        }
        anyf = anyf || (__ballot(pn2 < P && mx * p.maxEffectiveBirth != 0.0) != 0ull);
    }
    if (threadIdx.x == 0) *has_mig = anyf ? 1 : 0;
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_quad_prep(const VgxDevParams *p, const double *cd, double *effMig,
                                                                                 double *maxEBM, int32_t *has_mig, hipStream_t stream) {
    hipLaunchKernelGGL(vgx_quad_prep_kernel, dim3(1), dim3(64), 0, stream, *p, cd, effMig, maxEBM, has_mig);
    return hipGetLastError();
}

template <int QT>
static __device__ __forceinline__ void quad_body(const VgxDirectArgs &a, const QArgs &qa) {
    const int lane = threadIdx.x, row = lane >> 4, rl = lane & 15;
    const VgxDevParams &p = a.p;
    const VgxDevRep &r = a.r;
    const int P = p.P, sites = p.sites, H = p.H;
    const int64_t R = a.n_replicates;
    const int64_t rep_raw = (int64_t)blockIdx.x * 4 + row;
    const bool live = rep_raw < R;
    const int64_t rep = live ? rep_raw : R - 1;   // idle rows shadow the last replicate read-only
    const int nslot = (P + 15) >> 4;              // register slots of the population arrays in use
    const bool has_mig = qa.has_mig[0] != 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *k_cd = (double *)smem, *k_as = k_cd + 64, *k_smul = k_as + 64, *k_mebm = k_smul + 64;
    uint64_t *k_jump = (uint64_t *)(smem + Q_CONST_BYTES) + rl * 4;
    unsigned char *blk = smem + Q_CONST_BYTES + Q_RNG_BYTES + row * Q_REP_BYTES;
    double *s_inf = (double *)blk, *s_bc = s_inf + 64;
    int64_t *s_ts = (int64_t *)(s_bc + 64), *s_ti = s_ts + 64;
    int32_t *s_nocc = (int32_t *)(s_ti + 64);
    double *s_cc = (double *)(s_nocc + 64);
    int64_t *s_cnt = (int64_t *)(s_cc + 4);
    uint64_t *s_inc = (uint64_t *)(s_cnt + 6);         // counters use 5 of their 8 slots; the last two hold the PCG64 increment
    int32_t *s_zero = (int32_t *)(s_inc + 2);           // long-list kernel: zero-count entries in every population's list (vgx_rowlist.h)
    // Event records are staged here, eight per replicate, and written out together: 192 contiguous bytes of columns + 64 of rates
    // instead of eight lone 24 + 8 byte stores per stream (16 384 streams: every lone store cost a 64-byte memory transaction).
    // Dwords 0..47: the six columns of records 0..7; dwords 48..63: their rates.
    uint32_t *s_stage = (uint32_t *)(s_zero + 64);
    int stage_n = 0;                // records staged (row-uniform)
    int64_t stage_slot0 = 0;        // log slot of the first of them
    auto stage_flush = [&](bool f) {   // rows with f: their staged records go to the log
        WSYNC();
        if (f && live) {
            const int n6 = stage_n * 6;
            int32_t *c = r.ev_cols + (rep * r.evcap + stage_slot0) * VGX_EV_COLS;
            uint32_t *q = (uint32_t *)(r.ev_rate + rep * r.evcap + stage_slot0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (rl + 16 * j < n6) c[rl + 16 * j] = (int32_t)s_stage[rl + 16 * j];
            if (rl < 2 * stage_n) q[rl] = s_stage[48 + rl];
        }
        if (f) stage_n = 0;
        WSYNC();
    };
#define QBUMP(i) do { if (rl == 0) s_cnt[i] += 1; } while (0)

    // the single rate class
    const double c_b = p.cb_b[0], c_sig = p.cb_sigma[0], c_d = p.c_d[0], c_s = p.c_s[0], c_tm = p.c_tm[0];

    // ---- load state ----
    {
        const double *gD = r.popD + rep * PD_COUNT * P;
        const int64_t *gI64 = r.popI + rep * PI_COUNT * P;
        const int32_t *gN = r.nocc + rep * P;
        const int pn = lane;
        k_cd[pn] = pn < P ? gD[PD_CD * P + pn] : 0.0;   // identical in every replicate (no lockdown switches)
        k_as[pn] = pn < P ? p.actualSizes[pn] : 1.0;
        k_smul[pn] = pn < P ? c_s * p.sampMult[pn] : 0.0;
        k_mebm[pn] = pn < P ? qa.maxEBM[pn] : 0.0;
        for (int s = 0; s < 4; ++s) {
            const int pq = s * 16 + rl;
            const bool ok = pq < P;
            s_inf[pq] = 0.0; s_bc[pq] = 0.0;
            if (rl < 4) s_cc[rl] = 0.0;
            s_ts[pq] = ok ? gI64[PI_TOTSUS * P + pq] : 0;
            s_ti[pq] = ok ? gI64[PI_TOTINF * P + pq] : 0;
            s_nocc[pq] = ok ? gN[pq] : 0;
            s_zero[pq] = 0;                 // (the lists arrive settled)
        }
    }
    WSYNC();

    const int64_t cap = r.cap, capT = r.capT;
    int32_t *lhap = r.lhap + rep * P * cap;
    int32_t *lcls = r.lcls + rep * P * cap;
    int64_t *lcnt = r.lcnt + rep * P * cap;
    int32_t *lcnt32 = r.lcnt32 + rep * P * cap;   // the same counts in 4 bytes: what the streaming passes over long lists read
    // ... and in ONE byte (min(count, 255)) behind the 4-byte copies of all replicates: what the long-list kernel's rate refresh streams
#define L8(pop) ((uint8_t *)(r.lcnt32 + R * P * cap) + (rep * P + (int64_t)(pop)) * cap)
#define B8(v) ((uint8_t)((v) < 255 ? (v) : 255))
    int64_t *ltsum = r.ltsum + rep * P * capT;
    const bool has_traj = r.traj != nullptr;
    const VgxRepScalars *sc = r.sc + rep;

    double t_now = sc->currentTime, totalRate = 0.0, totalMig = 0.0;
    int64_t gI = sc->globalInfectious, ev_ptr = sc->ev_ptr;
    int64_t cS = sc->sCounter;
    if (rl == 0) {
        s_cnt[QC_B] = sc->bCounter; s_cnt[QC_D] = sc->dCounter; s_cnt[QC_M] = sc->mCounter;
        s_cnt[QC_MIGP] = sc->migPlus; s_cnt[QC_MIGN] = sc->migNonPlus;
    }
    int64_t loops = 0, att_loops = 0, good_attempt = sc->good_attempt;
    int att = 0, restarts = 0, last_att = -1, traj_next = 0;       // attempts and grid points are far below 2^31
    int st = live ? ST_REBUILD : ST_DONE, err = 0;
    bool open = false;
    const double tlimit = (double)a.time;
    const bool has_tl = !(a.time == -1.0f);

    // random stream of the row: 16 outputs (8 loop iterations) per refill, lane l of the row jumping l+1 steps ahead
    // (state * a^(l+1) + inc * (1 + a + ... + a^l), exact 128-bit arithmetic): the uniforms are the reference's, in order.
    // Even lanes hold -log(u) for SampleTime (pyx:477), odd lanes the uniform of GenerateEvent (pyx:488).
    if (row == 0) {
        const uint64_t MH = 0x2360ED051FC65DA4ull, ML = 0x4385DF649FCCF645ull;
        uint64_t Ah = MH, Al = ML, Gh = 0, Gl = 1;
        for (int j = 1; j < 16; ++j) {
            uint64_t nh, nl, gh, gl;
            vgx_mul128(Ah, Al, MH, ML, nh, nl);
            vgx_mul128(Gh, Gl, MH, ML, gh, gl);
            vgx_add128(gh, gl, 0, 1);
            if (j <= rl) { Ah = nh; Al = nl; Gh = gh; Gl = gl; }
        }
        k_jump[0] = Ah; k_jump[1] = Al; k_jump[2] = Gh; k_jump[3] = Gl;
    }
    WSYNC();
    uint64_t g_sh = 0, g_sl = 0;                       // row-uniform: stream position (the increment lives in LDS)
    double g_val = 0.0;                                // this lane's output of the current batch
    int pos = 8;                                       // iterations consumed from the row's batch (8 = empty)
    // a list of up to 64 entries read for the haplotype choice stays in registers for the rate refresh
    int64_t ch_cn[4] = {0, 0, 0, 0};

#ifdef VGX_PROFILE
    unsigned long long prof_acc[VGX_PROF_SLOTS], prof_t0 = __builtin_readcyclecounter();
    for (int i = 0; i < VGX_PROF_SLOTS; ++i) prof_acc[i] = 0;
#endif
    while (true) {
        QPROF(0);
        const bool run = st != ST_DONE;
        if (!__ballot(run)) break;
        const bool rebuild = st == ST_REBUILD;

        // ================= front: open the attempt, loop condition (pyx:402-407) =================
        bool end_attempt = false, ev = false;
        if (st == ST_RUN) {
            if (!open) {
                if (att >= a.attempts) {
                    st = ST_DONE;
                } else {
                    VgxPcg64 sd;
                    vgx_pcg64_seed(sd, (uint64_t)r.seeds[rep], (uint32_t)att);
                    g_sh = sd.sh; g_sl = sd.sl;
                    if (rl == 0) { s_inc[0] = sd.ih; s_inc[1] = sd.il; }     // the stream's increment: read back at every refill
                    pos = 8;
                    open = true;
                    last_att = att; att_loops = 0;
                    if (!(totalRate + totalMig != 0.0 && gI != 0)) end_attempt = true;   // pyx:404
                }
            }
            if (st == ST_RUN && !end_attempt &&
                !(ev_ptr < a.ev_size && (a.sample_size == -1 || cS <= a.sample_size) && (!has_tl || t_now < tlimit)))
                end_attempt = true;
            if (st == ST_RUN && !end_attempt) {
                if (loops >= a.max_loop) { err = Q_ERR_LOOP_GUARD; st = ST_DONE; }
                else ev = true;
            }
        }

        // update request of this pass: populations [u_lo, u_hi) of the row (rebuild: all; event: the one it touched)
        int u_lo = 0, u_hi = rebuild ? P : 0;
        // deferred list operations (mutation: +1 on the new haplotype, -1 on the old; death of the last carrier: -1;
        // accepted migration: +1 in the target population)
        int op_n = 0, op_pi = 0, op_h0 = 0, op_h1 = 0;
        int op_d0 = 0;
        int e_type = -1, e_hap = 0, e_pop = 0, e_nh = 0, e_np = 0;
        double den = 0.0;
        int ch_pi = -1;           // population whose list (<= 64 entries) ch_cn holds, as it stands after the event; -1: none
        double m_pre[4] = {0.0, 0.0, 0.0, 0.0};   // migrationRates[pm_pi, :] of the event's population, loaded with its list
        int pm_pi = -1;

        QPROF(1);
        if (__ballot(ev)) {
            // ---- random numbers: refill the batch of every row that ran dry ----
            if (__builtin_expect(__ballot(ev && pos == 8) != 0, 0)) {   // (every eighth iteration while the rows stay in step)
                const bool fill = ev && pos == 8;
                uint64_t h, l, ch, cl;
                vgx_mul128(k_jump[0], k_jump[1], g_sh, g_sl, h, l);
                vgx_mul128(k_jump[2], k_jump[3], s_inc[0], s_inc[1], ch, cl);
                vgx_add128(h, l, ch, cl);
                const double u = vgx_pcg64_output_double(h, l);
                const double v = (rl & 1) ? u : -vgx_log(u);
                const uint64_t nh = (uint64_t)rowget_i64((int64_t)h, 15), nl = (uint64_t)rowget_i64((int64_t)l, 15);
                if (fill) { g_val = v; g_sh = nh; g_sl = nl; pos = 0; }
            }
            const int pp = min(pos, 7);
            const double nlog = rowget_f64(g_val, 2 * pp), u2 = rowget_f64(g_val, 2 * pp + 1);
            if (ev) { pos += 1; loops += 1; att_loops += 1; }
            QPROF(2);
            den = totalRate + totalMig;
            const double t_new = t_now + (nlog / den);   // SampleTime pyx:476-478
            // summary trajectories: the state before the event for every grid point the step passes
            if (has_traj) {
                while (true) {
                    const double tg = r.traj_t0 + (double)traj_next * r.traj_dt;
                    const bool emit = ev && live && traj_next < r.traj_points && tg < t_new;
                    if (__builtin_expect(!__ballot(emit), 1)) break;
                    if (emit) {
                        double *o = r.traj + (rep * r.traj_points + traj_next) * (int64_t)P * 2;
                        for (int s = 0; s < nslot; ++s) {
                            const int pn = s * 16 + rl;
                            if (pn < P) { o[pn * 2 + 0] = (double)s_ti[pn]; o[pn * 2 + 1] = (double)s_ts[pn]; }
                        }
                        traj_next += 1;
                    }
                }
            }
            if (ev) t_now = t_new;

            QPROF(3);
            // ================= GenerateEvent (pyx:483-512) =================
            double rn = u2;
            const double choose0 = rn * den;               // kept for the migration branch (pyx:490 / pyx:512)
            double choose = choose0;
            const bool evn = ev && (totalRate > choose);   // an event inside a population
            const bool evm = ev && !evn;                   // a migration attempt

            // ---- population by fastChoose over popRate = infectPopRate (fc:18-31).  The serial prefix sums at the end of
            // every 16-population slot are cached (they are partial sums of the totalRate loop, pyx:537-539): the first slot
            // whose end reaches r holds the pick, and the prefix sums inside it are formed again — same additions, same order.
            int pi = 0;
            {
                rn = choose / totalRate;
                const double rr_ = totalRate * rn;
                const double c0 = s_cc[0], c1 = s_cc[1], c2 = s_cc[2], c3 = s_cc[3];   // slots beyond P repeat the total
                const int slot = !(c0 < rr_) ? 0 : !(c1 < rr_) ? 1 : !(c2 < rr_) ? 2 : 3;
                const bool any = !(c3 < rr_);
                const double cin = slot == 0 ? 0.0 : slot == 1 ? c0 : slot == 2 ? c1 : c2;
                const double w = s_inf[slot * 16 + rl];
                double tot_;
                const double pre = row_scan16(w, cin, tot_);
                const int q = row_min(any && slot * 16 + rl < P && !(pre < rr_) ? rl : 16);
                double total, wi;
                if (q < 16) { pi = slot * 16 + q; total = rowget_f64(pre, q); wi = rowget_f64(w, q); }
                else { pi = P - 1; total = c3; wi = s_inf[P - 1]; }       // clamp at n-1 (fc:26)
                if (evn && wi == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 1;
                rn = (rr_ - (total - wi)) / wi;
                choose = rn * wi;                    // pyx:493: rn * popRate[pi]
            }
            // immunePopRate[pi] = +0.0 is never > choose: the infect branch (pyx:499-500)
            const double infect_pi = s_inf[pi];
            rn = (choose - 0.0) / infect_pi;
            const double bC = s_bc[pi];
            const double smul = k_smul[pi];
            const double tE = ((bC + c_d) + smul) + c_tm;          // tEventHapPopRate (pyx:522-526)
            const int n_sel = evn ? s_nocc[pi] : 0;
            const int32_t *lh = lhap + (int64_t)pi * cap;
            int64_t *ln = lcnt + (int64_t)pi * cap;
            int32_t *l3 = lcnt32 + (int64_t)pi * cap;
            int64_t *lt = ltsum + (int64_t)pi * capT;
            {   // row pi of the migration matrix for the BirthRate refresh: in flight together with the list
                const double *mrow = p.mig + (int64_t)pi * P;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < nslot) m_pre[s] = mrow[min(s * 16 + rl, P - 1)];
                pm_pi = evn ? pi : -1;
            }

            QPROF(4);
            // ---- haplotype by fastChoose over hapPopRate[pi] = tE * infectious in haplotype order ----
            const double r2 = infect_pi * rn;
            int k_hit = -1;          // list index of the chosen entry
            double pre_hit = 0.0, w_hit = 0.0;
            int hap_hit = 0;
            int64_t cnt_hit = 0;
            const int maxn = rows_max(n_sel);
            if (__builtin_expect(maxn <= 64, QT < 4)) {
                // the lists fit four register chunks: all loads in flight together (unconditional, on clamped indices)
                const int nch = (maxn + 15) >> 4;
                const int last = max(n_sel - 1, 0);
                int64_t cn4[4];
                int hp4[4];
                double w4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    cn4[c] = 0; hp4[c] = 0; w4[c] = 0.0;
                    if (c < nch) {
                        const int k = c * 16 + rl;
                        cn4[c] = QT >= 4 ? (int64_t)l3[min(k, last)] : ln[min(k, last)];    // (the long-list kernel keeps the 4-byte counts only)
                        hp4[c] = lh[min(k, last)];
                        w4[c] = k < n_sel ? tE * (double)cn4[c] : 0.0;
                    }
                }
                // running sum chunk by chunk; the chunk in which it first reaches r is then scanned lane by lane
                double carry = 0.0, carry_hit = 0.0;
                int c_hit = nch == 1 ? 0 : -1;
                if (nch > 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c < nch) {
                            const double acc = row_sum16(w4[c], carry);
                            if (c_hit < 0 && c * 16 < n_sel && !(acc < r2)) { c_hit = c; carry_hit = carry; }
                            carry = acc;
                        }
                    }
                }
                const int cc = c_hit >= 0 ? c_hit : (last >> 4);    // rows without a hit: their last chunk (H-1 rule)
                const double w = cc == 0 ? w4[0] : cc == 1 ? w4[1] : cc == 2 ? w4[2] : w4[3];
                const int64_t cn = cc == 0 ? cn4[0] : cc == 1 ? cn4[1] : cc == 2 ? cn4[2] : cn4[3];
                const int hp = cc == 0 ? hp4[0] : cc == 1 ? hp4[1] : cc == 2 ? hp4[2] : hp4[3];
                double tot_;
                const double pre = row_scan16(w, c_hit >= 0 ? carry_hit : 0.0, tot_);
                if (nch == 1) carry = tot_;
                const int q = row_min(c_hit >= 0 && cc * 16 + rl < n_sel && !(pre < r2) ? rl : 16);
                const int qq = q < 16 ? q : (last & 15);
                pre_hit = q < 16 ? rowget_f64(pre, qq) : carry;      // no hit: the total of the whole list
                w_hit = rowget_f64(w, qq);
                hap_hit = rowget_i32(hp, qq); cnt_hit = rowget_i64(cn, qq);
                if (q < 16) k_hit = cc * 16 + q;
                else if (evn) {
                    // nothing reached r: the dense loop runs on to index H-1 (fc:26), a valid pick only if that haplotype
                    // is occupied, otherwise the reference reports a zero weight
                    if (n_sel > 0 && hap_hit == H - 1 && (QT < 4 || cnt_hit != 0)) k_hit = n_sel - 1; else err = Q_ERR_ZERO_WEIGHT + 256 * 2;
                }
                if (evn) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ch_cn[c] = cn4[c];
                    ch_pi = pi;
                }
            } else {
#ifdef VGX_PROFILE
                prof_acc[15] += 1;       // (diagnostic build: iterations whose haplotype choice took the long-list form)
#endif
                QSel sel;
                q_long_select<QT>(l3, lh, n_sel, maxn, tE, r2, evn, H, sel, (double *)(ltsum + (int64_t)pi * capT + R * P * capT))   /* the cached running sums lie behind the tile sums (vgx_dev.h) */;
                k_hit = sel.k_hit; pre_hit = sel.pre_hit; w_hit = sel.w_hit; hap_hit = sel.hap_hit; cnt_hit = sel.cnt_hit;
                if (sel.err) err = sel.err;
            }
            const bool evn_ok = evn && err == 0;
            if (evn && w_hit == 0.0 && err == 0) err = Q_ERR_ZERO_WEIGHT + 256 * 4;
            rn = (r2 - (pre_hit - w_hit)) / w_hit;

            QPROF(5);
            // ---- event class by fastChoose over (birth, death, sampling, mutation) rates (pyx:503-511) ----
            int ei = 0;
            {
                const double r3 = tE * rn;
                double total = bC, wi = bC;
                if (total < r3) { ei = 1; total += c_d; wi = c_d; }
                if (ei == 1 && total < r3) { ei = 2; total += smul; wi = smul; }
                if (ei == 2 && total < r3) { ei = 3; total += c_tm; wi = c_tm; }
                if (evn_ok && wi == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 5;
                rn = (r3 - (total - wi)) / wi;
            }
            const bool go = evn && err == 0;
            const int64_t ts_pi = s_ts[pi], ti_pi = s_ti[pi];
            const bool isB = go && ei == 0, isD = go && (ei == 1 || ei == 2), isM = go && ei == 3;
            if (isB) {
                // ---- Birth (pyx:568-605; one susceptibility group: si = 0, its weight susceptHapPopRate = S * sigma) ----
                if ((double)ts_pi * c_sig == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 6;
                if (rl == 0) { s_ts[pi] = ts_pi - 1; s_ti[pi] = ti_pi + 1; }
                gI += 1; QBUMP(QC_B);
                if (live && rl == 0) { if (QT < 4) ln[k_hit] = cnt_hit + 1; l3[k_hit] = (int32_t)(cnt_hit + 1); if (QT >= 4) L8(pi)[k_hit] = B8(cnt_hit + 1); if (n_sel > 64) lt[k_hit >> 6] += 1; }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c == (k_hit >> 4) && rl == (k_hit & 15)) ch_cn[c] += 1;
                e_type = QEV_BIRTH; e_hap = hap_hit; e_pop = pi; e_nh = 0; e_np = H;
                u_lo = pi; u_hi = pi + 1;
            }
            if (isD) {
                // ---- Death / Sampling (pyx:616-635): recovery into group suscType = 0 ----
                if (rl == 0) { s_ts[pi] = ts_pi + 1; s_ti[pi] = ti_pi - 1; }
                gI -= 1;
                if (ei == 2) { cS += 1; e_type = QEV_SAMPLING; } else { QBUMP(QC_D); e_type = QEV_DEATH; }
                // (the long-list kernel leaves a count of 0 in the list: vgx_rowlist.h)
                if (QT < 4 && cnt_hit == 1) { op_n = 1; op_pi = pi; op_h0 = hap_hit; op_d0 = -1; ch_pi = -1; }
                else {
                    if (live && rl == 0) { if (QT < 4) ln[k_hit] = cnt_hit - 1; l3[k_hit] = (int32_t)(cnt_hit - 1); if (QT >= 4) L8(pi)[k_hit] = B8(cnt_hit - 1); if (n_sel > 64) lt[k_hit >> 6] -= 1; }
                    if (QT >= 4 && cnt_hit == 1 && rl == 0) s_zero[pi] += 1;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c == (k_hit >> 4) && rl == (k_hit & 15)) ch_cn[c] -= 1;
                }
                e_hap = hap_hit; e_pop = pi; e_nh = 0; e_np = 0;
                u_lo = pi; u_hi = pi + 1;
            }
            QPROF(6);
            if (__builtin_expect(__ballot(isM) != 0, 0)) {   // slow paths: a few per cent of the iterations
                // ---- Mutation (pyx:640-667): site by mRate[h, :], derived state by hapMutType[h, site, :] ----
                const double *mr = p.mRate + (int64_t)hap_hit * sites;
                int mi = 0;
                {
                    const double rq = c_tm * rn;
                    double total = isM ? mr[0] : 1.0, wi = total;
                    for (int i = 1; i < sites; ++i) {
                        if (isM && mi == i - 1 && total < rq) { mi = i; wi = mr[i]; total += wi; }
                    }
                    if (isM && wi == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 7;
                    rn = (rq - (total - wi)) / wi;
                }
                const double *hm = p.hapMutType + ((int64_t)hap_hit * sites + mi) * 3;
                int DS = 0;
                if (isM) {
                    const double h0 = hm[0], h1 = hm[1], h2 = hm[2];
                    const double rq = ((h0 + h1) + h2) * rn;
                    double total = h0, wi = h0;
                    if (total < rq) { DS = 1; total += h1; wi = h1; }
                    if (DS == 1 && total < rq) { DS = 2; total += h2; wi = h2; }
                    if (wi == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 8;
                }
                if (isM && err == 0) {
                    const int digit4 = 1 << (2 * (sites - mi - 1));     // Mutate (pyx:2420-2427)
                    const int AS = (hap_hit / digit4) % 4;
                    if (DS >= AS) DS += 1;
                    const int nhi = hap_hit + (DS - AS) * digit4;
                    if (QT >= 4) {      // the source's count drops in place (0 stays in the list), the mutant is the one deferred insertion
                        if (live && rl == 0) { l3[k_hit] = (int32_t)(cnt_hit - 1); L8(pi)[k_hit] = B8(cnt_hit - 1); if (n_sel > 64) lt[k_hit >> 6] -= 1; if (cnt_hit == 1) s_zero[pi] += 1; }
                        op_n = 1; op_pi = pi; op_h0 = nhi; op_d0 = +1; ch_pi = -1;
                    } else {
                        op_n = 2; op_pi = pi; op_h0 = nhi; op_d0 = +1; op_h1 = hap_hit; ch_pi = -1;
                    }
                    QBUMP(QC_M);
                    e_type = QEV_MUTATION; e_hap = hap_hit; e_pop = pi; e_nh = nhi; e_np = 0;
                    u_lo = pi; u_hi = pi + 1;
                }
            }
            QPROF(7);
            if (__builtin_expect(__ballot(evm) != 0, 0)) {
                // ================= GenerateMigration (pyx:672-694) =================
                double rm = (choose0 - totalRate) / totalMig;
                // target population by fastChoose over migPopRate: serial prefix sums of the same terms as totalMigrationRate
                int tpi = 0;
                {
                    const double rr_ = totalMig * rm;
                    double carry = 0.0, tot_hit = 0.0, w_h = 0.0;
                    int cand = 64;
                    for (int s = 0; s < nslot; ++s) {
                        const int pn = s * 16 + rl;
                        const double w = pn < P ? k_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]) : 0.0;
                        double tot_;
                        const double pre = row_scan16(w, carry, tot_);
                        const int q = row_min(cand == 64 && pn < P && !(pre < rr_) ? rl : 16);
                        if (cand == 64 && q < 16) { cand = s * 16 + q; tot_hit = rowget_f64(pre, q); w_h = rowget_f64(w, q); }
                        else if (s == nslot - 1 && cand == 64) {        // clamp at P-1
                            const int ql = (P - 1) & 15;
                            tot_hit = rowget_f64(pre, ql); w_h = rowget_f64(w, ql);
                        }
                        carry = tot_;
                    }
                    tpi = cand < 64 ? cand : P - 1;
                    if (evm && w_h == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 9;
                    rm = (rr_ - (tot_hit - w_h)) / w_h;
                }
                // source population: fastChoose_skip(totalInfectious, globalInfectious - totalInfectious[tpi], rn, skip = tpi)
                int spi = -1;
                {
                    const double rr_ = (double)(gI - s_ti[tpi]) * rm;
                    const int start = tpi == 0 ? 1 : 0;
                    int64_t carry = 0, total = 0;
                    for (int s = 0; s < nslot; ++s) {
                        const int pn = s * 16 + rl;
                        const bool in = pn < P && pn != tpi && pn >= start;
                        const int64_t w = in ? s_ti[pn] : 0;
                        const int64_t pre = row_iscan(w) + carry;
                        const int q = row_min(spi < 0 && in && !((double)pre < rr_) ? rl : 16);
                        if (spi < 0 && q < 16) { spi = s * 16 + q; total = rowget_i64(pre, q); }
                        carry = rowget_i64(pre, 15);
                    }
                    if (spi < 0) { spi = P - 1; total = carry; }   // clamp at n-1 (may equal skip only then)
                    const int64_t wi = s_ti[spi];
                    if (evm && wi == 0) err = Q_ERR_ZERO_WEIGHT + 256 * 10;
                    rm = (rr_ - (double)(total - wi)) / (double)wi;
                }
                // haplotype by fastChoose(infectious[spi], totalInfectious[spi], rn): int64 weights over the occupancy list
                // (integer prefix sums are order-free: tile sums pick the 64-entry tile, then its four chunks)
                int hi = 0;
                {
                    const int n = evm ? s_nocc[spi] : 0;
                    const int32_t *lh2 = lhap + (int64_t)spi * cap;
                    const int64_t *ln2 = lcnt + (int64_t)spi * cap;
                    const int32_t *l32 = lcnt32 + (int64_t)spi * cap;
#define CN2(k) (QT >= 4 ? (int64_t)l32[k] : ln2[k])      /* (the long-list kernel keeps the 4-byte counts only) */
                    const int64_t *lt2 = ltsum + (int64_t)spi * capT;
                    const double rr_ = (double)s_ti[spi] * rm;
                    int64_t before = 0;
                    int base = 0;
                    bool none = false;
                    const int maxn2 = rows_max(n);
                    if (__builtin_expect(maxn2 > 64, QT >= 4)) {
                        const int nt = n > 64 ? (n + 63) >> 6 : 0;     // tile sums exist only for lists longer than a tile
                        const int maxt = rows_max(nt);
                        int jt = -1;
                        int64_t carry = 0;
                        for (int tb = 0; tb < maxt; tb += 16) {
                            const int j = tb + rl;
                            const int64_t w = j < nt ? lt2[j] : 0;
                            const int64_t pre = row_iscan(w) + carry;
                            const int q = row_min(jt < 0 && j < nt && !((double)pre < rr_) ? rl : 16);
                            if (jt < 0 && q < 16) { jt = tb + q; before = rowget_i64(pre, q) - rowget_i64(w, q); }
                            carry = rowget_i64(pre, 15);
                        }
                        if (nt > 0) { if (jt < 0) { none = true; before = carry; } else base = jt * 64; }
                    }
                    int kq = -1;
                    int64_t total = before, wi = 0;
                    {
                        int64_t carry = before;
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const int k = base + c4 * 16 + rl;
                            const bool in = !none && k < n;
                            const int64_t w = in ? CN2(k) : 0;
                            const int64_t pre = row_iscan(w) + carry;
                            const int q = row_min(kq < 0 && in && (QT < 4 || w != 0) && !((double)pre < rr_) ? rl : 16);
                            if (kq < 0 && q < 16) { kq = k - rl + q; total = rowget_i64(pre, q); wi = rowget_i64(w, q); }
                            carry = rowget_i64(pre, 15);
                            if (!__ballot(evm && kq < 0 && !none && base + (c4 + 1) * 16 < n)) break;
                        }
                        if (kq < 0) total = carry;
                    }
                    if (evm && kq < 0) {
                        if (n > 0 && lh2[n - 1] == H - 1 && (QT < 4 || CN2(n - 1) != 0)) { kq = n - 1; wi = CN2(n - 1); }
                        else {
                            err = Q_ERR_ZERO_WEIGHT + 256 * 11;
                            if (rl == 0 && r.prof) {
                                unsigned long long *d = r.prof + rep * VGX_PROF_SLOTS;
                                d[0] = n; d[1] = spi; d[2] = tpi; d[3] = s_ti[spi]; d[4] = __double_as_longlong(rr_); d[5] = __double_as_longlong(rm);
                                d[6] = total; d[7] = before; d[8] = maxn2; d[9] = gI; d[10] = s_ti[tpi]; d[11] = n > 0 ? CN2(0) : -1;
                                d[12] = ev_ptr; d[13] = loops; d[14] = __double_as_longlong(totalMig); d[15] = __double_as_longlong(choose);
                            }
                            kq = 0; wi = 1;
                        }
                    }
                    if (kq < 0) { kq = 0; wi = 1; }
                    rm = (rr_ - (double)(total - wi)) / (double)wi;
                    hi = (evm && n > 0) ? lh2[kq] : 0;
                }
                // susceptibility group of the target (one group): fastChoose(susceptible[tpi, :], totalSusceptible[tpi], rn)
                {
                    const int64_t wi = s_ts[tpi];
                    const double rr_ = (double)wi * rm;
                    if (evm && wi == 0 && err == 0) err = Q_ERR_ZERO_WEIGHT + 256 * 12;
                    rm = (rr_ - (double)(wi - wi)) / (double)wi;
                }
                const bool mgo = evm && err == 0;
                if (mgo) {
                    const double p_accept = qa.effMig[(int64_t)spi * P + tpi] * p.bRate[hi] * p.susc[hi] / k_mebm[tpi];
                    if (rm < p_accept) {
                        if (rl == 0) { s_ts[tpi] -= 1; s_ti[tpi] += 1; }     // NewInfections (pyx:246-251)
                        gI += 1; QBUMP(QC_MIGP);
                        op_n = 1; op_pi = tpi; op_h0 = hi; op_d0 = +1; ch_pi = -1;
                        e_type = QEV_MIGRATION; e_hap = hi; e_pop = spi; e_nh = 0; e_np = tpi;
                        u_lo = tpi; u_hi = tpi + 1;
                    } else {
                        QBUMP(QC_MIGN);
                    }
                }
            }
            WSYNC();
        }

        QPROF(8);
        // ================= deferred list operations: infectious[op_pi, hap] += delta, list kept ordered =================
        if (err != 0) op_n = 0;
        if (QT >= 4) {      // the long-list kernel: lists with zero-count entries, one deferred insertion at most (vgx_rowlist.h)
            if (__builtin_expect(__ballot(live && op_n > 0) != 0, 0)) {
                const bool act = live && op_n > 0;
                int n = act ? s_nocc[op_pi] : 0;
                const int n_was = n;
                WSYNC();
                int zeros = act ? s_zero[op_pi] : 0;
                const int z_was = zeros;
                const bool fits = q_list_add_one<true>(act, op_h0, lhap + (int64_t)op_pi * cap, lcls + (int64_t)op_pi * cap, lcnt32 + (int64_t)op_pi * cap,
                                                       ltsum + (int64_t)op_pi * capT, L8(op_pi), n, (int)cap, H, zeros);
                if (act && !fits) err = Q_ERR_CAPACITY;
                if (act && rl == 0) { if (n != n_was) s_nocc[op_pi] = n; if (zeros != z_was) s_zero[op_pi] = max(zeros, 0); }
                WSYNC();
            }
            // a list whose zero-count entries pile up (every one is a step of the refresh chain) is squeezed before the refresh below
            {
                WSYNC();
                const int zp = (live && err == 0 && u_hi == u_lo + 1) ? u_lo : -1;      // the population this event changed
                const int nz = zp >= 0 ? s_zero[zp] : 0, nl = zp >= 0 ? s_nocc[zp] : 0;
                const bool sq = zp >= 0 && nz > max(16, nl >> 3);
                if (__builtin_expect(__ballot(sq) != 0, 0)) {
                    const int zq = sq ? zp : 0;
                    int32_t *l3z = lcnt32 + (int64_t)zq * cap;
                    const int n2 = q_compact_list<1>(lhap + (int64_t)zq * cap, l3z, ltsum + (int64_t)zq * capT, nl, sq);
                    uint8_t *l8z = L8(zq);
                    const int m8 = rows_max(sq ? n2 : 0);
                    for (int k = rl; k < m8; k += 16) if (sq && k < n2) l8z[k] = B8(l3z[k]);
                    if (sq && rl == 0) { s_nocc[zq] = n2; s_zero[zq] = 0; }
                    if (sq && ch_pi == zq) ch_pi = -1;
                    WSYNC();
                }
            }
        } else
        for (int oi = 0; oi < 2; ++oi) {
            const bool act = live && oi < op_n;
            if (!__ballot(act)) break;
            const int hap = oi == 0 ? op_h0 : op_h1;
            const int delta = oi == 0 ? op_d0 : -1;
            const int n = act ? s_nocc[op_pi] : 0;
            int32_t *lh = lhap + (int64_t)op_pi * cap;
            int32_t *lc = lcls + (int64_t)op_pi * cap;
            int64_t *ln = lcnt + (int64_t)op_pi * cap;
            int32_t *l3 = lcnt32 + (int64_t)op_pi * cap;
            int64_t *lt = ltsum + (int64_t)op_pi * capT;
            // ---- lower bound: first index whose haplotype is >= hap ----
            int posn = 0;
            bool found = false;
            int64_t cur = 0;
            {
                int lo = 0;             // first entry of the 16^k-aligned window known to contain the bound
                const int maxn = rows_max(n);
                // 16-ary descent over the sorted list: strides 16^5 ... 16, 1
                for (int stride = 1 << 20; stride >= 1; stride >>= 4) {
                    if (stride >= 16 && maxn <= stride) continue;
                    const int k = lo + rl * stride;
                    const int h = (act && k < n) ? lh[k] : 0x7fffffff;
                    // probes are sorted: the first lane with h > hap = the number of probes <= hap; the bound lies at or
                    // after the last of those and before the next probe
                    const int nle = row_min(h <= hap ? 16 : rl);
                    if (stride == 1) {
                        const int q = row_min(h >= hap ? rl : 16);
                        posn = lo + q;
                        if (posn > n) posn = n;
                        const int hq = rowget_i32(h, min(q, 15));
                        found = q < 16 && hq == hap;
                    } else {
                        lo = lo + (nle > 0 ? (nle - 1) * stride : 0);
                    }
                }
                if (act && found) cur = ln[posn];
            }
            const bool bump = act && found && cur + delta != 0;       // count changes in place
            const bool rem = act && found && cur + delta == 0;        // the entry disappears
            const bool ins = act && !found;                           // a new entry (delta = +1)
            if (ins && n >= cap) { err = Q_ERR_CAPACITY; }
            const bool ins_ok = ins && err == 0;
            if (bump && rl == 0) { ln[posn] = cur + delta; l3[posn] = (int32_t)(cur + delta); if (QT >= 4) L8(op_pi)[posn] = B8(cur + delta); if (n > 64) lt[posn >> 6] += delta; }
            // ---- tile sums of lists longer than one tile (vgx_direct.hip list_insert_at / list_remove_at) ----
            if (__builtin_expect(__ballot((ins_ok || rem) && n > 64) != 0, 0)) {
                const bool tt = (ins_ok || rem) && n > 64;
                const int jp = posn >> 6, jl = ins_ok ? (n >> 6) : ((n - 1) >> 6);
                const int maxj = rows_max(tt ? jl + 1 : 0);
                for (int tb = 0; tb < maxj; tb += 16) {
                    const int j = tb + rl;
                    if (tt && j >= jp && j <= jl) {
                        int64_t in_, out_;
                        if (ins_ok) {
                            in_ = j == jp ? (int64_t)delta : ln[(int64_t)j * 64 - 1];
                            const int kout = j * 64 + 63;
                            out_ = kout < n ? ln[kout] : 0;
                        } else {
                            out_ = j == jp ? ln[posn] : ln[(int64_t)j * 64];
                            const int kin = j * 64 + 64;
                            in_ = kin < n ? ln[kin] : 0;
                        }
                        lt[j] += in_ - out_;
                    }
                }
                WSYNC();
            }
            // ---- shift: insertion moves [posn, n) one slot up (highest block first), removal (posn, n) one slot down ----
            if (__builtin_expect(__ballot(ins_ok) != 0, 0)) {
                enum { SU = 4 };
                int hi_ = ins_ok ? n : 0;
                const int lo_ = ins_ok ? posn : 0;
                while (__ballot(hi_ > lo_)) {
                    const int blo = max(lo_, hi_ - SU * 16);
                    int h[SU];
                    int64_t ct[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = blo + u * 16 + rl;
                        h[u] = 0; ct[u] = 0;
                        if (k < hi_) { h[u] = lh[k]; ct[u] = ln[k]; }
                    }
                    WSYNC();
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = blo + u * 16 + rl;
                        if (k < hi_) { lh[k + 1] = h[u]; lc[k + 1] = 0; ln[k + 1] = ct[u]; l3[k + 1] = (int32_t)ct[u]; if (QT >= 4) L8(op_pi)[k + 1] = B8(ct[u]); }
                    }
                    WSYNC();
                    hi_ = blo;
                }
                if (ins_ok && rl == 0) { lh[posn] = hap; lc[posn] = 0; ln[posn] = delta; l3[posn] = delta; if (QT >= 4) L8(op_pi)[posn] = B8(delta); s_nocc[op_pi] = n + 1; }
                WSYNC();
                if (ins_ok && n == 64) {   // the list outgrows one tile: start its tile sums
                    int64_t s0 = 0;
                    for (int c4 = 0; c4 < 4; ++c4) s0 += rowget_i64(row_iscan(ln[c4 * 16 + rl]), 15);
                    if (rl == 0) { lt[0] = s0; lt[1] = ln[64]; }
                }
                WSYNC();
            }
            if (__builtin_expect(__ballot(rem) != 0, 0)) {
                enum { SU = 4 };
                int lo_ = rem ? posn + 1 : 0;
                const int hi_ = rem ? n : 0;
                while (__ballot(lo_ < hi_)) {
                    int h[SU];
                    int64_t ct[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = lo_ + u * 16 + rl;
                        h[u] = 0; ct[u] = 0;
                        if (k < hi_) { h[u] = lh[k]; ct[u] = ln[k]; }
                    }
                    WSYNC();
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = lo_ + u * 16 + rl;
                        if (k < hi_) { lh[k - 1] = h[u]; ln[k - 1] = ct[u]; l3[k - 1] = (int32_t)ct[u]; if (QT >= 4) L8(op_pi)[k - 1] = B8(ct[u]); }
                    }
                    WSYNC();
                    lo_ += SU * 16;
                }
                if (rem && rl == 0) s_nocc[op_pi] = n - 1;
                WSYNC();
            }
        }

        QPROF(9);
        // ================= Events.AddEvent (events.pxi:37-44) =================
        if (err == 0 && e_type >= 0) {
            if (a.record_events) {
                const int64_t slot = ev_ptr - r.ev_base;
                if (slot >= 0 && slot < r.evcap) {
                    if (stage_n == 0) stage_slot0 = slot;
                    if (rl < 8) {
                        const int v = rl == 0 ? e_type : rl == 1 ? e_hap : rl == 2 ? e_pop : rl == 3 ? e_nh : rl == 4 ? e_np
                                      : rl == 5 ? (int)(uint32_t)att_loops : rl == 6 ? __double2loint(den) : __double2hiint(den);
                        s_stage[rl < 6 ? stage_n * 6 + rl : 48 + 2 * stage_n + (rl - 6)] = (uint32_t)v;
                    }
                    stage_n += 1;
                } else {
                    err = Q_ERR_CAPACITY;
                }
            }
            ev_ptr += 1;
        }
        if (__builtin_expect(__ballot(stage_n == 8) != 0, 0)) stage_flush(stage_n == 8);

        QPROF(10);
        // ================= UpdateRates for [u_lo, u_hi) (pyx:516-546) / UpdateAllRates (pyx:279-351) =================
        if (err != 0) u_hi = u_lo;
        const int maxu = rows_max(u_hi - u_lo);
        if (maxu > 0) {
            for (int us = 0; us < maxu; ++us) {
                const int pn0 = u_lo + us;
                const bool act = pn0 < u_hi;
                const int pi = act ? pn0 : 0;
                // BirthRate of the class (pyx:382-392): ps += ((x*m)*m*cd)/as over the source populations, in order
                const double x = (double)s_ts[pi] * c_sig;
                const double *mrow = p.mig + (int64_t)pi * P;
                const bool mhave = act && pi == pm_pi;
                double tv4[4], ps = 0.0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    tv4[s] = 0.0;
                    if (s < nslot) {
                        const int pn = s * 16 + rl;
                        const double m = mhave ? m_pre[s] : mrow[min(pn, P - 1)];
                        tv4[s] = x * m * m * k_cd[pn] / k_as[pn];        // lanes beyond P: cd = +0.0, as = 1.0
                        if (pn >= P) tv4[s] = 0.0;
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < nslot) ps = row_sum16(tv4[s], ps);
                const double bC = c_b * ps;
                const double tE = ((bC + c_d) + k_smul[pi]) + c_tm;
                QPROF(11);
                // infectPopRate[pi]: tE * infectious over the occupied haplotypes, in haplotype order (pyx:519-528)
                const int n = act ? s_nocc[pi] : 0;
                const int64_t *ln = lcnt + (int64_t)pi * cap;
                const int32_t *l3 = lcnt32 + (int64_t)pi * cap;
                const int maxn = rows_max(n);
                double acc = 0.0;
                if (__builtin_expect(maxn <= 64, QT < 4)) {
                    const int nch = (maxn + 15) >> 4;
                    const bool chave = act && pi == ch_pi;     // the list read for the haplotype choice, event applied
                    int64_t cn4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) cn4[c] = ch_cn[c];
                    if (__ballot(act && !chave)) {             // some row refreshes a population it did not just read
                        const int last = max(n - 1, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < nch) { const int64_t cl = QT >= 4 ? (int64_t)l3[min(c * 16 + rl, last)] : ln[min(c * 16 + rl, last)]; if (!chave) cn4[c] = cl; }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < nch) acc = row_sum16(c * 16 + rl < n ? tE * (double)cn4[c] : 0.0, acc);
                } else {
                    acc = q_long_sum<QT>(l3, L8(pi), n, maxn, tE, (double *)(ltsum + (int64_t)pi * capT + R * P * capT));   // (the cached running sums lie behind the tile sums, vgx_dev.h)
                }
                if (act && rl == 0) { s_bc[pi] = bC; s_inf[pi] = acc; }
                WSYNC();
                QPROF(12);
            }
            // totalRate and the serial prefix sums of popRate (pyx:537-539)
            {
                double w4[4], carry = 0.0, cend[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) w4[s] = s_inf[s * 16 + rl];      // lanes beyond P hold +0.0
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < nslot) carry = row_sum16(w4[s], carry);
                    cend[s] = carry;
                }
                if (u_hi > u_lo) {
                    if (rl < 4) s_cc[rl] = rl == 0 ? cend[0] : rl == 1 ? cend[1] : rl == 2 ? cend[2] : cend[3];
                    totalRate = carry;       // = the prefix sum at P-1: the lanes beyond P add +0.0
                }
                WSYNC();
            }
            QPROF(13);
            // totalMigrationRate = sum of maxEffectiveBirthMigration * totalSusceptible * (globalInfectious - totalInfectious)
            if (has_mig) {
                double w4[4], acc = 0.0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    w4[s] = 0.0;
                    if (s < nslot) {
                        const int pn = s * 16 + rl;     // lanes beyond P: maxEBM = +0.0, counts 0
                        w4[s] = k_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]);
                        if (pn >= P) w4[s] = 0.0;
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < nslot) acc = row_sum16(w4[s], acc);
                if (u_hi > u_lo) totalMig = acc;
            }
        }

        QPROF(14);
        // ================= after the pass =================
        if (rebuild && st == ST_REBUILD) st = err ? ST_DONE : ST_RUN;
        if (err != 0) st = ST_DONE;
        if (ev && st == ST_RUN && (totalRate == 0.0 || gI == 0)) end_attempt = true;   // pyx:410-411
        if (st == ST_RUN && end_attempt) {
            // end of an attempt (pyx:414-418)
            open = false;
            if (ev_ptr <= 100 && a.iterations > 100) {
                // Restart (pyx:714-738): compartments back to the initial snapshot, then UpdateAllRates
                ev_ptr = 0; cS = 0;
                stage_n = 0;                      // (the failed attempt's records are dropped with its log)
                if (rl < 6) s_cnt[rl] = 0;
                t_now = 0.0; traj_next = 0;
                restarts += 1; att += 1;
                st = ST_REBUILD;
            } else {
                good_attempt = (int64_t)att + 1;
                st = ST_DONE;
            }
        }
        if (__builtin_expect(__ballot(st == ST_REBUILD && restarts > 0 && !rebuild) != 0, 0)) {
            const bool rs = st == ST_REBUILD && restarts > 0 && !rebuild && live;
            int64_t g = 0;
            for (int pn = 0; pn < P; ++pn) {
                const int n = r.i_nocc[pn];
                const int n_old = rs ? s_nocc[pn] : 0;
                int64_t ti = 0;
                for (int base = 0; base < n; base += 64) {
                    int64_t tsum = 0;
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const int k = base + c4 * 16 + rl;
                        int64_t ct = 0;
                        if (k < n) {
                            ct = r.i_cnt[(int64_t)pn * r.i_cap + k];
                            if (rs) {
                                lhap[(int64_t)pn * cap + k] = r.i_hap[(int64_t)pn * r.i_cap + k];
                                lcls[(int64_t)pn * cap + k] = r.i_cls[(int64_t)pn * r.i_cap + k];
                                lcnt[(int64_t)pn * cap + k] = ct;
                                lcnt32[(int64_t)pn * cap + k] = (int32_t)ct;
                                if (QT >= 4) L8(pn)[k] = B8(ct);
                            }
                        }
                        tsum += rowget_i64(row_iscan(ct), 15);
                    }
                    if (rs && rl == 0) ltsum[(int64_t)pn * capT + base / 64] = tsum;
                    ti += tsum;
                }
                if (rs)
                    for (int j = (n + 63) / 64 + rl; j <= n_old / 64 && j < capT; j += 16) ltsum[(int64_t)pn * capT + j] = 0;
                if (rs && rl == 0) { s_nocc[pn] = n; s_zero[pn] = 0; s_ts[pn] = r.i_sus[pn]; s_ti[pn] = ti; }
                g += ti;
            }
            if (rs) gI = g;
            WSYNC();
        }
    }

    // trailing grid points of the trajectories: the final state
    if (has_traj) {
        while (true) {
            const bool emit = live && traj_next < r.traj_points;
            if (!__ballot(emit)) break;
            if (emit) {
                double *o = r.traj + (rep * r.traj_points + traj_next) * (int64_t)P * 2;
                for (int s = 0; s < nslot; ++s) {
                    const int pn = s * 16 + rl;
                    if (pn < P) { o[pn * 2 + 0] = (double)s_ti[pn]; o[pn * 2 + 1] = (double)s_ts[pn]; }
                }
                traj_next += 1;
            }
        }
    }

#ifdef VGX_PROFILE
    if (lane == 0 && r.prof)
        for (int i = 0; i < VGX_PROF_SLOTS; ++i) r.prof[rep * VGX_PROF_SLOTS + i] = prof_acc[i];
#endif
    // ---- state back to HBM ----
    stage_flush(stage_n > 0);
    WSYNC();
    if (live) {
        double *gD = r.popD + rep * PD_COUNT * P;
        int64_t *gI64 = r.popI + rep * PI_COUNT * P;
        int32_t *gN = r.nocc + rep * P;
        VgxRepScalars *sc = r.sc + rep;
        for (int s = 0; s < nslot; ++s) {
            const int pn = s * 16 + rl;
            if (pn < P) {
                gD[PD_POPRATE * P + pn] = s_inf[pn];
                gD[PD_INFECT * P + pn] = s_inf[pn];
                gD[PD_IMMUNE * P + pn] = 0.0;
                gD[PD_MIG * P + pn] = k_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]);
                gD[PD_MAXEBM * P + pn] = k_mebm[pn];
                gI64[PI_TOTSUS * P + pn] = s_ts[pn];
                gI64[PI_TOTINF * P + pn] = s_ti[pn];
                gN[pn] = s_nocc[pn];
                r.sus[rep * P + pn] = s_ts[pn];
                r.immSrc[rep * P + pn] = 0.0;
            }
        }
        if (rl == 0) {
            sc->currentTime = t_now; sc->totalRate = totalRate; sc->totalMig = totalMig;
            sc->globalInfectious = gI;
            sc->bCounter = s_cnt[QC_B]; sc->dCounter = s_cnt[QC_D]; sc->sCounter = cS; sc->mCounter = s_cnt[QC_M];
            sc->migPlus = s_cnt[QC_MIGP]; sc->migNonPlus = s_cnt[QC_MIGN];
            sc->good_attempt = good_attempt;
            sc->ev_ptr = ev_ptr; sc->loop_iterations = loops; sc->restarts = restarts;
            sc->loc_n = 0; sc->error = err; sc->traj_next = traj_next;
            sc->last_attempt = last_att; sc->last_attempt_loops = att_loops;
            sc->fa_n = 0;
        }
    }
}

#ifndef VGX_QT_SHORT
#define VGX_QT_SHORT 1   // measured at the headline workload: 3 -> 7.9e8, 2 -> 8.2e8, 1 -> 8.6e8 events/s (8.8e8 with the branch hints)
#endif
extern "C" __global__ void __launch_bounds__(64, VGX_QUAD_WAVES) vgx_quad_kernel(VgxDirectArgs a, QArgs qa) { quad_body<VGX_QT_SHORT>(a, qa); }
#ifndef VGX_QT_LONG
#define VGX_QT_LONG 7     // (254 VGPRs, no spill; 5: 1.31e8, 7: 1.33e8 events/s at 4096-entry lists)
#endif
extern "C" __global__ void __launch_bounds__(64, VGX_QUAD_WAVES) vgx_quad_long_kernel(VgxDirectArgs a, QArgs qa) { quad_body<VGX_QT_LONG>(a, qa); }

// The 4-byte copy of the counts after another kernel changed the lists (the copy is kept by vgx_quad_kernel only).
extern "C" __global__ void __launch_bounds__(256) vgx_quad_counts32_kernel(const int64_t *c64, int32_t *c32, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c32[i] = (int32_t)c64[i];
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_counts32(const int64_t *c64, int32_t *c32, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(vgx_quad_counts32_kernel, dim3(4096), dim3(256), 0, stream, c64, c32, n);
    return hipGetLastError();
}

// The one-byte copy behind the 4-byte one (vgx_quad_long_kernel streams it): before every launch of that kernel.
extern "C" __global__ void __launch_bounds__(256) vgx_quad_counts8_kernel(const int32_t *c32, uint8_t *c8, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int4 v = *(const int4 *)(c32 + 4 * i);
        *(uint32_t *)(c8 + 4 * i) = (uint32_t)min(max(v.x, 0), 255) | ((uint32_t)min(max(v.y, 0), 255) << 8) | ((uint32_t)min(max(v.z, 0), 255) << 16) |
                                    ((uint32_t)min(max(v.w, 0), 255) << 24);
    }
}

// Summary trajectories as 32-bit integers for the wire (compartment totals are whole numbers below 2^31).
extern "C" __global__ void __launch_bounds__(256) vgx_traj_i32_kernel(const double *src, int32_t *dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (int32_t)src[i];
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_traj_i32(const double *src, int32_t *dst, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(vgx_traj_i32_kernel, dim3(8192), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

// ---- host-side launchers ----
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_lists_settle(const VgxDirectArgs *a, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_quad(const VgxDirectArgs *a, const double *cd,
                                                                            double *effMig, double *maxEBM, int32_t *has_mig,
                                                                            int long_lists, hipStream_t stream) {
    hipLaunchKernelGGL(vgx_quad_prep_kernel, dim3(1), dim3(64), 0, stream, a->p, cd, effMig, maxEBM, has_mig);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    QArgs qa;
    qa.effMig = effMig; qa.maxEBM = maxEBM; qa.has_mig = has_mig;
    const unsigned grid = (unsigned)((a->n_replicates + 3) / 4);
    if (long_lists) {
        const int64_t n = (int64_t)a->n_replicates * a->p.P * a->r.cap;      // (list capacities are multiples of 4 from one site on)
        hipLaunchKernelGGL(vgx_quad_counts8_kernel, dim3(4096), dim3(256), 0, stream, a->r.lcnt32, (uint8_t *)(a->r.lcnt32 + n), (n + 3) / 4);
        hipLaunchKernelGGL(vgx_quad_long_kernel, dim3(grid), dim3(64), Q_LDS_BYTES, stream, *a, qa);
        // its lists hold zero-count entries and 4-byte counts only: squeeze, tile sums, 8-byte counts (vgx_quadf.hip)
        hipError_t e2 = vgxi_launch_lists_settle(a, stream);
        if (e2 != hipSuccess) return e2;
    }
    else hipLaunchKernelGGL(vgx_quad_kernel, dim3(grid), dim3(64), Q_LDS_BYTES, stream, *a, qa);
    return hipGetLastError();
}
