// vgx_lone.hip — direct Gillespie for ONE trajectory of a LARGE haplotype space: the latency kernel on occupancy lists.
//
// What `Simulator.simulate()` is upstream: one sequential event loop (src/_BirthDeath.pyx:396-429).  vgx_solo.hip runs that loop for
// small models with the reference's dense arrays in registers; here the haplotype space is large (BASELINE config 3: 4^8 haplotypes x 64
// populations) and the state is the ORDERED OCCUPANCY LISTS of vgx_dev.h — but resident in LDS for the whole call, one wavefront per
// replicate, so that no memory round trip sits between two events:
//   * a heap of list slots in LDS (rows of 16 slots; every population owns a contiguous run of rows, laid out again when one fills
//     up): per slot the haplotype, its infectious count (4 bytes) and THE SERIAL PREFIX SUM OF hapPopRate as the population's last
//     infect-update left it (pyx:519-528) — fastChoose(hapPopRate[pi], ...) (fast_choose.pxi:22-28) would form exactly those sums
//     again, so the haplotype choice is a compare + ballot over them (first over the sums at the ends of the 64-entry tiles, then
//     inside one tile) and the list is walked ONCE per event, by the refresh, not 1.5 times;
//   * lane p <-> population p: popRate and its serial prefix sums (the partial sums of the totalRate loop, pyx:537-539), migPopRate and
//     ITS prefix sums (fastChoose(migPopRate), pyx:676, is a ballot too), totals, BirthRate of the class, the list's place in the heap;
//   * every sequential f64 sum of the reference is one v_fmac_f64 (DPP row_newbcast) per term (vgx_flat.h), in the reference's order;
//   * migrationRates in LDS; uniforms 64 per refill by lane-parallel PCG64 jump-ahead; event records staged in LDS and written 64 at a
//     time (2 KB bursts); the logarithm of SampleTime only in calls that need the device clock (vgx_solo.hip, vgx_api.hip host_clock).
// Two forms (template GEN).  ONE CLASS (the one-class row kernel's scope, vgx_quad.hip: one susceptibility group, one rate class, no
// population that can switch its lockdown state): immunePopRate is +0.0, popRate = infectPopRate exactly, effectiveMigration /
// maxEffectiveBirthMigration come from vgx_quad_prep_kernel.  GENERAL (the general row kernel's scope, vgx_quadg.hip, at <= 64
// populations): several rate classes — a list entry's class rides in the top six bits of its haplotype word, tEventHapPopRate per class
// in the class lanes, BirthRate as the host's program of chain segments (vgx_quadg.h) —, several susceptibility groups (counts, their
// copy as of the population's last infect-update, immuneSourcePopRate: [P][S] tables in LDS; ImmunityTransition pyx:550-564), lockdown
// switches (CheckLockdown pyx:698-710 -> UpdateAllRates incl. effectiveMigration, pyx:279-351).  Both: popNum <= 64, no recombination,
// population sizes < 2^31, exact mode.
// When the lists outgrow the heap the replicate ends with capacity | VGX_LONE_FULL_SITE << 8 and the host runs the call again on the
// row kernel (the call is a function of state and seeds).  Start and end state are exchanged in the other kernels' layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_flat.h"
#include "vgx_lone.h"

#ifdef VGX_LONE_MARKS
#define MARK(name) asm volatile("; MARK " name)
#else
#define MARK(name)
#endif
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

namespace {

enum { ERR_ZERO_WEIGHT = 3, ERR_CAPACITY = 4, ERR_LOOP_GUARD = 5 };
enum { EV_BIRTH = 0, EV_DEATH, EV_SAMPLING, EV_MUTATION, EV_SUSCCHANGE, EV_MIGRATION };
// slots of the cold block (LDS): bookkeeping of the call that the event loop itself never reads
enum { C_EV_PTR = 0, C_LOOPS, C_ATT_LOOPS, C_TRAJ_NEXT, C_RESTARTS, C_ATT, C_GOOD, C_LAST_ATT, C_LOC_N, C_FA_N, C_ATT_EV0, C_ATT_LOC0 };
enum { CNT_MIGN = 6, CNT_SWAP = 7 };
#ifndef VGX_LONE_SEG
#define VGX_LONE_SEG (1 << 30)   // iterations per segment of the event loop (32-bit countdowns; a test build shortens it)
#endif
#define LONE_BIG VGX_LONE_SEG
// flags in the high word of zero_w (the low word: a fastChoose stopped on a zero weight)
#define LONE_F_CAP (2ull << 32)      // event log / list capacity in HBM
#define LONE_F_FULL (4ull << 32)     // the LDS heap is full
#define HAP_MASK ((1 << VGX_LONE_HAP_BITS) - 1)

struct VgxLoneKArgs { VgxDirectArgs a; VgxLoneArgs la; };
typedef const VgxLoneKArgs __attribute__((address_space(4))) *LoneKA;
static __device__ __forceinline__ LoneKA cold_args(LoneKA k) {
    asm volatile("" : "+s"(k));
    return k;
}
static __device__ __forceinline__ int lane_get(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
static __device__ __forceinline__ int wave_sum_i32(int v, int lane) { return (int)bcast_i64(iscan((int64_t)v, lane), 63); }

template <bool CLOCK, bool RCPDIV, bool GEN>
struct Lone {
    int P, H, sites, lane;
    int prow;                     // DPP rows of 16 lanes that hold populations
    int nrows;                    // rows of the heap
    Masks M;
    // LDS
    double *ldCum, *ldTend, *ldMig, *ldRng;
    int32_t *ldHap, *ldCnt;
    uint32_t *ldStage;
    uint64_t *ldRngK, *ldRngS;
    int64_t *ldCold;
    // the single rate class
    double c_b, c_sig, c_d, c_tm;
    // ---- population lanes ----
    double popRate, cumPop, migR, cumMig, maxEBM, totS, totI, cd, asz, rcp, smul, bC;
    int row0, rows, nocc;         // the population's list: first heap row, rows owned, entries
    // ---- general form ----
    int S, C, CB, nseg, no_imm, ld_any;
    double *ldS, *ldSst, *ldImm, *ldBC, *ldSig, *ldTrans;   // [P][S] x 3, [P][CB], [CB][S], [S][S]
    double *gEff;                 // effectiveMigration of this replicate [P][P] (global memory: read by accepted / rejected migrations only)
    double infP, immP, thrCur, sgnLD, mult;        // population lanes: infectPopRate, immunePopRate, the threshold whose crossing switches the
                                                   // lockdown state and the side it lies on, samplingMultiplier
    double cl_d, cl_s, cl_tm;     // class lanes (lane c < C): recovery, sampling, total mutation rate
    int cl_bidx, cl_stype;        // ... its birth class and suscType
    double cbb;                   // birth-class lanes (lane cb < CB): transmission rate
    int cb_segl;                  // ... the segment its BirthRate sum ends with (-1: every susceptibility of the class is zero)
    double sg_sig;                // segment lanes (lane sg < nseg): the segment's susceptibility,
    int sg_par, sg_sn;            // its parent segment (-1: starts at 0.0) and group
    double cumul_l;               // group lanes (lane s < S): suscepCumulTransition
    // ---- mutation lanes (sites <= 16): mRate[h, lane], hapMutType[h, lane / 3, lane % 3] of haplotype 0 when all haplotypes share them ----
    double mut_m, mut_h;
    int mut_uni;
    // ---- counters: lane t < 6 <-> events of type t, lane 6 rejected migrations ----
    uint64_t cnt;
    // ---- wave-uniform ----
    double totalRate, totalMig, Rtot, currentTime, gI;
    int has_mig;
    unsigned long long zero_w;
    int stage_n;
    int pos;                      // iterations consumed from the current batch of 64 uniforms (32 = empty)
    double u_pre, n_pre;
    double tlimit, next_tg;
    int ev_left, loop_left, s_left, ev_left0, loop_left0;
    uint32_t iter_base;
#ifdef VGX_PROFILE
    unsigned long long prof_t0, prof_acc;
#endif

    __device__ __forceinline__ void prof(int i) { (void)i; PROF(i); }
    __device__ __forceinline__ int64_t cold_get(int i) const { return ldCold[i]; }
    __device__ __forceinline__ void cold_set(int i, int64_t v) {
        if (lane == 0) ldCold[i] = v;
        WSYNC();
    }
    __device__ __forceinline__ void prefetch_uniforms() {
        const int q = min(pos, 31);
        u_pre = ldRng[2 * q + 1];
        if (CLOCK) n_pre = ldRng[2 * q];
    }
    __device__ __forceinline__ void zero_weight(bool c) { zero_w |= (__builtin_amdgcn_ballot_w64(c) ? 1ull : 0ull); }
    static __device__ __forceinline__ int first_or(unsigned long long hit, int other) {
        return uni_i32(hit ? (int)__builtin_ctzll(hit) : other);
    }
    static __device__ __forceinline__ int hap_of(int w) { return GEN ? (w & HAP_MASK) : w; }
    static __device__ __forceinline__ int cls_of(int w) { return GEN ? (int)((uint32_t)w >> VGX_LONE_HAP_BITS) : 0; }

    // ---- the heap ---------------------------------------------------------------------------------------------------------
    // A layout for the current lists (+ one entry for `need_pi`): every population its minimum of rows, the spare rows in proportion.
    __device__ __forceinline__ bool plan(int need_pi, int &nr0, int &nrw) const {
        const int nneed = lane < P ? nocc + (lane == need_pi ? 1 : 0) : 0;
        const int mn = lane < P ? vgx_lone_min_rows(nneed) : 0;
        const int tot = wave_sum_i32(mn, lane);
        if (tot > nrows) return false;
        const int spare = nrows - tot;
        const int extra = lane < P ? (int)(((int64_t)spare * mn) / tot) : 0;
        nrw = mn + extra;
        nr0 = (int)iscan((int64_t)nrw, lane) - nrw;
        return true;
    }
    // n list slots from slot `src` to slot `dst` (hap, cnt, cum); the ranges may overlap
    __device__ __forceinline__ void move_slots(int src, int dst, int n) {
        if (src == dst || n <= 0) return;
        const int nt = (n + 63) >> 6;
        for (int i = 0; i < nt; ++i) {
            const int t = dst < src ? i : nt - 1 - i;     // towards lower slots: first tile first; towards higher: last tile first
            const int idx = 64 * t + lane;
            int h = 0, c = 0;
            double q = 0.0;
            if (idx < n) { h = ldHap[src + idx]; c = ldCnt[src + idx]; q = ldCum[src + idx]; }
            WSYNC();
            if (idx < n) { ldHap[dst + idx] = h; ldCnt[dst + idx] = c; ldCum[dst + idx] = q; }
            WSYNC();
        }
    }
    __device__ __forceinline__ void tile_ends(int p) {
        const int n = uni_i32(lane_get(nocc, p)), r0 = uni_i32(lane_get(row0, p));
        const int nt = (n + 63) >> 6;
        if (lane < nt) ldTend[r0 + 4 * lane] = ldCum[16 * r0 + min(64 * lane + 63, n - 1)];
    }
    // room for one more entry in population pi's list: the lists move to a new layout; false: the heap is full
    __device__ __forceinline__ bool relayout(int need_pi) {
        int nr0, nrw;
        if (!plan(need_pi, nr0, nrw)) return false;
        const int mn = lane < P ? vgx_lone_min_rows(nocc) : 0;
        const int cp = (int)iscan((int64_t)mn, lane) - mn;       // every list at its lowest place first ...
        for (int p = 0; p < P; ++p)
            move_slots(16 * uni_i32(lane_get(row0, p)), 16 * uni_i32(lane_get(cp, p)), uni_i32(lane_get(nocc, p)));
        for (int p = P - 1; p >= 0; --p)                          // ... then up to its new one, the last population first
            move_slots(16 * uni_i32(lane_get(cp, p)), 16 * uni_i32(lane_get(nr0, p)), uni_i32(lane_get(nocc, p)));
        row0 = nr0; rows = nrw;
        for (int p = 0; p < P; ++p) tile_ends(p);
        WSYNC();
        return true;
    }
    // the lists of a state in the other kernels' layout (occupancy lists in HBM) into the heap; false: they do not fit
    __device__ __forceinline__ bool load_lists(const int32_t *gn, const int32_t *gh, const int32_t *gcl, const int64_t *gc, int64_t gcap) {
        nocc = lane < P ? gn[lane] : 0;
        int nr0, nrw;
        if (!plan(-1, nr0, nrw)) return false;
        row0 = nr0; rows = nrw;
        double ti = 0.0;
        for (int p = 0; p < P; ++p) {
            const int n = uni_i32(lane_get(nocc, p)), base = 16 * uni_i32(lane_get(row0, p));
            int64_t s = 0;
            for (int k = lane; k < n; k += 64) {
                const int64_t c = gc[(int64_t)p * gcap + k];
                int w = gh[(int64_t)p * gcap + k];
                if (GEN) w |= gcl[(int64_t)p * gcap + k] << VGX_LONE_HAP_BITS;
                ldHap[base + k] = w;
                ldCnt[base + k] = (int)c;
                ldCum[base + k] = 0.0;
                s += c;
            }
            const double tp = (double)bcast_i64(iscan(s, lane), 63);
            ti = lane == p ? tp : ti;
        }
        totI = ti;
        WSYNC();
        return true;
    }

    // first index whose haplotype is >= h in population's list (n entries from slot base); found: that entry holds h
    __device__ __forceinline__ int lower_bound(int h, int n, int base, bool &found) const {
        for (int t = 0; 64 * t < n; ++t) {
            const int idx = 64 * t + lane;
            const int hv = idx < n ? hap_of(ldHap[base + idx]) : 0x7fffffff;
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(hv >= h);
            if (hit) {
                const int l = (int)__builtin_ctzll(hit);
                found = lane_get(hv, l) == h;
                return uni_i32(64 * t + l);
            }
        }
        found = false;
        return n;
    }
    // entry k leaves the list: (k, n) one slot down, first tile first
    __device__ __forceinline__ void list_remove(int pi, int k, int n, int base) {
        for (int t = k >> 6; 64 * t < n - 1; ++t) {
            const int idx = 64 * t + lane;
            const bool mv = idx >= k && idx < n - 1;
            int h = 0, c = 0;
            if (mv) { h = ldHap[base + idx + 1]; c = ldCnt[base + idx + 1]; }
            WSYNC();
            if (mv) { ldHap[base + idx] = h; ldCnt[base + idx] = c; }
            WSYNC();
        }
        nocc = lane == pi ? n - 1 : nocc;
    }
    // a new entry (hap, 1) at index k: [k, n) one slot up, last tile first.  The caller made room.
    __device__ __forceinline__ void list_insert(int pi, int k, int word, int n, int base) {
        if (n > k)
            for (int t = (n - 1) >> 6; t >= (k >> 6); --t) {
                const int idx = 64 * t + lane;
                const bool mv = idx >= k && idx < n;
                int h = 0, c = 0;
                if (mv) { h = ldHap[base + idx]; c = ldCnt[base + idx]; }
                WSYNC();
                if (mv) { ldHap[base + idx + 1] = h; ldCnt[base + idx + 1] = c; }
                WSYNC();
            }
        if (lane == 0) { ldHap[base + k] = word; ldCnt[base + k] = 1; }
        WSYNC();
        nocc = lane == pi ? n + 1 : nocc;
    }
    // infectious[pi, hap] += 1 (NewInfections of a mutant or a migrant, pyx:246-251 / 662)
    __device__ __forceinline__ void list_add(int pi, int hap, int cls) {
        int n = uni_i32(lane_get(nocc, pi)), base = 16 * uni_i32(lane_get(row0, pi));
        bool found;
        const int k = lower_bound(hap, n, base, found);
        if (found) {
            if (lane == 0) ldCnt[base + k] += 1;
            WSYNC();
            return;
        }
        if (n + 1 > 16 * uni_i32(lane_get(rows, pi))) {
            if (!relayout(pi)) { zero_w |= LONE_F_FULL; return; }
            base = 16 * uni_i32(lane_get(row0, pi));
        }
        list_insert(pi, k, GEN ? (hap | (cls << VGX_LONE_HAP_BITS)) : hap, n, base);
    }
    // infectious[pi, entry k] -= 1; an entry that reaches 0 leaves the list
    __device__ __forceinline__ void list_dec(int pi, int k, int cnt_k, int n, int base) {
        if (cnt_k == 1) { list_remove(pi, k, n, base); return; }
        if (lane == 0) ldCnt[base + k] = cnt_k - 1;
        WSYNC();
    }

    // ---- UpdateRates, infect branch (pyx:518-528) ----
    // BirthRate's terms of population pi for a susceptible weight x (pyx:389-390): ((x * m) * m * cd) / as, one source population per lane
    __device__ __forceinline__ double birth_terms(int pi, double x) const {
        const double m = ldMig[pi * P + min(lane, P - 1)];
        const double t = x * m * m * cd;
        const double T = RCPDIV ? div_by_const(t, asz, rcp) : t / asz;
        return lane < P ? T : 0.0;
    }
    // one-class form: BirthRate of the class in population pi (pyx:382-392): the sum over the source populations, in order
    __device__ __forceinline__ double birth_rate(int pi) {
        const double T = birth_terms(pi, bcast(totS, pi) * c_sig);
        const double acc = flat_rows<false>(T, prow, 0.0, M);
        return c_b * bcast(acc, P - 1);
    }
    // general form: BirthRate of every birth class in population pi by the program of chain segments — per class the reference's
    // (sn, pn) order with the zero-susceptibility groups left out (+0.0 upstream), common prefixes of classes shared (vgx_quadg.h) —
    // into ldBC[pi, :]; and the copy of the susceptible counts the sums were built from (susceptHapPopRate = S * sigma, pyx:385-386)
    __device__ __forceinline__ void birth_rates_gen(int pi) {
        double segv = 0.0;                         // lane sg: the sum at the end of segment sg
        for (int sg = 0; sg < nseg; ++sg) {
            const int par = uni_i32(lane_get(sg_par, sg)), sn = uni_i32(lane_get(sg_sn, sg));
            const double x = ldS[pi * S + sn] * bcast(sg_sig, sg);
            const double T = birth_terms(pi, x);
            const double carry = par >= 0 ? bcast(segv, par) : 0.0;
            const double acc = flat_rows<false>(T, prow, carry, M);
            const double tot = bcast(acc, P - 1);
            segv = lane == sg ? tot : segv;
        }
        const double ps = bperm_f64(segv, max(cb_segl, 0));
        if (lane < CB) ldBC[pi * CB + lane] = cbb * (cb_segl >= 0 ? ps : 0.0);
        if (lane < S) ldSst[pi * S + lane] = ldS[pi * S + lane];
        WSYNC();
    }
    // BirthRate of population pi and the migration rates after a birth / death there: the two sums do not depend on each other
    __device__ __forceinline__ double birth_rate_and_remig(int pi) {
        if (!(has_mig && prow == 4)) {
            const double b = birth_rate(pi);
            remig();
            return b;
        }
        const double T = birth_terms(pi, bcast(totS, pi) * c_sig);
        migR = lane < P ? maxEBM * totS * (gI - totI) : 0.0;
        double acc;
        flat_two64(T, migR, M, acc, cumMig);
        totalMig = bcast(cumMig, P - 1);
        return c_b * bcast(acc, P - 1);
    }
    __device__ __forceinline__ double tE_of(double b, double sm) const { return ((b + c_d) + sm) + c_tm; }   // pyx:522-526
    // general form: the event rates of every class in population pi, one class per lane (pyx:309-317 / 522-526)
    __device__ __forceinline__ void class_rates(int pi, double &e0, double &e1, double &e2, double &tE) const {
        e0 = ldBC[pi * CB + min(cl_bidx, CB - 1)];
        e1 = e0 + cl_d;
        e2 = e1 + cl_s * bcast(mult, pi);
        tE = e2 + cl_tm;
    }
    // hapPopRate = tE * infectious over the list in haplotype order, its serial prefix sums into the heap; returns infectPopRate[pi]
    // (one-class form: tE wave-uniform; general form: tEv = tEventHapPopRate per class lane)
    __device__ __forceinline__ double refresh(int pi, double tE, double tEv = 0.0) {
        const int n = uni_i32(lane_get(nocc, pi)), r0 = uni_i32(lane_get(row0, pi)), base = 16 * r0;
        double carry = 0.0;
        const int nfull = n >> 6;                  // tiles that lie inside the list: no bounds to look at, four rows each
        int c = lane < n ? ldCnt[base + lane] : 0;
        int hw = (GEN && lane < n) ? ldHap[base + lane] : 0;
        for (int t = 0; t < nfull; ++t) {
            const int at = base + 64 * t + lane;
            const double te = GEN ? bperm_f64(tEv, cls_of(hw)) : tE;
            const double w = te * (double)c;
            const int nx = 64 * t + 64 + lane;     // the next tile's counts are on their way during this tile's chain
            c = nx < n ? ldCnt[base + nx] : 0;
            if (GEN) hw = nx < n ? ldHap[base + nx] : 0;
            double acc = carry;
            const double mu = 1.0;
            {
                const double v = w;
                constexpr bool SCAN = true;
                FLAT_ROW("0x1", "s_nop 1\n\t");
                acc = row_carry(acc, 1); FLAT_ROW("0x2", "");
                acc = row_carry(acc, 2); FLAT_ROW("0x4", "");
                acc = row_carry(acc, 3); FLAT_ROW("0x8", "");
                (void)mu;
            }
            ldCum[at] = acc;
            carry = bcast(acc, 63);
            if (lane == 0) ldTend[r0 + 4 * t] = carry;
        }
        if (n & 63) {                              // the list's last, partial tile
            const int t = nfull, idx = 64 * t + lane;
            const double te = GEN ? bperm_f64(tEv, cls_of(hw)) : tE;
            const double w = idx < n ? te * (double)c : 0.0;
            const int m = n - 64 * t;
            const double cum = flat_rows<true>(w, (m + 15) >> 4, carry, M);
            if (idx < n) ldCum[base + idx] = cum;
            carry = bcast(cum, m - 1);
            if (lane == 0) ldTend[r0 + 4 * t] = carry;
        }
        WSYNC();
        return carry;
    }
    // immunePopRate[pi] = 0 + immuneSourcePopRate[pi, 0] + ... (pyx:530-533)
    __device__ __forceinline__ double immune_sum(int pi) {
        if (!GEN || no_imm) return 0.0;
        const double v = lane < S ? ldImm[pi * S + lane] : 0.0;
        const double acc = flat_rows<false>(v, 1, 0.0, M);
        return bcast(acc, S - 1);
    }
    // popRate changed: its serial prefix sums and totalRate (pyx:536-539)
    __device__ __forceinline__ void rescan_pop() {
        cumPop = flat_rows<true>(popRate, prow, 0.0, M);
        totalRate = bcast(cumPop, P - 1);
    }
    // migPopRate of every population, its prefix sums and totalMigrationRate (pyx:541-546)
    __device__ __forceinline__ void remig() {
        if (!has_mig) { totalMig = 0.0; return; }
        migR = lane < P ? maxEBM * totS * (gI - totI) : 0.0;
        cumMig = flat_rows<true>(migR, prow, 0.0, M);
        totalMig = bcast(cumMig, P - 1);
    }
    // UpdateAllRates (pyx:279-351) from the compartments.  One-class form: effectiveMigration and its maxima are parameters;
    // general form: they follow the contact densities and are formed here (lane <-> pn2, the sums over pn3 serially in every lane)
    __device__ __forceinline__ void rebuild_all(LoneKA ka) {
        for (int pn = 0; pn < P; ++pn) {
            if (GEN) {
                if (lane < S) ldImm[pn * S + lane] = cumul_l * ldS[pn * S + lane];       // pyx:319-321
                WSYNC();
                birth_rates_gen(pn);
                double e0, e1, e2, tEv;
                class_rates(pn, e0, e1, e2, tEv);
                const double inP = refresh(pn, 0.0, tEv);
                const double imP = immune_sum(pn);
                infP = lane == pn ? inP : infP;
                immP = lane == pn ? imP : immP;
                popRate = lane == pn ? inP + imP : popRate;
            } else {
                const double b = birth_rate(pn);
                bC = lane == pn ? b : bC;
                const double inP = refresh(pn, tE_of(b, bcast(smul, pn)));
                popRate = lane == pn ? inP : popRate;
            }
        }
        rescan_pop();
        if (GEN) {
            double mx = 0.0;
            for (int p1 = 0; p1 < P; ++p1) {
                double e = 0.0;
                for (int p3 = 0; p3 < P; ++p3) {
                    const double m13 = ldMig[p1 * P + p3], c3 = bcast(cd, p3), a3 = bcast(asz, p3);
                    const double m23 = ldMig[min(lane, P - 1) * P + p3];
                    e += m13 * m23 * c3 / a3;
                }
                if (lane < P && lane != p1) {
                    gEff[p1 * P + lane] = e;
                    if (e > mx) mx = e;
                }
            }
            maxEBM = lane < P ? mx * cold_args(ka)->a.p.maxEffectiveBirth : 0.0;
            has_mig = any_lane(maxEBM > 0.0) ? 1 : 0;
            migR = 0.0; cumMig = 0.0;
        }
        remig();
        Rtot = totalRate + totalMig;
        WSYNC();
    }
    // CheckLockdown (pyx:698-710) for populations [lo, hi): applies and logs the switches; returns whether any happened
    __device__ __forceinline__ bool check_lockdowns(LoneKA ka_, int64_t rep, int lo, int hi) {
        const auto *a = &cold_args(ka_)->a;
        const auto &p = a->p;
        const auto &r = a->r;
        bool any = false;
        int64_t loc_n = cold_get(C_LOC_N);
        const int64_t iter_key = (cold_get(C_ATT) << 40) | (cold_get(C_ATT_LOOPS) + (int64_t)(loop_left0 - loop_left));
        for (int pi = lo; pi < hi; ++pi)
            for (int pass = 0; pass < 2; ++pass) {
                const double ti = bcast(totI, pi), sg = bcast(sgnLD, pi);
                const double sz = (double)p.sizes[pi];
                const bool flip = pass == 0 ? (ti > p.startLD[pi] * sz && sg > 0.0) : (ti < p.endLD[pi] * sz && sg < 0.0);
                if (!any_lane(flip)) continue;
                const double ncd = pass == 0 ? p.cdAfter[pi] : p.cdBefore[pi];
                cd = lane == pi ? ncd : cd;
                sgnLD = lane == pi ? (pass == 0 ? -1.0 : 1.0) : sgnLD;
                thrCur = lane == pi ? (pass == 0 ? p.endLD[pi] * sz : p.startLD[pi] * sz) : thrCur;
                if (lane == 0 && loc_n < r.loc_cap) {
                    r.loc_rec[(rep * r.loc_cap + loc_n) * 2 + 0] = pass == 0 ? 1 : 0;
                    r.loc_rec[(rep * r.loc_cap + loc_n) * 2 + 1] = pi;
                    r.loc_time[rep * r.loc_cap + loc_n] = currentTime;
                    r.loc_iter[rep * r.loc_cap + loc_n] = iter_key;
                }
                if (loc_n >= r.loc_cap) zero_w |= LONE_F_CAP;
                cnt += (lane == CNT_SWAP) ? 1u : 0u;
                loc_n += 1;
                any = true;
            }
        if (any) cold_set(C_LOC_N, loc_n);
        return any;
    }
    // after an event in population pi: has its infectious total crossed the threshold that switches its lockdown state?
    __device__ __forceinline__ bool crossed(int pi) const {
        return GEN && ld_any && any_lane(lane == pi && (totI - thrCur) * sgnLD > 0.0);
    }
    // general form: Birth's fastChoose(susceptHapPopRate[pi, hi, :], their sum, rn7) (pyx:569-572): the group that loses a host.
    // susceptHapPopRate[pi, hi, s] = (S[pi, s] as of the population's last infect-update) * sigma[birth class of hi, s]
    __device__ __forceinline__ int choose_group(int pi, int bidx, double rn7) {
        const double x = lane < S ? ldSst[pi * S + lane] * ldSig[bidx * S + lane] : 0.0;
        int sidx = 0;
        if (S > 1) {
            const double cx = flat_rows<true>(x, 1, 0.0, M);
            const double r8 = bcast(cx, S - 1) * rn7;
            sidx = first_or(__builtin_amdgcn_ballot_w64(lane < S && !(cx < r8)), S - 1);
        }
        zero_weight(lane == sidx && x == 0.0);
        return sidx;
    }
    // general form: one host of population pi moves between the infected and susceptible group sidx (NewInfections sgn = +1 /
    // NewRecoveries sgn = -1, pyx:246-260); with_imm: the group's immuneSourcePopRate follows (pyx:559-560, 602, 618-620)
    __device__ __forceinline__ void group_move(int pi, int sidx, double sgn, bool with_imm) {
        const double cm = bcast(cumul_l, sidx);
        if (lane == 0) {
            const double sv = ldS[pi * S + sidx] - sgn;
            ldS[pi * S + sidx] = sv;
            if (with_imm) ldImm[pi * S + sidx] = cm * sv;
        }
        WSYNC();
    }

    // ---- event log ----
    __device__ __forceinline__ void stage_flush(LoneKA ka_, int64_t rep) {
        if (stage_n > 0) {
            const auto *a = &cold_args(ka_)->a;
            WSYNC();
            const int64_t ev_now = cold_get(C_EV_PTR) + (int64_t)(ev_left0 - ev_left);
            const int64_t slot0 = ev_now - stage_n - a->r.ev_base;
            if (slot0 < 0 || slot0 + stage_n > a->r.evcap) {
                zero_w |= LONE_F_CAP;
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
    // AddEvent (events.pxi:37-44) into the LDS stage + the counters; type < 0: a rejected migration (counter only)
    __device__ __forceinline__ void log_event(int type, int hap, int pop, int nh, int np, double den) {
        const int ctr = type >= 0 ? type : CNT_MIGN;
        cnt += (lane == ctr) ? 1u : 0u;
        if (type >= 0) {
            int v = __double2hiint(den);
            const int it = uni_i32((int)(iter_base - (uint32_t)loop_left));
            type = uni_i32(type); hap = uni_i32(hap); pop = uni_i32(pop); nh = uni_i32(nh); np = uni_i32(np);
            SOLO_WRITELANE(v, type, 0); SOLO_WRITELANE(v, hap, 1); SOLO_WRITELANE(v, pop, 2); SOLO_WRITELANE(v, nh, 3);
            SOLO_WRITELANE(v, np, 4); SOLO_WRITELANE(v, it, 5);
            v = lane == 6 ? __double2loint(den) : v;
            if (lane < 8) ldStage[stage_n * 8 + lane] = (uint32_t)v;
            stage_n += 1;
            ev_left -= 1;
            s_left -= (type == EV_SAMPLING) ? 1 : 0;
        }
    }
    __device__ __forceinline__ void traj_emit(LoneKA ka_, int64_t rep, double t_new, bool final_fill) {
        const auto &r = cold_args(ka_)->a.r;
        int64_t traj_next = cold_get(C_TRAJ_NEXT);
        const int64_t n0 = traj_next;
        while (traj_next < r.traj_points) {
            const double tg = r.traj_t0 + (double)traj_next * r.traj_dt;
            if (!final_fill && !(tg < t_new)) break;
            double *o = r.traj + (rep * r.traj_points + traj_next) * (int64_t)P * 2;
            if (lane < P) { o[lane * 2 + 0] = totI; o[lane * 2 + 1] = totS; }
            traj_next += 1;
        }
        if (traj_next != n0) cold_set(C_TRAJ_NEXT, traj_next);
        next_tg = traj_next < r.traj_points ? r.traj_t0 + (double)traj_next * r.traj_dt : __builtin_inf();
    }

    // ---- the event loop proper: iterations whose event is a Birth, a Death or a Sampling that leaves its haplotype in the list
    // (pyx:568-635: 95 % of the events of BASELINE config 3) run here back to back as ONE straight path; it returns when something else
    // has to happen.  For FAST_SLOW nothing of the iteration is consumed yet: the general form below repeats its choices. ----
    enum { FAST_END = 0, FAST_REFILL, FAST_SLOW, FAST_POST };
    __device__ __forceinline__ int fast_loop(int &pi_post) {
        for (;;) {
            PROF(9);
            MARK("loop_top");
            if (min(min(ev_left, loop_left), s_left) <= 0) return FAST_END;
            if (CLOCK && !any_lane(currentTime < tlimit)) return FAST_END;
            if (__builtin_expect(pos == 32, 0)) return FAST_REFILL;
            const double u = u_pre;
            double t_new = 0.0;
            if (CLOCK) {
                t_new = currentTime + (n_pre / Rtot);                      // SampleTime pyx:476-478
                if (__builtin_expect(any_lane(next_tg < t_new), 0)) return FAST_SLOW;
            }
            PROF(0);
            MARK("uniforms_done");
            // GenerateEvent (pyx:483-512)
            double choose = u * Rtot;
            if (__builtin_expect(!any_lane(totalRate > choose), 0)) return FAST_SLOW;
            double rn = fdiv(choose, totalRate);
            const double r2 = totalRate * rn;
            const int pi = first_or(__builtin_amdgcn_ballot_w64(lane < P && !(cumPop < r2)), P - 1);   // fastChoose(popRate), fc:18-31
            const double W = bcast(popRate, pi), Cm = bcast(cumPop, pi);
            const double yW = refined_rcp(W);
            rn = fdiv_y(r2 - (Cm - W), W, yW);
            choose = rn * W;
            double IN = W;
            if (GEN) {
                const double IM = bcast(immP, pi);
                IN = bcast(infP, pi);
                if (__builtin_expect(any_lane(IM > choose), 0)) return FAST_SLOW;       // ImmunityTransition
                rn = fdiv(choose - IM, IN);                                             // pyx:499-500
            } else {
                rn = fdiv_y(choose, W, yW);     // immunePopRate[pi] = +0.0: the infect branch, (choose - 0.0) / infectPopRate[pi] (pyx:499-500)
            }
            const double r4 = IN * rn;
            PROF(1);
            MARK("pop_chosen");
            // haplotype: fastChoose(hapPopRate[pi], infectPopRate[pi], rn) on the list's stored prefix sums
            const int n = uni_i32(lane_get(nocc, pi)), r0 = uni_i32(lane_get(row0, pi)), base = 16 * r0;
            const int nt = (n + 63) >> 6;
            int t = 0;
            if (nt > 1) {
                const double te = ldTend[r0 + 4 * min(lane, nt - 1)];
                t = first_or(__builtin_amdgcn_ballot_w64(lane < nt && !(te < r4)), nt - 1);
            }
            const int idx = 64 * t + lane;
            const bool in = idx < n;
            const int ic = base + max(min(idx, n - 1), 0);      // (unconditional loads on clamped indices: no EXEC juggling)
            const double cv = ldCum[ic];
            const int hv = ldHap[ic], nv = ldCnt[ic];
            double e0v = 0.0, e1v = 0.0, e2v = 0.0, tEv = 0.0;  // general form: the class lanes' rates in population pi
            if (GEN) class_rates(pi, e0v, e1v, e2v, tEv);
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(in && !(cv < r4));
            const int kl = first_or(hit, (n - 1) & 63);
            const double cum_k = bcast(cv, kl);
            const int cnt_k = uni_i32(lane_get(nv, kl)), hw_k = uni_i32(lane_get(hv, kl));
            const int hap_k = hap_of(hw_k), cls_k = cls_of(hw_k);
            const double smpi = GEN ? 0.0 : bcast(smul, pi);
            const double bpi = GEN ? bcast(e0v, cls_k) : bcast(bC, pi);
            const double e1 = GEN ? bcast(e1v, cls_k) : bpi + c_d;
            const double e2 = GEN ? bcast(e2v, cls_k) : e1 + smpi;
            const double tE = GEN ? bcast(tEv, cls_k) : e2 + c_tm;
            const double hpr_k = tE * (double)cnt_k;
            const double rn5 = fdiv(r4 - (cum_k - hpr_k), hpr_k);
            // event class: fastChoose(eventHapPopRate[pi, hi, 0..3], tEventHapPopRate[pi, hi], rn), pyx:503-511
            const double r6 = tE * rn5;
            const bool is_b = !any_lane(bpi < r6);
            const bool is_m = any_lane(e2 < r6);
            const bool is_s = any_lane(e1 < r6);       // (with !is_m: Sampling)
            // anything but the plain cases goes to the general form: a zero weight or the clamp at H-1 (fc:26-30), a mutation, the
            // last carrier's death (its entry leaves the list), no susceptible host left
            const double ts_pi = bcast(totS, pi);
            const double w_d = GEN ? bcast(cl_d, cls_k) : c_d;
            const double w_s = GEN ? bcast(cl_s, cls_k) * bcast(mult, pi) : smpi;
            const int w0 = is_b ? ((int)any_lane(bpi == 0.0) | (int)(!GEN && any_lane(ts_pi * c_sig == 0.0))) : is_s ? (int)any_lane(w_s == 0.0) : (int)any_lane(w_d == 0.0);
            const int odd = (int)any_lane(W == 0.0) | (int)(hit == 0ull) | (int)any_lane(hpr_k == 0.0) | (int)is_m | ((int)!is_b & (int)(cnt_k == 1)) | w0;
            if (__builtin_expect(odd != 0, 0)) return FAST_SLOW;
            int sidx = 0;
            if (GEN) {
                if (is_b) {
                    const unsigned long long z0 = zero_w;
                    sidx = choose_group(pi, uni_i32(lane_get(cl_bidx, cls_k)), S > 1 ? fdiv(r6, bpi) : 0.0);      // (r - (e0 - e0)) / e0
                    if (__builtin_expect(zero_w != z0, 0)) { zero_w = z0; return FAST_SLOW; }   // (the general form reports it)
                } else {
                    sidx = uni_i32(lane_get(cl_stype, cls_k));
                }
            }
            PROF(2);
            MARK("class_chosen");
            // ---- the iteration is this path's ----
            loop_left -= 1;
            pos += 1;
            prefetch_uniforms();
            if (CLOCK) currentTime = t_new;
            const int ei = is_b ? EV_BIRTH : is_s ? EV_SAMPLING : EV_DEATH;
            const double sgn = is_b ? 1.0 : -1.0;
            if (lane == 0) ldCnt[base + 64 * t + kl] = cnt_k + (is_b ? 1 : -1);
            if (GEN) group_move(pi, sidx, sgn, true);
            const double d1 = lane == pi ? sgn : 0.0;            // NewInfections / NewRecoveries (pyx:246-260)
            totS -= d1; totI += d1;
            gI += sgn;
            PROF(3);
            MARK("event_applied");
            // UpdateRates(pi, True, True, True), pyx:516-546
            if (GEN) {
                birth_rates_gen(pi);
                PROF(4);
                MARK("birthrate_done");
                remig();
                PROF(5);
                MARK("remig_done");
                class_rates(pi, e0v, e1v, e2v, tEv);
                const double inP = refresh(pi, 0.0, tEv);
                const double imP = immune_sum(pi);
                infP = lane == pi ? inP : infP;
                immP = lane == pi ? imP : immP;
                popRate = lane == pi ? inP + imP : popRate;
            } else {
                const double b = birth_rate_and_remig(pi);
                bC = lane == pi ? b : bC;
                PROF(4);
                MARK("birthrate_done");
                PROF(5);
                MARK("remig_done");
                const double inP = refresh(pi, tE_of(b, smpi));
                popRate = lane == pi ? inP : popRate;
            }
            PROF(6);
            MARK("refresh_done");
            rescan_pop();
            PROF(7);
            MARK("rescan_done");
            const double den = Rtot;
            Rtot = totalRate + totalMig;
            log_event(ei, hap_k, pi, sidx, is_b ? H : 0, den);
            PROF(8);
            MARK("logged");
            // what ends the run of fast iterations: a full stage, extinction (pyx:410-411), a lockdown threshold crossed (pyx:412)
            if (__builtin_expect(((int)any_lane(totalRate == 0.0) | (int)any_lane(gI == 0.0) | (int)(stage_n == 64) | (int)crossed(pi)) != 0, 0)) {
                pi_post = pi;
                return FAST_POST;
            }
        }
    }

    // ---- one iteration of the event loop in its general form (pyx:408-409): SampleTime, GenerateEvent with UpdateRates and AddEvent;
    // returns the population whose lockdown state has to be checked (pyx:412) ----
    __device__ __forceinline__ int iteration(LoneKA ka, int64_t rep) {
        const double u = u_pre;
        if (CLOCK) {
            const double t_new = currentTime + (n_pre / Rtot);             // SampleTime pyx:476-478
            if (any_lane(next_tg < t_new)) traj_emit(ka, rep, t_new, false);
            currentTime = t_new;
        }
        loop_left -= 1;
        pos += 1;
        prefetch_uniforms();
        const double den = Rtot;
        PROF(9);
        double choose = u * Rtot;                                          // GenerateEvent pyx:483-512
        int u_pi = -1;                 // population whose rates change: UpdateRates(u_pi, f_infect, f_immune, f_birth)
        bool f_infect = false, f_birth = false, f_immune = false;   // f_birth: its susceptible counts changed too (BirthRate and the migration rates follow)
        int ev_type = -1, ev_hap = 0, ev_pop = 0, ev_nh = 0, ev_np = 0;
        int ret_pi = 0;
        if (any_lane(totalRate > choose)) {
            double rn = fdiv(choose, totalRate);
            const double r2 = totalRate * rn;
            const int pi = first_or(__builtin_amdgcn_ballot_w64(lane < P && !(cumPop < r2)), P - 1);   // fastChoose(popRate), fc:18-31
            ret_pi = pi;
            const double W = bcast(popRate, pi), Cm = bcast(cumPop, pi);
            zero_weight(W == 0.0);
            rn = fdiv(r2 - (Cm - W), W);
            choose = rn * W;
            const double IM = GEN ? bcast(immP, pi) : 0.0, IN = GEN ? bcast(infP, pi) : W;
            ev_pop = pi;
            if (GEN && any_lane(IM > choose)) {
                // ---- ImmunityTransition (pyx:550-564) ----
                rn = choose / IM;
                int ssi, tsi;
                {
                    const double v = lane < S ? ldImm[pi * S + lane] : 0.0;
                    const double ci = flat_rows<true>(v, 1, 0.0, M);
                    const double r = IM * rn;
                    ssi = first_or(__builtin_amdgcn_ballot_w64(lane < S && !(ci < r)), S - 1);
                    const double w = bcast(v, ssi), tot = bcast(ci, ssi);
                    zero_weight(w == 0.0);
                    rn = (r - (tot - w)) / w;
                }
                {
                    const double tr = lane < S ? ldTrans[ssi * S + lane] : 0.0;
                    const double ct = flat_rows<true>(tr, 1, 0.0, M);
                    const double r = bcast(cumul_l, ssi) * rn;
                    tsi = first_or(__builtin_amdgcn_ballot_w64(lane < S && !(ct < r)), S - 1);
                    zero_weight(lane == tsi && tr == 0.0);
                }
                if (zero_w) return pi;
                group_move(pi, ssi, 1.0, true);
                group_move(pi, tsi, -1.0, true);
                u_pi = pi; f_immune = true;
                ev_type = EV_SUSCCHANGE; ev_hap = ssi; ev_nh = tsi; ev_np = 0;
            } else {
                // the infect branch: rn = (choose - immunePopRate[pi]) / infectPopRate[pi] (pyx:499-500)
                rn = GEN ? fdiv(choose - IM, IN) : fdiv(choose, W);
                const double r4 = IN * rn;
                // ---- haplotype: fastChoose(hapPopRate[pi], infectPopRate[pi], rn) on the list's stored prefix sums ----
                const int n = uni_i32(lane_get(nocc, pi)), r0 = uni_i32(lane_get(row0, pi)), base = 16 * r0;
                const int nt = (n + 63) >> 6;
                int t = 0;
                if (nt > 1) {
                    const double te = lane < nt ? ldTend[r0 + 4 * lane] : 0.0;
                    t = first_or(__builtin_amdgcn_ballot_w64(lane < nt && !(te < r4)), nt - 1);
                }
                const int idx = 64 * t + lane;
                const bool in = idx < n;
                const double cv = in ? ldCum[base + idx] : 0.0;
                const int hv = in ? ldHap[base + idx] : 0, nv = in ? ldCnt[base + idx] : 0;
                double e0v = 0.0, e1v = 0.0, e2v = 0.0, tEv = 0.0;
                if (GEN) class_rates(pi, e0v, e1v, e2v, tEv);
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(in && !(cv < r4));
                const int kl = first_or(hit, (n - 1) & 63);
                const int k = 64 * t + kl;
                const double cum_k = bcast(cv, kl);
                const int cnt_k = uni_i32(lane_get(nv, kl)), hw_k = uni_i32(lane_get(hv, kl));
                const int hap_k = hap_of(hw_k), cls_k = cls_of(hw_k);
                const double smpi = GEN ? bcast(cl_s, cls_k) * bcast(mult, pi) : bcast(smul, pi);
                const double dk = GEN ? bcast(cl_d, cls_k) : c_d, tmv = GEN ? bcast(cl_tm, cls_k) : c_tm;
                const double bpi = GEN ? bcast(e0v, cls_k) : bcast(bC, pi);
                const double e1 = bpi + dk, e2 = e1 + smpi, tE = e2 + tmv;
                // nothing reached r: the dense loop runs on to index H-1 (fc:26), a valid pick only if that haplotype is occupied
                if (!hit && !(n > 0 && hap_k == H - 1)) zero_w |= 1ull;
                const double hpr_k = tE * (double)cnt_k;
                zero_weight(hpr_k == 0.0);
                const double rn5 = fdiv(r4 - (cum_k - hpr_k), hpr_k);
                // ---- event class: fastChoose(eventHapPopRate[pi, hi, 0..3], tEventHapPopRate[pi, hi], rn), pyx:503-511 ----
                const double r6 = tE * rn5;
                const int ei = uni_i32((any_lane(bpi < r6) ? 1 : 0) + (any_lane(e1 < r6) ? 1 : 0) + (any_lane(e2 < r6) ? 1 : 0));
                const double w_ei = ei == 0 ? bpi : ei == 1 ? dk : ei == 2 ? smpi : tmv;
                zero_weight(w_ei == 0.0);
                if (zero_w) return pi;          // (indices may be meaningless: the call ends with the error)
                ev_hap = hap_k;
                if (ei < 3) {
                    // ---- Birth (pyx:568-605) / Death / Sampling (pyx:616-635) ----
                    const double sgn = ei == 0 ? 1.0 : -1.0;
                    int sidx = 0;
                    if (ei == 0) {
                        if (GEN) sidx = choose_group(pi, uni_i32(lane_get(cl_bidx, cls_k)), S > 1 ? r6 / bpi : 0.0);   // (r - (e0 - e0)) / e0
                        else zero_weight(bcast(totS, pi) * c_sig == 0.0);
                        if (zero_w) return pi;
                        if (lane == 0) ldCnt[base + k] = cnt_k + 1;
                        WSYNC();
                    } else {
                        if (GEN) sidx = uni_i32(lane_get(cl_stype, cls_k));
                        list_dec(pi, k, cnt_k, n, base);
                    }
                    if (GEN) group_move(pi, sidx, sgn, true);
                    const double d1 = lane == pi ? sgn : 0.0;
                    totS -= d1; totI += d1;
                    gI += sgn;
                    u_pi = pi; f_infect = true; f_birth = true; f_immune = true;
                    ev_type = ei; ev_nh = sidx; ev_np = ei == 0 ? H : 0;
                } else {
                    // ---- Mutation (pyx:640-667): site by mRate[h, :], derived state by hapMutType[h, site, :] ----
                    const auto &p = cold_args(ka)->a.p;
                    rn = (r6 - (tE - tmv)) / tmv;
                    int mi, DS;
                    if (sites <= 16) {
                        // one site per lane, the three derived states of site s in lanes 3 s .. 3 s + 2: ONE round trip to memory for both
                        // choices (none when every haplotype has the same rows: they then sit in registers since the start of the call)
                        double wm = mut_m, wh = mut_h;
                        if (!mut_uni) {
                            wm = lane < sites ? p.mRate[(int64_t)hap_k * sites + lane] : 0.0;
                            wh = lane < 3 * sites ? p.hapMutType[(int64_t)hap_k * sites * 3 + lane] : 0.0;
                        }
                        {   // fastChoose(mRate[hi, :], tmRate[hi], rn)
                            const double cm = flat_rows<true>(wm, 1, 0.0, M);
                            const double r = tmv * rn;
                            mi = first_or(__builtin_amdgcn_ballot_w64(lane < sites && !(cm < r)), sites - 1);
                            const double wi = bcast(wm, mi), total = bcast(cm, mi);
                            zero_weight(wi == 0.0);
                            rn = (r - (total - wi)) / wi;
                        }
                        {   // fastChoose(hapMutType[hi, mi, :], their sum, rn)
                            const double w0 = bcast(wh, 3 * mi), w1 = bcast(wh, 3 * mi + 1), w2 = bcast(wh, 3 * mi + 2);
                            const double r = (w0 + w1 + w2) * rn;
                            int i = 0;
                            double total = w0;
                            if (any_lane(total < r)) { i = 1; total += w1; if (any_lane(total < r)) i = 2; }
                            zero_weight((i == 0 ? w0 : i == 1 ? w1 : w2) == 0.0);
                            DS = uni_i32(i);
                        }
                    } else {
                        {   // fastChoose(mRate[hi, :], tmRate[hi], rn)
                            const double *w = p.mRate + (int64_t)hap_k * sites;
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
                            const double *w = p.hapMutType + ((int64_t)hap_k * sites + mi) * 3;
                            const double r = (w[0] + w[1] + w[2]) * rn;
                            int i = 0;
                            double total = w[0];
                            while (any_lane(total < r) && i < 2) { i += 1; total += w[i]; }
                            zero_weight(w[i] == 0.0);
                            DS = uni_i32(i);
                        }
                    }
                    if (zero_w) return pi;
                    const int digit4 = 1 << (2 * (sites - mi - 1));      // Mutate, pyx:2420-2427
                    const int AS = (hap_k / digit4) % 4;
                    if (DS >= AS) DS += 1;
                    const int nhi = uni_i32(hap_k + (DS - AS) * digit4);
                    const int ncls = GEN ? uni_i32(p.cls[nhi]) : 0;
                    list_dec(pi, k, cnt_k, n, base);
                    list_add(pi, nhi, ncls);
                    if (zero_w) return pi;
                    u_pi = pi; f_infect = true;
                    ev_type = EV_MUTATION; ev_nh = nhi; ev_np = 0;
                }
            }
        } else {
            // ---- GenerateMigration (pyx:672-694) ----
            double rn = (choose - totalRate) / totalMig;
            int tpi;
            {   // fastChoose(migPopRate, totalMigrationRate, rn) on the stored prefix sums
                const double r = totalMig * rn;
                tpi = first_or(__builtin_amdgcn_ballot_w64(lane < P && !(cumMig < r)), P - 1);
                const double w = bcast(migR, tpi), tot = bcast(cumMig, tpi);
                zero_weight(w == 0.0);
                rn = (r - (tot - w)) / w;
            }
            ret_pi = tpi;
            int spi;
            {   // fastChoose_skip(totalInfectious, globalInfectious - totalInfectious[tpi], rn, tpi), fc:36-52: integer weights
                const int64_t wv = (lane == tpi || lane >= P) ? 0 : (int64_t)totI;
                const int64_t pre = iscan(wv, lane);
                const double r = (gI - bcast(totI, tpi)) * rn;
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < P && lane != tpi && !((double)pre < r));
                spi = first_or(hit, P - 1);
                // running total at the stop: every weight up to spi except the skipped one (also at a clamp on the skipped index, where
                // upstream's total does not hold the weight it then subtracts)
                const int64_t tot = bcast_i64(pre, spi), wi = (int64_t)bcast(totI, spi);
                zero_weight(wi == 0);
                rn = (r - (double)(tot - wi)) / (double)wi;
            }
            if (zero_w) return tpi;
            int hi, hcls = 0;
            {   // fastChoose(infectious[spi], totalInfectious[spi], rn): integer prefix sums over the list
                const int n = uni_i32(lane_get(nocc, spi)), base = 16 * uni_i32(lane_get(row0, spi));
                const double r = bcast(totI, spi) * rn;
                int64_t carry = 0, tot = 0, wi = 0;
                int hq = -1, hlast = -1;
                bool got = false;
                for (int t = 0; 64 * t < n; ++t) {
                    const int idx = 64 * t + lane;
                    const bool in = idx < n;
                    const int c = in ? ldCnt[base + idx] : 0, hv = in ? ldHap[base + idx] : 0;
                    const int64_t pre = iscan((int64_t)c, lane) + carry;
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64(in && !((double)pre < r));
                    const int last = min(63, n - 1 - 64 * t);
                    if (hit) {
                        const int l = (int)__builtin_ctzll(hit);
                        tot = bcast_i64(pre, l); wi = (int64_t)lane_get(c, l); hq = lane_get(hv, l);
                        got = true;
                        break;
                    }
                    carry = bcast_i64(pre, 63);
                    hlast = lane_get(hv, last); wi = (int64_t)lane_get(c, last);
                }
                if (!got) {          // clamp at H-1 (fc:26)
                    if (n > 0 && hap_of(hlast) == H - 1) { hq = hlast; tot = carry; } else { zero_w |= 1ull; hq = 0; wi = 1; }
                }
                zero_weight(wi == 0);
                rn = (r - (double)(tot - wi)) / (double)wi;
                hq = uni_i32(hq);
                hi = hap_of(hq); hcls = cls_of(hq);
            }
            int si = 0;
            if (GEN && S > 1) {   // fastChoose(susceptible[tpi], totalSusceptible[tpi], rn): integer weights over the groups
                const int64_t wv = lane < S ? (int64_t)ldS[tpi * S + lane] : 0;
                const int64_t pre = iscan(wv, lane);
                const double r = bcast(totS, tpi) * rn;
                si = first_or(__builtin_amdgcn_ballot_w64(lane < S && !((double)pre < r)), S - 1);
                const int64_t tot = bcast_i64(pre, si), wi = bcast_i64(wv, si);
                zero_weight(wi == 0);
                rn = (r - (double)(tot - wi)) / (double)wi;
            } else {
                const double wi = bcast(totS, tpi);
                const double r = wi * rn;
                zero_weight(wi == 0.0);
                rn = (r - (wi - wi)) / wi;
            }
            if (zero_w) return tpi;
            double p_accept;
            if (GEN) {
                const int bi = uni_i32(lane_get(cl_bidx, hcls));
                p_accept = gEff[spi * P + tpi] * bcast(cbb, bi) * ldSig[bi * S + si] / bcast(maxEBM, tpi);
            } else {
                p_accept = cold_args(ka)->la.effMig[spi * P + tpi] * c_b * c_sig / bcast(maxEBM, tpi);
            }
            ev_hap = hi; ev_pop = spi; ev_nh = si; ev_np = tpi;
            if (any_lane(rn < p_accept)) {
                list_add(tpi, hi, hcls);                             // NewInfections (pyx:246-251)
                if (zero_w) return tpi;
                if (GEN) group_move(tpi, si, 1.0, false);            // (GenerateMigration leaves immuneSourcePopRate[tpi, si] as it is, pyx:685-687)
                const double d1 = lane == tpi ? 1.0 : 0.0;
                totS -= d1; totI += d1;
                gI += 1.0;
                u_pi = tpi; f_infect = true; f_birth = true; f_immune = true;
                ev_type = EV_MIGRATION;
            }
        }
        PROF(10);
        // ---- UpdateRates(u_pi, ...) (pyx:516-546) ----
        if (u_pi >= 0) {
            if (GEN) {
                double inP = bcast(infP, u_pi), imP = bcast(immP, u_pi);
                if (f_infect) {
                    // BirthRate is formed again by EVERY infect-update (pyx:520-521), also a mutation's: immunity transitions since the
                    // population's last one have changed the susceptible counts it is built from
                    birth_rates_gen(u_pi);
                    double e0v, e1v, e2v, tEv;
                    class_rates(u_pi, e0v, e1v, e2v, tEv);
                    inP = refresh(u_pi, 0.0, tEv);
                }
                if (f_immune) imP = immune_sum(u_pi);
                infP = lane == u_pi ? inP : infP;
                immP = lane == u_pi ? imP : immP;
                popRate = lane == u_pi ? inP + imP : popRate;
                if (f_birth) remig();
            } else {
                double b = bcast(bC, u_pi);
                if (f_birth) {
                    b = birth_rate(u_pi);
                    bC = lane == u_pi ? b : bC;
                    remig();
                }
                const double inP = refresh(u_pi, tE_of(b, bcast(smul, u_pi)));
                popRate = lane == u_pi ? inP : popRate;
            }
            rescan_pop();
        }
        Rtot = totalRate + totalMig;
        log_event(ev_type, ev_hap, ev_pop, ev_nh, ev_np, den);
        PROF(11);
        return ret_pi;
    }
};

template <bool CLOCK, bool RCPDIV, bool GEN>
static __device__ __forceinline__ void lone_body() {
    const LoneKA ka = (LoneKA)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &a = ka->a;
    const auto &la = ka->la;
    const int64_t rep = blockIdx.x;
    if (rep >= a.n_replicates) return;
    const int lane = threadIdx.x;
    const auto &p = a.p;
    const auto &r = a.r;
    const int P = uni_i32(p.P), H = uni_i32(p.H), sites = uni_i32(p.sites);
    const int S = GEN ? uni_i32(p.S) : 1, C = GEN ? uni_i32(p.C) : 1, CB = GEN ? uni_i32(p.CB) : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgxLoneLayout L = vgx_lone_layout(P, la.lds_bytes, GEN ? S : 0, GEN ? CB : 0);

    Lone<CLOCK, RCPDIV, GEN> c;
    c.P = P; c.H = H; c.sites = sites; c.lane = lane;
    c.prow = uni_i32((P + 15) >> 4);
    c.nrows = uni_i32(L.nrows);
    c.S = S; c.C = C; c.CB = CB; c.nseg = GEN ? uni_i32(la.nseg) : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) c.M.m[k] = (lane & 15) >= k ? 1.0 : 0.0;
    c.ldRng = (double *)(smem + L.rng); c.ldStage = (uint32_t *)(smem + L.stage);
    c.ldRngK = (uint64_t *)(smem + L.rngk); c.ldRngS = (uint64_t *)(smem + L.rngs); c.ldCold = (int64_t *)(smem + L.cold);
    c.ldMig = (double *)(smem + L.mig);
    c.ldS = (double *)(smem + L.sus); c.ldSst = (double *)(smem + L.sst); c.ldImm = (double *)(smem + L.imm);
    c.ldBC = (double *)(smem + L.bc); c.ldSig = (double *)(smem + L.sig); c.ldTrans = (double *)(smem + L.trans);
    c.ldCum = (double *)(smem + L.cum); c.ldTend = (double *)(smem + L.tend);
    c.ldHap = (int32_t *)(smem + L.hap); c.ldCnt = (int32_t *)(smem + L.cnt);
    c.gEff = r.effMig + rep * P * P;
    c.c_b = p.cb_b[0]; c.c_sig = p.cb_sigma[0]; c.c_d = p.c_d[0]; c.c_tm = p.c_tm[0];
    const double c_s = p.c_s[0];
    for (int i = lane; i < P * P; i += 64) c.ldMig[i] = p.mig[i];
    // ---- general form: the class, birth-class, segment and group lanes; the small tables ----
    c.cl_d = 0.0; c.cl_s = 0.0; c.cl_tm = 0.0; c.cl_bidx = 0; c.cl_stype = 0; c.cbb = 0.0; c.cb_segl = -1;
    c.sg_sig = 0.0; c.sg_par = -1; c.sg_sn = 0; c.cumul_l = 0.0; c.no_imm = 1;
    if (GEN) {
        if (lane < C) { c.cl_d = p.c_d[lane]; c.cl_s = p.c_s[lane]; c.cl_tm = p.c_tm[lane]; c.cl_bidx = p.c_bidx[lane]; c.cl_stype = p.c_stype[lane]; }
        if (lane < CB) { c.cbb = p.cb_b[lane]; c.cb_segl = la.cb_seg[lane]; }
        if (lane < c.nseg) { c.sg_sig = la.seg_sig[lane]; c.sg_par = la.seg_par[lane]; c.sg_sn = la.seg_sn[lane]; }
        if (lane < S) c.cumul_l = p.suscepCumul[lane];
        c.no_imm = any_lane(c.cumul_l != 0.0) ? 0 : 1;
        for (int i = lane; i < CB * S; i += 64) c.ldSig[i] = p.cb_sigma[i];
        for (int i = lane; i < S * S; i += 64) c.ldTrans[i] = p.suscepTransition[i];
        for (int i = lane; i < P * CB; i += 64) c.ldBC[i] = 0.0;
    }

    // ---- start state from the layout of the other direct kernels (vgx_dev.h) ----
    double *gD = r.popD + rep * PD_COUNT * P;
    int64_t *gI = r.popI + rep * PI_COUNT * P;
    int32_t *gN = r.nocc + rep * P;
    bool ldp = false;
    {
        const bool in = lane < P;
        c.popRate = 0.0; c.cumPop = 0.0; c.migR = 0.0; c.cumMig = 0.0; c.bC = 0.0; c.infP = 0.0; c.immP = 0.0;
        c.maxEBM = (in && !GEN) ? la.maxEBM[lane] : 0.0;
        c.cd = in ? gD[PD_CD * P + lane] : 0.0;
        c.asz = in ? p.actualSizes[lane] : 1.0;
        c.rcp = in ? la.rcpAs[lane] : 1.0;
        c.mult = in ? p.sampMult[lane] : 0.0;
        c.smul = in ? c_s * p.sampMult[lane] : 0.0;
        c.totS = in ? (double)gI[PI_TOTSUS * P + lane] : 0.0;
        c.totI = 0.0;
        c.row0 = 0; c.rows = 0; c.nocc = 0;
        const bool on = in && gI[PI_LOCK * P + lane] != 0;
        const double sz = in ? (double)p.sizes[lane] : 0.0;
        const double thrOn = in ? p.startLD[lane] * sz : 0.0, thrOff = in ? p.endLD[lane] * sz : 0.0;
        c.sgnLD = on ? -1.0 : 1.0;
        c.thrCur = on ? thrOff : thrOn;
        // a population can switch on only if its threshold lies below its size, off only if it is on
        ldp = in && (thrOn < sz || on);
    }
    c.ld_any = (GEN && any_lane(ldp)) ? 1 : 0;
    if (GEN) {
        for (int i = lane; i < P * S; i += 64) {
            const double v = (double)r.sus[rep * P * S + i];
            c.ldS[i] = v; c.ldSst[i] = v; c.ldImm[i] = 0.0;
        }
    }
    c.has_mig = GEN ? 1 : uni_i32(la.has_mig[0] != 0 ? 1 : 0);
    c.mut_uni = uni_i32((la.mut_uniform && sites >= 1 && sites <= 16) ? 1 : 0);
    c.mut_m = (c.mut_uni && lane < sites) ? p.mRate[lane] : 0.0;
    c.mut_h = (c.mut_uni && lane < 3 * sites) ? p.hapMutType[lane] : 0.0;
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
    if (lane == 0) {
        c.ldCold[C_EV_PTR] = sc->ev_ptr; c.ldCold[C_LOOPS] = 0; c.ldCold[C_ATT_LOOPS] = 0; c.ldCold[C_TRAJ_NEXT] = 0;
        c.ldCold[C_RESTARTS] = 0; c.ldCold[C_ATT] = 0; c.ldCold[C_GOOD] = sc->good_attempt; c.ldCold[C_LAST_ATT] = -1;
        c.ldCold[C_LOC_N] = 0; c.ldCold[C_FA_N] = 0; c.ldCold[C_ATT_EV0] = sc->ev_ptr; c.ldCold[C_ATT_LOC0] = 0;
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
    int error = 0;
    if (!c.load_lists(gN, r.lhap + rep * P * r.cap, r.lcls + rep * P * r.cap, r.lcnt + rep * P * r.cap, r.cap)) c.zero_w |= LONE_F_FULL;

    const bool has_tlimit = !(a.time == -1.0f);
    const double tlimit = has_tlimit ? (double)a.time : __builtin_inf();
    c.tlimit = tlimit;
    const bool has_traj = r.traj != nullptr;
    c.next_tg = (CLOCK && has_traj && r.traj_points > 0) ? r.traj_t0 : __builtin_inf();
    const int record_events = a.record_events;

    // PrepareParameters tail (pyx:449-451): CheckLockdown for every population, UpdateAllRates
    if (!c.zero_w) {
        if (GEN && c.ld_any) c.check_lockdowns(ka, rep, 0, P);
        c.rebuild_all(ka);
    }

    for (int64_t att = 0; att < a.attempts && !error && !c.zero_w; ++att) {   // pyx:399-418
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
                const int64_t ev_ptr = c.cold_get(C_EV_PTR), loops = c.cold_get(C_LOOPS), att_loops = c.cold_get(C_ATT_LOOPS);
                {
                    const int64_t cS = (int64_t)bcast_i64((int64_t)c.cnt, EV_SAMPLING);
                    const bool go = ev_ptr < a.ev_size && (a.sample_size == -1 || cS <= a.sample_size) && (!has_tlimit || any_lane(c.currentTime < tlimit));
                    if (!go) break;
                    if (loops >= a.max_loop) { error = ERR_LOOP_GUARD; break; }
                    c.ev_left0 = (int)min((int64_t)LONE_BIG, a.ev_size - ev_ptr);
                    c.loop_left0 = (int)min((int64_t)LONE_BIG, a.max_loop - loops);
                    c.s_left = a.sample_size == -1 ? LONE_BIG : (int)min((int64_t)LONE_BIG, a.sample_size - cS + 1);
                    c.ev_left = c.ev_left0; c.loop_left = c.loop_left0;
                    c.iter_base = (uint32_t)att_loops + (uint32_t)c.loop_left0;
                }
                for (;;) {
                    int pi = 0;
                    const int why = c.fast_loop(pi);
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
                    if (why == c.FAST_SLOW) pi = c.iteration(ka, rep);
                    // after an iteration: a full stage, an error, extinction (pyx:410-411), CheckLockdown (pyx:412)
                    if (c.stage_n == 64) {
                        if (record_events) c.stage_flush(ka, rep);
                        c.stage_n = 0;
                    }
                    if (c.zero_w) { done = true; break; }
                    if (any_lane(c.totalRate == 0.0) || any_lane(c.gI == 0.0)) { done = true; break; }
                    if (GEN && c.crossed(pi)) {
                        if (c.check_lockdowns(ka, rep, pi, pi + 1)) c.rebuild_all(ka);
                        if (c.zero_w) { done = true; break; }
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
        if (c.zero_w || error) break;
        const int64_t ev_ptr = c.cold_get(C_EV_PTR);
        if (ev_ptr <= 100 && a.iterations > 100) {
            // Restart (pyx:714-738): compartments back to the initial snapshot, CheckLockdown for every population, UpdateAllRates;
            // swapLockdown and the lockdown state survive.  Lockdown records of the failed attempt stay in the log: keep the (rate,
            // iteration) pairs the host clock needs for them
            if (GEN) {
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
                if (lane == 0) { c.ldCold[C_FA_N] = fa_n; c.ldCold[C_ATT_EV0] = 0; c.ldCold[C_ATT_LOC0] = loc_n; }
            }
            if (lane == 0) {
                c.ldCold[C_EV_PTR] = 0; c.ldCold[C_TRAJ_NEXT] = 0;
                c.ldCold[C_RESTARTS] = c.ldCold[C_RESTARTS] + 1;
                c.ldCold[C_ATT] = att + 1; c.ldCold[C_ATT_LOOPS] = 0;   // the CheckLockdown below belongs to the next attempt, before its first iteration
            }
            c.cnt = lane == CNT_SWAP ? c.cnt : 0ull;
            c.currentTime = 0.0;
            c.next_tg = (CLOCK && has_traj && r.traj_points > 0) ? r.traj_t0 : __builtin_inf();
            WSYNC();
            if (!c.load_lists(r.i_nocc, r.i_hap, r.i_cls, r.i_cnt, r.i_cap)) { c.zero_w |= LONE_F_FULL; break; }
            if (GEN) {
                for (int i = lane; i < P * S; i += 64) c.ldS[i] = (double)r.i_sus[i];
                WSYNC();
                double ts = 0.0;
                if (lane < P)
                    for (int sn = 0; sn < S; ++sn) ts += c.ldS[lane * S + sn];
                c.totS = ts;
            } else {
                c.totS = lane < P ? (double)r.i_sus[lane] : 0.0;
            }
            c.gI = (double)bcast_i64(iscan((int64_t)c.totI, lane), 63);
            if (GEN && c.ld_any) c.check_lockdowns(ka, rep, 0, P);
            c.rebuild_all(ka);
        } else {
            if (lane == 0) c.ldCold[C_GOOD] = att + 1;
            WSYNC();
            break;
        }
    }
    if (c.zero_w) error = (c.zero_w >> 32) & 4ull ? (ERR_CAPACITY | (VGX_LONE_FULL_SITE << 8)) : (c.zero_w >> 32) & 2ull ? ERR_CAPACITY : ERR_ZERO_WEIGHT;
    if (has_traj && !(c.zero_w & LONE_F_FULL)) c.traj_emit(ka, rep, 0.0, true);

    // ---- end state back in the layout of the other direct kernels ----
    if (!(c.zero_w & LONE_F_FULL)) {
        for (int pn = 0; pn < P; ++pn) {
            int32_t *lh = r.lhap + (rep * P + pn) * r.cap, *lc = r.lcls + (rep * P + pn) * r.cap;
            int64_t *ln = r.lcnt + (rep * P + pn) * r.cap, *lt = r.ltsum + (rep * P + pn) * r.capT;
            const int n = uni_i32(lane_get(c.nocc, pn)), base = 16 * uni_i32(lane_get(c.row0, pn));
            if (n > r.cap) { error = ERR_CAPACITY; continue; }
            for (int t = 0; 64 * t < n; ++t) {
                const int idx = 64 * t + lane;
                const int cv = idx < n ? c.ldCnt[base + idx] : 0;
                if (idx < n) {
                    const int w = c.ldHap[base + idx];
                    lh[idx] = c.hap_of(w); lc[idx] = c.cls_of(w); ln[idx] = (int64_t)cv;
                }
                const int64_t ts = bcast_i64(iscan((int64_t)cv, lane), 63);
                if (lane == 0 && t < r.capT) lt[t] = ts;          // the tile sums of the row kernels (kept while a list is longer than a tile)
            }
            for (int j = ((n + 63) >> 6) + lane; j < r.capT; j += 64) lt[j] = 0;
            if (n == 0 && lane == 0) lt[0] = 0;
            if (lane == 0) gN[pn] = n;
        }
        if (lane < P) {
            gD[PD_POPRATE * P + lane] = c.popRate;
            gD[PD_INFECT * P + lane] = GEN ? c.infP : c.popRate;
            gD[PD_IMMUNE * P + lane] = GEN ? c.immP : 0.0;
            gD[PD_MIG * P + lane] = c.maxEBM * c.totS * (c.gI - c.totI);
            gD[PD_MAXEBM * P + lane] = c.maxEBM;
            gI[PI_TOTSUS * P + lane] = (int64_t)c.totS;
            gI[PI_TOTINF * P + lane] = (int64_t)c.totI;
            if (GEN) {
                gD[PD_CD * P + lane] = c.cd;
                gI[PI_LOCK * P + lane] = c.sgnLD < 0.0 ? 1 : 0;
            } else {
                r.sus[rep * P + lane] = (int64_t)c.totS;
                r.immSrc[rep * P + lane] = 0.0;
            }
        }
        if (GEN)
            for (int i = lane; i < P * S; i += 64) {
                r.sus[rep * P * S + i] = (int64_t)c.ldS[i];
                r.immSrc[rep * P * S + i] = c.ldImm[i];
            }
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
            sc->fa_n = c.ldCold[C_FA_N];
        }
    }
}

}  // namespace

#define LONE_KERNEL(name, CLOCK, RCPDIV, GEN) \
    extern "C" __global__ void __launch_bounds__(64, 1) name(VgxLoneKArgs) { lone_body<CLOCK, RCPDIV, GEN>(); }
LONE_KERNEL(vgx_lone_kernel, false, true, false)
LONE_KERNEL(vgx_lone_kernel_clock, true, true, false)
LONE_KERNEL(vgx_lone_kernel_plaindiv, false, false, false)      // validation: VGX_SOLO_PLAIN_DIV=1
LONE_KERNEL(vgx_lone_gen_kernel, false, true, true)
LONE_KERNEL(vgx_lone_gen_kernel_clock, true, true, true)
LONE_KERNEL(vgx_lone_gen_kernel_plaindiv, false, false, true)

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_lone(const VgxDirectArgs *a, const VgxLoneArgs *la, int clock,
                                                                            hipStream_t stream) {
    void (*k)(VgxLoneKArgs);
    if (la->general) k = clock ? vgx_lone_gen_kernel_clock : la->exact_rcp_div ? vgx_lone_gen_kernel : vgx_lone_gen_kernel_plaindiv;
    else k = clock ? vgx_lone_kernel_clock : la->exact_rcp_div ? vgx_lone_kernel : vgx_lone_kernel_plaindiv;
    hipError_t err = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, la->lds_bytes);
    if (err != hipSuccess) return err;
    VgxLoneKArgs ka;
    ka.a = *a; ka.la = *la;
    if (getenv("VGX_TIMING")) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k, 64, (size_t)la->lds_bytes) == hipSuccess)
            fprintf(stderr, "vgx_lone%s: LDS %d B per wavefront (%d heap rows), %d wavefronts per CU\n", la->general ? " (general form)" : "",
                    (int)la->lds_bytes, vgx_lone_layout(a->p.P, la->lds_bytes, la->general ? a->p.S : 0, la->general ? a->p.CB : 0).nrows, nb);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)a->n_replicates), dim3(64), (size_t)la->lds_bytes, stream, ka);
    return hipGetLastError();
}
