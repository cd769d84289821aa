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
// wave exploits the parallelism INSIDE an event: lanes map to populations (all P-wide arrays and the
// per-class tables live in LDS for the whole run), to the entries of the chosen population's ordered
// occupancy list (one coalesced 16 B/entry stream from HBM/L2 per event, kept in registers between the
// selection scan and the rate refresh), and to the terms of the transmission sum.  Everything the
// reference sums left-to-right is summed left-to-right here too (vgx_wave.h: seq_sum / seq_scan), with
// no FMA contraction (-ffp-contract=off), so every rate, every comparison and the recycled random
// number are bit-identical to the reference's.  Three restructurings keep that property:
//   * only haplotypes with a non-zero count are visited: x + 0.0 == x and a zero weight can never stop a
//     scan;
//   * per-haplotype rates are factored through CLASSES of identical parameter rows (bRate, susceptibility
//     row, dRate, sRate, sum of mRate, suscType): BirthRate (pyx:382-392) depends on the haplotype only
//     through its class, so it is evaluated once per class — same operations on the same operands;
//   * the running prefix sums of popRate / migPopRate that fastChoose would recompute (fast_choose.pxi:25)
//     are kept (they ARE the partial sums of the totalRate / totalMigrationRate loops, pyx:537-546) and
//     refreshed from the changed population onwards, so population selection is one compare + ballot.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"

#define LANES VGX_LANES

// In-kernel stamps (diagnostic build only, -DVGX_PROFILE: tools/profile_phases.py): cycles per phase of the
// event loop, summed per replicate into the debug buffer r.prof.  No stamp executes in the product build.
#ifdef VGX_PROFILE
#define PROF(i)                                                       \
    do {                                                              \
        unsigned long long prof_t1 = __builtin_readcyclecounter();    \
        c.prof_acc[i] += prof_t1 - c.prof_t0;                         \
        c.prof_t0 = prof_t1;                                          \
    } while (0)
#else
#define PROF(i)
#endif

enum { ERR_ZERO_WEIGHT = 3, ERR_CAPACITY = 4, ERR_LOOP_GUARD = 5 };
enum { EV_BIRTH = 0, EV_DEATH, EV_SAMPLING, EV_MUTATION, EV_SUSCCHANGE, EV_MIGRATION };
enum { CNT_B = 0, CNT_D, CNT_S, CNT_M, CNT_I, CNT_SWAP, CNT_MIGP, CNT_MIGN };
#define BUMP(i) do { if (c.lane == 0) c.cnt[i] += 1; } while (0)

// the occupancy-list tile of one population held in registers (lane k <-> entry k) while it has <= 64 entries
struct Tile {
    bool valid;
    int pi, n;
    int hap, cls;
    int64_t cnt;
    double mrow0;  // migrationRates[pi][lane], fetched together with the tile (first 64 populations)
};

// 64 PCG64 outputs per refill: lane k jumps the stream k+1 steps ahead (state*a^(k+1) + inc*(a^k+...+1), exact
// 128-bit arithmetic), so the uniforms are the reference's, in order.  Even lanes hold -log(u) for SampleTime
// (pyx:477), odd lanes the uniform of GenerateEvent (pyx:488): 32 loop iterations per refill, and the logarithm
// is evaluated 32-wide instead of once per iteration on one lane.
struct RngBatch {
    uint64_t Ah, Al, Gh, Gl;   // per lane: a^(lane+1), sum_{j<=lane} a^j
    uint64_t sh, sl, ih, il;   // stream position before the batch; increment
    double val;
    int pos;                   // iterations consumed from the current batch (32 = empty)
    // counter-based variant (FAST mode, vgx_run_opts.mode = 2): outputs base .. base + 63 of vgx_philox_stream_*
    int philox;
    uint64_t seed, base;
    uint32_t att;
};
static __device__ __forceinline__ void rng_init_lane(RngBatch &g, int lane) {
    const uint64_t MH = 0x2360ED051FC65DA4ull, ML = 0x4385DF649FCCF645ull;
    uint64_t Ah = MH, Al = ML, Gh = 0, Gl = 1;
    for (int j = 1; j < LANES; ++j) {
        uint64_t nh, nl, gh, gl;
        vgx_mul128(Ah, Al, MH, ML, nh, nl);
        vgx_mul128(Gh, Gl, MH, ML, gh, gl);
        vgx_add128(gh, gl, 0, 1);
        if (j <= lane) { Ah = nh; Al = nl; Gh = gh; Gl = gl; }
    }
    g.Ah = Ah; g.Al = Al; g.Gh = Gh; g.Gl = Gl;
    g.pos = 32;
}
static __device__ __forceinline__ void rng_seed(RngBatch &g, uint64_t seed, uint32_t attempt) {
    VgxPcg64 s;
    vgx_pcg64_seed(s, seed, attempt);
    g.sh = s.sh; g.sl = s.sl; g.ih = s.ih; g.il = s.il;
    g.pos = 32;
    g.seed = seed; g.att = attempt; g.base = 0;
}
static __device__ __forceinline__ void rng_refill(RngBatch &g, int lane) {
    if (g.philox) {   // every lane forms its own output of the counter-based stream
        const double u = vgx_philox_stream_double(g.seed, g.att, g.base + (uint64_t)lane);
        g.val = (lane & 1) ? u : -vgx_log(u);
        g.base += LANES;
        g.pos = 0;
        return;
    }
    uint64_t h, l, ch, cl;
    vgx_mul128(g.Ah, g.Al, g.sh, g.sl, h, l);
    vgx_mul128(g.Gh, g.Gl, g.ih, g.il, ch, cl);
    vgx_add128(h, l, ch, cl);
    double u = vgx_pcg64_output_double(h, l);
    g.val = (lane & 1) ? u : -vgx_log(u);
    g.sh = (uint64_t)bcast_i64((int64_t)h, LANES - 1);
    g.sl = (uint64_t)bcast_i64((int64_t)l, LANES - 1);
    g.pos = 0;
}

struct Ctx {
    int P, S, H, C, CB, sites, lane;
    const VgxDevParams *p;
    // ---- LDS, f64 ----
    double *popRate, *infect, *immune, *migRate, *maxEBM, *cd, *as, *cum, *cumMig, *sampMult;  // [P]
    double *immSrc;    // [P][S]  immuneSourcePopRate
    double *birthC;    // [P][CB] eventHapPopRate[.,.,0] per birth class as of the population's last infect-update
    double *xC;        // [P][CB][S] susceptHapPopRate per birth class, same staleness
    double *tE;        // [C] tEventHapPopRate per class for the population being processed
    double *c_d, *c_s, *c_tm;  // [C]
    double *cb_b;      // [CB]
    double *cb_sigma;  // [CB][S]
    double *kmig;      // [P] FAST mode: sum_pn m[pi][pn]^2 * cd[pn] / as[pn], constant between lockdown switches
    double *cumul;     // [S] suscepCumulTransition
    double *trans;     // [S][S]
    // ---- LDS, i64 / i32 ----
    int64_t *totalSus, *totalInf, *lockON;  // [P]
    int64_t *sus;      // [P][S]
    int32_t *nocc;     // [P]
    int32_t *c_bidx, *c_stype;  // [C]
    // ---- global, this replicate ----
    double *effMig;
    int32_t *lhap, *lcls;
    int64_t *lcnt;
    int64_t cap;
    int64_t *ltsum;    // per 64-entry tile of each list: sum of its counts
    int64_t capT;
    double *ev_rate;
    int32_t *ev_cols;
    int32_t *loc_rec;
    double *loc_time;
    int64_t *loc_iter;
    int64_t loc_cap;
    double den;        // totalRate + totalMigrationRate of the current iteration's time step (pyx:477)
    int64_t iter_key;  // (attempt << 40) | loop iteration of the attempt (0 outside the event loop)
    double *traj;
    int64_t traj_points, traj_next;
    int64_t *rec;          // forward record of recombinant births [rec_cap][5] (models.pxi:69-89), or null
    int64_t rec_cap, rec_n;
    double traj_t0, traj_dt;
    int64_t evcap, ev_base;
    int record_events;
    // ---- wave-uniform scalars ----
    double currentTime, totalRate, totalMig, rn;
    int64_t gI;
    int64_t *cnt;      // LDS [8]: bCounter, dCounter, sCounter, mCounter, iCounter, swapLockdown, migPlus, migNonPlus
    int64_t ev_ptr, ev_size, loc_n;
    int error;
    int fast;      // compile-time constant of the instantiation: order-free sums instead of the reference's order
    bool has_mig;  // some maxEffectiveBirthMigration > 0
    bool mig_cum_ok;  // cumMig[] matches migRate[]
    bool ld_any;   // some population can switch its lockdown state at all
#ifdef VGX_PROFILE
    unsigned long long prof_t0, prof_acc[VGX_PROF_SLOTS];
#endif
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

// fastChoose over a P-sized f64 array whose serial prefix sums are cached in `cum`
// (cum[k] == the reference's running `total` at index k, bit for bit).
static __device__ __forceinline__ int choose_prefix(Ctx &c, const double *w, const double *cum, int n, double tw, double &rn) {
    const int lane = c.lane;
    double r = tw * rn;
    for (int base = 0; base < n; base += LANES) {
        int k = base + lane;
        double pre = (k < n) ? cum[k] : 0.0;
        unsigned long long hit = __ballot((k < n) && !(pre < r));
        if (hit) {
            int j = base + __ffsll((long long)hit) - 1;
            double total = cum[j], wi = w[j];
            if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
            rn = (r - (total - wi)) / wi;
            return j;
        }
    }
    double wi = w[n - 1], total = cum[n - 1];  // clamp at n-1 (fast_choose.pxi:26)
    if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
    rn = (r - (total - wi)) / wi;
    return n - 1;
}

// ------------------------------------------------------------------------------------------------
// Occupancy list of a population: ordered (hap, class, count) entries in global memory.
static __device__ __forceinline__ int32_t *LH(Ctx &c, int pi) { return c.lhap + (int64_t)pi * c.cap; }
static __device__ __forceinline__ int32_t *LC(Ctx &c, int pi) { return c.lcls + (int64_t)pi * c.cap; }
static __device__ __forceinline__ int64_t *LN(Ctx &c, int pi) { return c.lcnt + (int64_t)pi * c.cap; }
static __device__ __forceinline__ int64_t *LT(Ctx &c, int pi) { return c.ltsum + (int64_t)pi * c.capT; }

static __device__ __forceinline__ void tile_load(Ctx &c, Tile &t, int pi) {
    t.pi = pi;
    t.n = c.nocc[pi];
    t.valid = t.n <= LANES;
    t.hap = 0; t.cls = 0; t.cnt = 0;
    t.mrow0 = (c.lane < c.P) ? c.p->mig[(int64_t)pi * c.P + c.lane] : 0.0;
    if (t.valid && c.lane < t.n) {
        t.hap = LH(c, pi)[c.lane];
        t.cls = LC(c, pi)[c.lane];
        t.cnt = LN(c, pi)[c.lane];
    }
}

// first index whose haplotype is >= target (n if none); found tells whether it is the target itself.
// Two levels: the first haplotypes of the 64-entry tiles (one strided load per 64 tiles), then one tile.
static __device__ __forceinline__ int list_lower_bound(Ctx &c, int pi, int target, bool &found) {
    const int n = c.nocc[pi];
    const int32_t *lh = LH(c, pi);
    int tile = 0;  // last tile whose first haplotype is <= target (tile 0 if there is none)
    if (n > LANES) {
        const int nt = (n + LANES - 1) / LANES;
        for (int tb = 0; tb < nt; tb += LANES) {
            int j = tb + c.lane;
            int h0 = (j < nt) ? lh[(int64_t)j * LANES] : 0x7fffffff;
            int nle = __popcll(__ballot(h0 <= target));
            if (nle > 0) tile = tb + nle - 1;
            if (nle < LANES) break;
        }
    }
    const int base = tile * LANES;
    int k = base + c.lane;
    int h = (k < n) ? lh[k] : 0x7fffffff;
    unsigned long long ge = __ballot(h >= target);
    if (ge) {
        int j = __ffsll((long long)ge) - 1;
        int hj = __builtin_amdgcn_readlane(h, j);
        found = (hj == target);
        return base + j;
    }
    found = false;  // the whole tile is below the target and the next tile starts above it
    return min(n, base + LANES);
}

#define VGX_SHIFT_U 4   // 64-entry chunks moved per step of a list shift (loads of a step are in flight together)

static __device__ __forceinline__ void list_insert_at(Ctx &c, int pi, int pos, int hap, int cls, int64_t cnt) {
    int n = c.nocc[pi];
    if (n >= c.cap) { c.error = ERR_CAPACITY; return; }
    int32_t *lh = LH(c, pi), *lc = LC(c, pi);
    int64_t *ln = LN(c, pi), *lt = LT(c, pi);
    // tile sums (kept while a list is longer than one tile): tile j gains the entry entering from below (the new
    // one for pos's tile) and loses its last entry
    if (n > LANES) {
        const int jp = pos / LANES, jl = n / LANES;
        for (int tb = jp; tb <= jl; tb += LANES) {
            int j = tb + c.lane;
            if (j <= jl) {
                int64_t in = (j == jp) ? cnt : ln[(int64_t)j * LANES - 1];
                int kout = j * LANES + LANES - 1;
                int64_t out = (kout < n) ? ln[kout] : 0;
                lt[j] += in - out;
            }
        }
        WSYNC();
    }
    // shift [pos, n) one slot up, highest block first; each block is read completely before it is written
    for (int hi = n; hi > pos; hi -= VGX_SHIFT_U * LANES) {
        int lo = max(pos, hi - VGX_SHIFT_U * LANES);
        int h[VGX_SHIFT_U], cl[VGX_SHIFT_U];
        int64_t ct[VGX_SHIFT_U];
#pragma unroll
        for (int u = 0; u < VGX_SHIFT_U; ++u) {
            int k = lo + u * LANES + c.lane;
            h[u] = 0; cl[u] = 0; ct[u] = 0;
            if (k < hi) { h[u] = lh[k]; cl[u] = lc[k]; ct[u] = ln[k]; }
        }
        WSYNC();
#pragma unroll
        for (int u = 0; u < VGX_SHIFT_U; ++u) {
            int k = lo + u * LANES + c.lane;
            if (k < hi) { lh[k + 1] = h[u]; lc[k + 1] = cl[u]; ln[k + 1] = ct[u]; }
        }
        WSYNC();
    }
    if (c.lane == 0) { lh[pos] = hap; lc[pos] = cls; ln[pos] = cnt; c.nocc[pi] = n + 1; }
    WSYNC();
    if (n == LANES) {  // the list outgrows one tile: start its tile sums
        int64_t s0 = bcast_i64(iscan(ln[c.lane], c.lane), LANES - 1);
        if (c.lane == 0) { lt[0] = s0; lt[1] = ln[LANES]; }
        WSYNC();
    }
}

static __device__ __forceinline__ void list_remove_at(Ctx &c, int pi, int pos) {
    int n = c.nocc[pi];
    int32_t *lh = LH(c, pi), *lc = LC(c, pi);
    int64_t *ln = LN(c, pi), *lt = LT(c, pi);
    // tile sums: tile j loses its first entry (the removed one for pos's tile) and gains the first of the next tile
    if (n > LANES) {
        const int jp = pos / LANES, jl = (n - 1) / LANES;
        for (int tb = jp; tb <= jl; tb += LANES) {
            int j = tb + c.lane;
            if (j <= jl) {
                int64_t out = (j == jp) ? ln[pos] : ln[(int64_t)j * LANES];
                int kin = j * LANES + LANES;
                int64_t in = (kin < n) ? ln[kin] : 0;
                lt[j] += in - out;
            }
        }
        WSYNC();
    }
    for (int lo = pos + 1; lo < n; lo += VGX_SHIFT_U * LANES) {
        int h[VGX_SHIFT_U], cl[VGX_SHIFT_U];
        int64_t ct[VGX_SHIFT_U];
#pragma unroll
        for (int u = 0; u < VGX_SHIFT_U; ++u) {
            int k = lo + u * LANES + c.lane;
            h[u] = 0; cl[u] = 0; ct[u] = 0;
            if (k < n) { h[u] = lh[k]; cl[u] = lc[k]; ct[u] = ln[k]; }
        }
        WSYNC();
#pragma unroll
        for (int u = 0; u < VGX_SHIFT_U; ++u) {
            int k = lo + u * LANES + c.lane;
            if (k < n) { lh[k - 1] = h[u]; lc[k - 1] = cl[u]; ln[k - 1] = ct[u]; }
        }
        WSYNC();
    }
    if (c.lane == 0) c.nocc[pi] = n - 1;
    WSYNC();
}

// infectious[pi, hap] += delta (delta = +1 / -1), keeping the list ordered and free of zero counts
static __device__ __forceinline__ void list_add(Ctx &c, int pi, int hap, int64_t delta) {
    bool found;
    int pos = list_lower_bound(c, pi, hap, found);
    if (found) {
        int64_t *ln = LN(c, pi);
        int64_t v = ln[pos] + delta;  // uniform address
        if (v == 0) {
            list_remove_at(c, pi, pos);
        } else {
            if (c.lane == 0) { ln[pos] = v; if (c.nocc[pi] > LANES) LT(c, pi)[pos / LANES] += delta; }
            WSYNC();
        }
    } else {
        list_insert_at(c, pi, pos, hap, c.p->cls[hap], delta);
    }
}

// ------------------------------------------------------------------------------------------------
// BirthRate per class (pyx:382-392): ps += susceptHapPopRate * m * m * cd / as over (sn, pn), in order.
static __device__ __forceinline__ void birth_update(Ctx &c, int pi, const Tile &t) {
    const int P = c.P, S = c.S, lane = c.lane;
    const double *mrow = c.p->mig + (int64_t)pi * P;
    const bool pref = (t.pi == pi);
    for (int cb = 0; cb < c.CB; ++cb) {
        double ps = 0.0;
        for (int sn = 0; sn < S; ++sn) {
            double x = (double)c.sus[pi * S + sn] * c.cb_sigma[cb * S + sn];
            if (lane == 0) c.xC[(pi * c.CB + cb) * S + sn] = x;
            if (c.fast) {  // factored: the sum over pn is constant between lockdown switches
                ps += x * c.kmig[pi];
                continue;
            }
            for (int base = 0; base < P; base += LANES) {
                int pn = base + lane;
                double tv = 0.0;
                if (pn < P) {
                    double m = (pref && base == 0) ? t.mrow0 : mrow[pn];
                    tv = x * m * m * c.cd[pn] / c.as[pn];
                }
                ps = seq_sum(tv, min(LANES, P - base), ps);
            }
        }
        if (lane == 0) c.birthC[pi * c.CB + cb] = c.cb_b[cb] * ps;
    }
    WSYNC();
}

// tEventHapPopRate per class (pyx:522-526) from the cached birth rates of population pi
static __device__ __forceinline__ void tE_fill(Ctx &c, int pi) {
    const double mult = c.sampMult[pi];
    for (int base = 0; base < c.C; base += LANES) {
        int k = base + c.lane;
        if (k < c.C) {
            double b = c.birthC[pi * c.CB + c.c_bidx[k]];
            c.tE[k] = ((b + c.c_d[k]) + c.c_s[k] * mult) + c.c_tm[k];
        }
    }
    WSYNC();
}

// A long occupancy list is read 64 entries at a time with the loads of the next VGX_STREAM_DEPTH chunks already in
// flight (the serial sum of one chunk is much shorter than an HBM round trip).
#define VGX_STREAM_DEPTH 4
struct ListStream {
    const int32_t *lc;
    const int64_t *ln;
    int64_t cnt[VGX_STREAM_DEPTH];
    int cls[VGX_STREAM_DEPTH];
};
static __device__ __forceinline__ void stream_load(Ctx &c, ListStream &ls, int slot, int base, int n) {
    int k = base + c.lane;
    ls.cnt[slot] = 0;
    ls.cls[slot] = 0;
    if (k < n) {
        ls.cnt[slot] = ls.ln[k];
        if (c.C != 1) ls.cls[slot] = ls.lc[k];
    }
}
static __device__ __forceinline__ void stream_open(Ctx &c, ListStream &ls, int pi, int n) {
    ls.lc = LC(c, pi);
    ls.ln = LN(c, pi);
#pragma unroll
    for (int d = 0; d < VGX_STREAM_DEPTH; ++d) stream_load(c, ls, d, d * LANES, n);
}
// weight tEvent * infectious of entry base + lane (0.0 beyond the list); refills the freed slot
static __device__ __forceinline__ double stream_next(Ctx &c, ListStream &ls, int base, int n) {
    int64_t cn = ls.cnt[0];
    int cl = ls.cls[0];
#pragma unroll
    for (int d = 0; d + 1 < VGX_STREAM_DEPTH; ++d) { ls.cnt[d] = ls.cnt[d + 1]; ls.cls[d] = ls.cls[d + 1]; }
    stream_load(c, ls, VGX_STREAM_DEPTH - 1, base + VGX_STREAM_DEPTH * LANES, n);
    return (base + c.lane < n) ? c.tE[cl] * (double)cn : 0.0;
}

// infectPopRate[pi] = sum over occupied haplotypes, in haplotype order, of tEvent * infectious (pyx:519-528)
static __device__ __forceinline__ double row_sum(Ctx &c, int pi, const Tile &t) {
    if (c.fast) {
        if (c.C == 1) return c.tE[0] * (double)c.totalInf[pi];  // one rate class: sum_h tE * infectious[pi, h]
        // several classes: tree sums over the lanes, chunk after chunk (order-free; haplotype order is kept)
        if (t.valid && t.pi == pi) {
            double w = (c.lane < t.n) ? c.tE[t.cls] * (double)t.cnt : 0.0;
            return bcast(fscan(w), LANES - 1);
        }
        const int n = c.nocc[pi];
        ListStream ls;
        stream_open(c, ls, pi, n);
        double acc = 0.0;
        for (int base = 0; base < n; base += LANES) acc += bcast(fscan(stream_next(c, ls, base, n)), LANES - 1);
        return acc;
    }
    if (t.valid && t.pi == pi) {
        double w = (c.lane < t.n) ? c.tE[t.cls] * (double)t.cnt : 0.0;
        return seq_sum(w, t.n, 0.0);
    }
    const int n = c.nocc[pi];
    ListStream ls;
    stream_open(c, ls, pi, n);
    double acc = 0.0;
    for (int base = 0; base < n; base += LANES) {
        double w = stream_next(c, ls, base, n);
        acc = seq_sum(w, min(LANES, n - base), acc);
    }
    return acc;
}

// fastChoose(infectious[pi], tw = totalInfectious[pi], rn) (fast_choose.pxi:18-31, int64 weights) over the occupancy
// list: integer prefix sums are order-free, so the tile sums give the tile and one tile scan gives the entry.
// Returns the list index and recycles rn like the reference.
static __device__ __forceinline__ int count_select(Ctx &c, int pi, const Tile &t, int64_t tw, double &rn) {
    const int lane = c.lane;
    const double r = (double)tw * rn;
    const bool reg = t.valid && t.pi == pi;
    const int n = reg ? t.n : c.nocc[pi];
    const int64_t *ln = LN(c, pi);
    int64_t before = 0;
    int base = 0;
    bool none = false;
    if (!reg && n > LANES) {
        const int nt = (n + LANES - 1) / LANES;
        const int64_t *lt = LT(c, pi);
        int jt = -1;
        int64_t carry = 0;
        for (int tb = 0; tb < nt && jt < 0; tb += LANES) {
            int j = tb + lane;
            int64_t w = (j < nt) ? lt[j] : 0;
            int64_t pre = iscan(w, lane) + carry;
            unsigned long long hit = __ballot(j < nt && !((double)pre < r));
            if (hit) {
                int q = __ffsll((long long)hit) - 1;
                jt = tb + q;
                before = bcast_i64(pre, q) - bcast_i64(w, q);
            }
            carry = bcast_i64(pre, LANES - 1);
        }
        if (jt < 0) { none = true; before = carry; } else base = jt * LANES;
    }
    int q = -1;
    int64_t total = before, wi = 0;
    if (!none) {
        int k = base + lane;
        int64_t w = reg ? ((lane < n) ? t.cnt : 0) : ((k < n) ? ln[k] : 0);
        int64_t pre = iscan(w, lane) + before;
        unsigned long long hit = __ballot(k < n && !((double)pre < r));
        if (hit) {
            q = __ffsll((long long)hit) - 1;
            total = bcast_i64(pre, q);
            wi = bcast_i64(w, q);
        } else {
            total = bcast_i64(pre, LANES - 1);
        }
    }
    if (q < 0) {
        // nothing reached r: the dense loop runs on to index H-1 (fast_choose.pxi:26); that is a valid pick only if
        // haplotype H-1 is occupied, otherwise the reference reports a zero weight
        if (n > 0 && LH(c, pi)[n - 1] == c.H - 1) {
            wi = ln[n - 1];
            rn = (r - (double)(total - wi)) / (double)wi;
            return n - 1;
        }
        c.error = ERR_ZERO_WEIGHT;
        return 0;
    }
    rn = (r - (double)(total - wi)) / (double)wi;
    return base + q;
}

// fastChoose(hapPopRate[pi], infectPopRate[pi], rn) over the occupied entries; returns the list index
static __device__ __forceinline__ int row_select(Ctx &c, int pi, const Tile &t, double tw, double &rn) {
    if (c.fast && c.C == 1) return count_select(c, pi, t, c.totalInf[pi], rn);  // one rate class: weights tE * count
    const int lane = c.lane;
    const int n = t.valid ? t.n : c.nocc[pi];
    const int32_t *lc = LC(c, pi);
    const int64_t *ln = LN(c, pi);
    const double r = tw * rn;
    double carry = 0.0, w = 0.0;
    int base = 0, k0 = 0, nn = n;
    if (c.fast) {  // several classes: tree prefix sums chunk by chunk, first entry whose prefix reaches r
        ListStream ls;
        if (!t.valid) stream_open(c, ls, pi, n);
        for (; base < n; base += LANES) {
            w = t.valid ? ((lane < n) ? c.tE[t.cls] * (double)t.cnt : 0.0) : stream_next(c, ls, base, n);
            double pre = fscan(w) + carry;
            unsigned long long hit = __ballot(base + lane < n && !(pre < r));
            if (hit) {
                int j = __ffsll((long long)hit) - 1;
                double total = bcast(pre, j), wi = bcast(w, j);
                if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
                rn = (r - (total - wi)) / wi;
                return base + j;
            }
            carry = bcast(pre, LANES - 1);
        }
        // rounding left the total below r: the last occupied entry (the reference clamps at the end of the array)
        if (n > 0) {
            double wi = c.tE[t.valid ? __builtin_amdgcn_readlane(t.cls, n - 1) : lc[n - 1]] * (double)(t.valid ? bcast_i64(t.cnt, n - 1) : ln[n - 1]);
            rn = (r - (carry - wi)) / wi;
            return n - 1;
        }
        c.error = ERR_ZERO_WEIGHT;
        return 0;
    }
    bool inside = true;   // the running sum reaches r inside lanes [k0, nn) of the chunk at `base`
    if (t.valid) {
        w = (lane < n) ? c.tE[t.cls] * (double)t.cnt : 0.0;
    } else {
        // long list: the running sum advances one row of 16 entries at a time (one fmac per entry, vgx_wave.h);
        // the row in which it first reaches r is then scanned lane by lane below — same additions, same order
        inside = false;
        ListStream ls;
        stream_open(c, ls, pi, n);
        while (base < n) {
            nn = min(LANES, n - base);
            w = stream_next(c, ls, base, n);
            for (int row = 0; row < 4 && 16 * row < nn; ++row) {
                double before = carry;
                carry = row_chain(w, row, min(16, nn - 16 * row), carry);
                if (!(carry < r)) {
                    inside = true;
                    k0 = 16 * row;
                    nn = min(nn, 16 * row + 16);
                    carry = before;
                    break;
                }
            }
            if (inside) break;
            base += LANES;
        }
    }
    if (inside) {
        double pre = seq_scan(w, nn, carry, k0);
        unsigned long long hit = __ballot(lane >= k0 && lane < nn && !(pre < r));
        if (hit) {
            int j = __ffsll((long long)hit) - 1;
            double total = bcast(pre, j), wi = bcast(w, j);
            if (wi == 0.0) c.error = ERR_ZERO_WEIGHT;
            rn = (r - (total - wi)) / wi;
            return base + j;
        }
        carry = nn > 0 ? bcast(pre, nn - 1) : 0.0;  // register tile whose total stays below r
    }
    // nothing reached r: the dense loop runs on to index H-1 (fast_choose.pxi:26); that is a valid pick
    // only if haplotype H-1 is occupied, otherwise the reference reports a zero weight
    if (n > 0 && LH(c, pi)[n - 1] == c.H - 1) {
        double wi = c.tE[c.C == 1 ? 0 : lc[n - 1]] * (double)ln[n - 1];
        rn = (r - (carry - wi)) / wi;
        return n - 1;
    }
    c.error = ERR_ZERO_WEIGHT;
    return 0;
}

// totalRate and the prefix sums of popRate from population `from` onwards (pyx:537-539)
static __device__ __forceinline__ void refresh_cum(Ctx &c, int from) {
    const int P = c.P, lane = c.lane;
    if (c.fast) {
        double carry = 0.0;
        for (int base = 0; base < P; base += LANES) {
            int pn = base + lane;
            double pre = fscan(pn < P ? c.popRate[pn] : 0.0) + carry;
            if (pn < P) c.cum[pn] = pre;
            carry = bcast(pre, LANES - 1);
        }
        c.totalRate = carry;
        WSYNC();
        return;
    }
    int base0 = (from / LANES) * LANES;
    double carry = (base0 > 0) ? c.cum[base0 - 1] : 0.0;
    for (int base = base0; base < P; base += LANES) {
        int pn = base + lane;
        int k0 = (base == base0) ? ((from - base0) & ~7) : 0;  // first chain step actually executed
        double w = 0.0;
        if (pn < P && pn >= base + k0) w = c.popRate[pn];
        if (base == base0 && k0 > 0) carry = c.cum[base + k0 - 1];
        double pre = seq_scan(w, min(LANES, P - base), carry, k0);
        if (pn < P && pn >= base + k0) c.cum[pn] = pre;
        carry = bcast(pre, LANES - 1);
    }
    WSYNC();
    c.totalRate = c.cum[P - 1];
}

// migPopRate, its prefix sums and totalMigrationRate (pyx:541-546)
static __device__ __forceinline__ void refresh_mig(Ctx &c) {
    const int P = c.P, lane = c.lane;
    if (c.fast && c.has_mig) {
        double carry = 0.0;
        for (int base = 0; base < P; base += LANES) {
            int pn = base + lane;
            double w = 0.0;
            if (pn < P) {
                w = c.maxEBM[pn] * (double)c.totalSus[pn] * (double)(c.gI - c.totalInf[pn]);
                c.migRate[pn] = w;
            }
            double pre = fscan(w) + carry;
            if (pn < P) c.cumMig[pn] = pre;
            carry = bcast(pre, LANES - 1);
        }
        c.totalMig = carry;
        c.mig_cum_ok = true;
        WSYNC();
        return;
    }
    if (!c.has_mig) {  // every maxEffectiveBirthMigration is +0.0: all products and sums are +0.0
        for (int pn = lane; pn < P; pn += LANES) { c.migRate[pn] = 0.0; c.cumMig[pn] = 0.0; }
        c.totalMig = 0.0;
        c.mig_cum_ok = true;
        WSYNC();
        return;
    }
    // totalMigrationRate only: the prefix sums GenerateMigration's fastChoose needs (same additions, same order)
    // are produced by mig_scan() when a migration is actually drawn
    double acc = 0.0;
    for (int base = 0; base < P; base += LANES) {
        int pn = base + lane;
        double w = 0.0;
        if (pn < P) {
            w = c.maxEBM[pn] * (double)c.totalSus[pn] * (double)(c.gI - c.totalInf[pn]);
            c.migRate[pn] = w;
        }
        acc = seq_sum(w, min(LANES, P - base), acc);
    }
    c.totalMig = acc;
    c.mig_cum_ok = false;
    WSYNC();
}

// running totals of migPopRate (pyx:541-546 / fast_choose.pxi:25) for the population choice of a migration
static __device__ __forceinline__ void mig_scan(Ctx &c) {
    const int P = c.P, lane = c.lane;
    double carry = 0.0;
    for (int base = 0; base < P; base += LANES) {
        int pn = base + lane;
        double w = (pn < P) ? c.migRate[pn] : 0.0;
        double pre = seq_scan(w, min(LANES, P - base), carry);
        if (pn < P) c.cumMig[pn] = pre;
        carry = bcast(pre, LANES - 1);
    }
    c.mig_cum_ok = true;
    WSYNC();
}

// ------------------------------------------------------------------------------------------------
// Deferred work of one loop iteration.  Event handlers only change compartments and describe what has to
// be refreshed; the kernel body then runs the (large) list and rate routines from ONE call site each.
struct UpdReq {      // UpdateRates(pi, infect, immune, migration) (pyx:516) or, with full, UpdateAllRates (pyx:279)
    bool any, infect, immune, migration, full;
    int lo, hi;      // populations [lo, hi)
};
struct ListOp { int pi, hap; int64_t delta; };
struct EvRec { int type, hap, pop, nh, np; };  // type < 0: nothing to log (rejected migration)

// UpdateRates for populations [lo, hi) / UpdateAllRates.  suscepCumulTransition, the migration diagonal,
// actualSizes and maxEffectiveBirth depend on parameters only and come from the host (vgx_api.hip).
static __device__ __forceinline__ void update(Ctx &c, const UpdReq &q, const Tile &t) {
    const VgxDevParams &p = *c.p;
    const int P = c.P, S = c.S, lane = c.lane;
    if (c.fast && q.full) {  // kmig[pi] = sum_pn m[pi][pn]^2 * cd[pn] / as[pn]
        for (int pi = 0; pi < P; ++pi) {
            const double *mrow = p.mig + (int64_t)pi * P;
            double acc = 0.0;
            for (int base = 0; base < P; base += LANES) {
                int pn = base + lane;
                double tv = 0.0;
                if (pn < P) { double m = mrow[pn]; tv = m * m * c.cd[pn] / c.as[pn]; }
                acc += bcast(fscan(tv), LANES - 1);
            }
            if (lane == 0) c.kmig[pi] = acc;
        }
        WSYNC();
    }
    for (int pn = q.lo; pn < q.hi; ++pn) {
        if (q.infect) {
            birth_update(c, pn, t);
            PROF(8);
            tE_fill(c, pn);
            PROF(9);
            double v = row_sum(c, pn, t);
            if (lane == 0) c.infect[pn] = v;
            PROF(10);
        }
        if (q.full) {  // pyx:320-323
            for (int sn = lane; sn < S; sn += LANES) c.immSrc[pn * S + sn] = c.cumul[sn] * (double)c.sus[pn * S + sn];
            WSYNC();
        }
        if (q.immune) {
            double v = 0.0;
            for (int sn = 0; sn < S; ++sn) v += c.immSrc[pn * S + sn];
            if (lane == 0) c.immune[pn] = v;
        }
        WSYNC();
        if (lane == 0) c.popRate[pn] = c.infect[pn] + c.immune[pn];
    }
    WSYNC();
    PROF(11);
    if (q.infect || q.immune) refresh_cum(c, q.lo);
    PROF(12);
    if (q.full) {
        // effectiveMigration[pn1, pn2] (pyx:327-338): lane <-> target pn2, serial over sources and the inner sum
        bool any = false;
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
            any = any || (__ballot(pn2 < P && mx * p.maxEffectiveBirth != 0.0) != 0ull);
        }
        c.has_mig = any;
        WSYNC();
        PROF(13);
    }
    if (q.migration) refresh_mig(c);
    PROF(14);
}

static __device__ __forceinline__ void add_event(Ctx &c, const EvRec &e) {  // events.pxi:37-44
    if (c.record_events) {
        int64_t slot = c.ev_ptr - c.ev_base;
        if (slot >= 0 && slot < c.evcap) {
            int lane = c.lane;
            if (lane < VGX_EV_COLS) {
                int v = lane == 0 ? e.type : lane == 1 ? e.hap : lane == 2 ? e.pop : lane == 3 ? e.nh : lane == 4 ? e.np
                                                                                                             : (int)(uint32_t)c.iter_key;
                c.ev_cols[slot * VGX_EV_COLS + lane] = v;
            } else if (lane == VGX_EV_COLS) {
                c.ev_rate[slot] = c.den;
            }
        } else {
            c.error = ERR_CAPACITY;
        }
    }
    c.ev_ptr += 1;
}

// CheckLockdown (pyx:698-710) for populations [lo, hi): applies the switches and logs them; returns whether
// any happened.  UpdateAllRates is a pure function of the state, so running it once after the last switch
// (the caller does) leaves exactly what the reference's call after every switch leaves.
static __device__ __forceinline__ bool check_lockdowns(Ctx &c, int lo, int hi) {
    const VgxDevParams &p = *c.p;
    bool any = false;
    for (int pi = lo; pi < hi; ++pi) {
        for (int pass = 0; pass < 2; ++pass) {
            bool flip;
            // thresholds startLD * sizes / endLD * sizes (pyx:699, 705) from the read-only parameters (scalar loads)
            if (pass == 0) flip = ((double)c.totalInf[pi] > p.startLD[pi] * (double)p.sizes[pi]) && c.lockON[pi] == 0;
            else flip = ((double)c.totalInf[pi] < p.endLD[pi] * (double)p.sizes[pi]) && c.lockON[pi] == 1;
            if (flip) {
                WSYNC();
                if (c.lane == 0) {
                    c.cd[pi] = pass == 0 ? p.cdAfter[pi] : p.cdBefore[pi];
                    c.lockON[pi] = pass == 0 ? 1 : 0;
                    if (c.loc_n < c.loc_cap) {
                        c.loc_rec[c.loc_n * 2 + 0] = pass == 0 ? 1 : 0;
                        c.loc_rec[c.loc_n * 2 + 1] = pi;
                        c.loc_time[c.loc_n] = c.currentTime;
                        c.loc_iter[c.loc_n] = c.iter_key;
                    }
                }
                if (c.loc_n >= c.loc_cap) c.error = ERR_CAPACITY;
                BUMP(CNT_SWAP);
                c.loc_n += 1;
                any = true;
                WSYNC();
            }
        }
    }
    return any;
}

static __device__ __forceinline__ int mutate(const Ctx &c, int hi, int s, int DS) {  // pyx:2420-2427
    int digit4 = 1 << (2 * (c.sites - s - 1));
    int AS = (hi / digit4) % 4;
    if (DS >= AS) DS += 1;
    return hi + (DS - AS) * digit4;
}

// NewInfections on the population counters (pyx:246-251); the list is updated by the caller
static __device__ __forceinline__ void counters_infect(Ctx &c, int pi, int si) {
    if (c.lane == 0) {
        c.sus[pi * c.S + si] -= 1;
        c.totalSus[pi] -= 1;
        c.totalInf[pi] += 1;
    }
    c.gI += 1;
    WSYNC();
}

// summary trajectories: emit the state for every grid point passed by the time step (state before the event)
static __device__ __forceinline__ void traj_emit(Ctx &c, double t_new, bool final_fill) {
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
// One GenerateEvent (pyx:483-512): selects and applies the event, fills the deferred work.
// The second parent of a recombinant birth (pyx:575-585): fastChoose over eventHapPopRate[pi, h, 0] * (infectious[pi, h] minus
// the first parent's host) in haplotype order, on the occupancy list (empty haplotypes add +0.0 and cannot be the first index
// at which the running sum reaches r > 0).  Recombinant births are rare: plain chunks of 64 entries, exact mode only.
// Returns the haplotype; c.rn is rescaled as fast_choose.pxi:30 does.
static __device__ __forceinline__ int recomb_partner(Ctx &c, int pi, int hi) {
    const int lane = c.lane, n = c.nocc[pi];
    const int32_t *lh = LH(c, pi), *lc = LC(c, pi);
    const int64_t *ln = LN(c, pi);
    auto weight = [&](int k) -> double {
        if (k >= n) return 0.0;
        return c.birthC[pi * c.CB + c.c_bidx[lc[k]]] * (double)(ln[k] - (lh[k] == hi ? 1 : 0));
    };
    double hs = 0.0;
    for (int base = 0; base < n; base += LANES) hs = seq_sum(weight(base + lane), min(LANES, n - base), hs);
    const double r = hs * c.rn;
    if (!(0.0 < r) && !(n > 0 && lh[0] == 0)) { c.error = ERR_ZERO_WEIGHT; return 0; }   // the dense loop stops at haplotype 0
    double carry = 0.0;
    for (int base = 0; base < n; base += LANES) {
        const int nn = min(LANES, n - base);
        const double w = weight(base + lane);
        const double pre = seq_scan(w, nn, carry);
        const unsigned long long hit = __ballot(lane < nn && !(pre < r));
        if (hit) {
            const int j = __ffsll((long long)hit) - 1;
            const double total = bcast(pre, j), wi = bcast(w, j);
            if (wi == 0.0) { c.error = ERR_ZERO_WEIGHT; return 0; }
            c.rn = (r - (total - wi)) / wi;
            return lh[base + j];
        }
        carry = bcast(pre, nn - 1);
    }
    // rounding left the total below r: the dense loop ends at haplotype H - 1 (fast_choose.pxi:26)
    if (n > 0 && lh[n - 1] == c.H - 1) {
        const double wi = c.birthC[pi * c.CB + c.c_bidx[lc[n - 1]]] * (double)(ln[n - 1] - (lh[n - 1] == hi ? 1 : 0));
        if (wi == 0.0) { c.error = ERR_ZERO_WEIGHT; return 0; }
        c.rn = (r - (carry - wi)) / wi;
        return c.H - 1;
    }
    c.error = ERR_ZERO_WEIGHT;
    return 0;
}

static __device__ __forceinline__ int generate_event(Ctx &c, double u, Tile &t, UpdReq &q, ListOp &op0, ListOp &op1,
                                                     int &n_ops, EvRec &ev) {
    const VgxDevParams &p = *c.p;
    const int P = c.P, S = c.S, lane = c.lane;
    int pi;
    c.rn = u;
    double choose = c.rn * (c.totalRate + c.totalMig);
    if (c.totalRate > choose) {
        c.rn = choose / c.totalRate;
        pi = choose_prefix(c, c.popRate, c.cum, P, c.totalRate, c.rn);
        choose = c.rn * c.popRate[pi];
        q.lo = pi; q.hi = pi + 1; q.any = true;
        PROF(2);
        if (c.immune[pi] > choose) {
            // ---- ImmunityTransition (pyx:550-564) ----
            c.rn = choose / c.immune[pi];
            int ssi = choose_serial(c, [&](int i) { return c.immSrc[pi * S + i]; }, S, c.immune[pi], c.rn);
            int tsi = choose_serial(c, [&](int i) { return c.trans[ssi * S + i]; }, S, c.cumul[ssi], c.rn);
            WSYNC();
            if (lane == 0) {
                c.sus[pi * S + ssi] -= 1;
                c.sus[pi * S + tsi] += 1;
                c.immSrc[pi * S + ssi] = (double)c.sus[pi * S + ssi] * c.cumul[ssi];
                c.immSrc[pi * S + tsi] = (double)c.sus[pi * S + tsi] * c.cumul[tsi];
            }
            WSYNC();
            q.immune = true;
            BUMP(CNT_I);
            ev.type = EV_SUSCCHANGE; ev.hap = ssi; ev.pop = pi; ev.nh = tsi; ev.np = 0;
        } else {
            c.rn = (choose - c.immune[pi]) / c.infect[pi];
            tile_load(c, t, pi);
            tE_fill(c, pi);  // from the cached per-class birth rates: the same values UpdateRates stored
            PROF(3);
            int k = row_select(c, pi, t, c.infect[pi], c.rn);
            PROF(4);
            if (c.error) return pi;
            int hi, cls;
            if (t.valid) { hi = __builtin_amdgcn_readlane(t.hap, k); cls = __builtin_amdgcn_readlane(t.cls, k); }
            else { hi = LH(c, pi)[k]; cls = LC(c, pi)[k]; }
            int cb = c.c_bidx[cls];
            double e0 = c.birthC[pi * c.CB + cb], e1 = c.c_d[cls], e2 = c.c_s[cls] * c.sampMult[pi], e3 = c.c_tm[cls];
            double tEv = c.tE[cls];
            int ei = choose_serial(c, [&](int i) { return i == 0 ? e0 : i == 1 ? e1 : i == 2 ? e2 : e3; }, 4, tEv, c.rn);
            q.infect = true;
            if (ei == 0) {
                // ---- Birth (pyx:568-605, recombination off) ----
                const double *x = c.xC + (pi * c.CB + cb) * S;
                double ws = 0.0;
                for (int sn = 0; sn < S; ++sn) ws += x[sn];
                int si = choose_serial(c, [&](int i) { return x[i]; }, S, ws, c.rn);
                if (__builtin_expect(p.recombination != 0.0 && c.rn < p.recombination && c.totalInf[pi] > 1, 0)) {
                    // ---- recombinant birth (pyx:575-596): the new host carries nhi, the event names both parents ----
                    c.rn = c.rn / p.recombination;
                    const int hi2 = recomb_partner(c, pi, hi);
                    if (c.error) return pi;
                    const int64_t posRecomb = (int64_t)((double)p.genome_length * c.rn);
                    // pyx:586-591 as written: the recombinant carries only the last site of one parent
                    int nhi = 0;
                    if (c.sites > 0) nhi = (p.sitesPosition[c.sites - 1] < posRecomb ? hi : hi2) % 4;
                    if (c.rec && lane == 0) {
                        if (c.rec_n < c.rec_cap) {
                            int64_t *o = c.rec + c.rec_n * 5;
                            o[0] = c.ev_ptr; o[1] = hi; o[2] = hi2; o[3] = nhi; o[4] = posRecomb;
                        }
                    }
                    if (c.rec && c.rec_n >= c.rec_cap) { c.error = ERR_CAPACITY; return pi; }
                    c.rec_n += 1;
                    counters_infect(c, pi, si);
                    if (lane == 0) c.immSrc[pi * S + si] = c.cumul[si] * (double)c.sus[pi * S + si];
                    WSYNC();
                    t.valid = false;
                    op0.pi = pi; op0.hap = nhi; op0.delta = +1; n_ops = 1;
                    q.immune = true; q.migration = true;
                    BUMP(CNT_B);
                    ev.type = EV_BIRTH; ev.hap = hi; ev.pop = pi; ev.nh = si; ev.np = hi2;
                    return pi;
                }
                counters_infect(c, pi, si);
                if (t.valid) {
                    if (lane == k) { t.cnt += 1; LN(c, pi)[k] = t.cnt; }
                } else if (lane == 0) {
                    LN(c, pi)[k] += 1;
                }
                if (lane == 0) { if (!t.valid) LT(c, pi)[k / LANES] += 1; c.immSrc[pi * S + si] = c.cumul[si] * (double)c.sus[pi * S + si]; }
                WSYNC();
                q.immune = true; q.migration = true;
                BUMP(CNT_B);
                ev.type = EV_BIRTH; ev.hap = hi; ev.pop = pi; ev.nh = si; ev.np = c.H;
            } else if (ei == 1 || ei == 2) {
                // ---- Death / Sampling (pyx:616-635) ----
                int st = c.c_stype[cls];
                if (lane == 0) {
                    c.sus[pi * S + st] += 1;
                    c.totalSus[pi] += 1;
                    c.totalInf[pi] -= 1;
                }
                c.gI -= 1;
                int64_t left = (t.valid ? bcast_i64(t.cnt, k) : LN(c, pi)[k]) - 1;
                if (left == 0) {
                    op0.pi = pi; op0.hap = hi; op0.delta = -1; n_ops = 1;
                    t.valid = false;
                } else {
                    if (t.valid) {
                        if (lane == k) { t.cnt = left; LN(c, pi)[k] = left; }
                    } else if (lane == 0) {
                        LN(c, pi)[k] = left;
                    }
                    if (lane == 0 && !t.valid) LT(c, pi)[k / LANES] -= 1;
                }
                WSYNC();
                if (lane == 0) c.immSrc[pi * S + st] = (double)c.sus[pi * S + st] * c.cumul[st];
                WSYNC();
                q.immune = true; q.migration = true;
                if (ei == 2) { BUMP(CNT_S); ev.type = EV_SAMPLING; } else { BUMP(CNT_D); ev.type = EV_DEATH; }
                ev.hap = hi; ev.pop = pi; ev.nh = st; ev.np = 0;
            } else {
                // ---- Mutation (pyx:640-667) ----
                const int sites = c.sites;
                const double *mr = p.mRate + (int64_t)hi * sites;
                int mi = choose_serial(c, [&](int i) { return mr[i]; }, sites, c.c_tm[cls], c.rn);
                const double *hm = p.hapMutType + ((int64_t)hi * sites + mi) * 3;
                int DS = choose_serial(c, [&](int i) { return hm[i]; }, 3, hm[0] + hm[1] + hm[2], c.rn);
                int nhi = mutate(c, hi, mi, DS);
                t.valid = false;
                op0.pi = pi; op0.hap = nhi; op0.delta = +1;
                op1.pi = pi; op1.hap = hi; op1.delta = -1;
                n_ops = 2;
                BUMP(CNT_M);
                ev.type = EV_MUTATION; ev.hap = hi; ev.pop = pi; ev.nh = nhi; ev.np = 0;
            }
        }
    } else {
        // ---- GenerateMigration (pyx:672-694) ----
        c.rn = (choose - c.totalRate) / c.totalMig;
        if (!c.mig_cum_ok) mig_scan(c);
        int tpi = choose_prefix(c, c.migRate, c.cumMig, P, c.totalMig, c.rn);
        pi = tpi;
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
            int64_t wi = c.totalInf[spi];
            if (wi == 0) c.error = ERR_ZERO_WEIGHT;
            c.rn = (r - (double)(total - wi)) / (double)wi;
        }
        // fastChoose(infectious[spi], totalInfectious[spi], rn): int64 weights over the occupancy list
        int hi = 0;
        {
            int kk = count_select(c, spi, t, c.totalInf[spi], c.rn);  // t is not valid here (no tile loaded yet)
            if (!c.error) hi = LH(c, spi)[kk];
        }
        if (c.error) return pi;
        int si = choose_serial_i64(c, [&](int i) { return c.sus[tpi * S + i]; }, S, c.totalSus[tpi], c.rn);
        double p_accept = c.effMig[(int64_t)spi * P + tpi] * p.bRate[hi] * p.susc[(int64_t)hi * S + si] / c.maxEBM[tpi];
        if (c.rn < p_accept) {
            counters_infect(c, tpi, si);
            op0.pi = tpi; op0.hap = hi; op0.delta = +1; n_ops = 1;
            q.any = true; q.lo = tpi; q.hi = tpi + 1; q.infect = true; q.immune = true; q.migration = true;
            BUMP(CNT_MIGP);
            ev.type = EV_MIGRATION; ev.hap = hi; ev.pop = spi; ev.nh = si; ev.np = tpi;
        } else {
            BUMP(CNT_MIGN);
        }
    }
    PROF(5);
    return pi;
}

// Restart (pyx:714-738): compartments back to the initial snapshot; the caller re-checks lockdowns and
// rebuilds all rates.
static __device__ __forceinline__ void restart_state(Ctx &c, const VgxDevRep &r) {
    const int P = c.P, S = c.S, lane = c.lane;
    c.ev_ptr = 0;
    if (lane < 8 && lane != CNT_SWAP) c.cnt[lane] = 0;  // swapLockdown survives a Restart (pyx:714-738)
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
        const int n_old = c.nocc[pn];
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
            int64_t tsum = bcast_i64(iscan(ct, lane), LANES - 1);
            if (lane == 0) LT(c, pn)[base / LANES] = tsum;
            ti += tsum;
        }
        for (int j = (n + LANES - 1) / LANES + lane; j <= n_old / LANES && j < c.capT; j += LANES) LT(c, pn)[j] = 0;
        WSYNC();
        if (lane == 0) { c.nocc[pn] = n; c.totalSus[pn] = ts; c.totalInf[pn] = ti; }
        g += ti;
    }
    c.gI = g;
    WSYNC();
}

// Kernel body, specialised on the LDS stride of the per-population arrays (PT: 64 when popNum <= 64, so every
// LDS address is a compile-time constant; 0 = runtime stride), on the number of susceptibility groups (ST: 1 or 0 =
// runtime) and on a single rate class (ONE): the common shapes lose their address arithmetic and inner loops.
template <int PT, int ST, int ONE, int FAST>
static __device__ __forceinline__ void direct_body(const VgxDirectArgs &a) {
    const int rep = blockIdx.x;
    if (rep >= a.n_replicates) return;
    const int lane = threadIdx.x;
    const VgxDevParams &p = a.p;
    const VgxDevRep &r = a.r;
    const int P = p.P, S = ST ? ST : p.S, C = ONE ? 1 : p.C, CB = ONE ? 1 : p.CB;
    const int PL = PT ? PT : P;   // LDS stride of the [P] arrays

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx c;
    c.P = P; c.S = S; c.H = p.H; c.C = C; c.CB = CB; c.sites = p.sites; c.lane = lane;
    c.fast = FAST;
    c.p = &a.p;
    // LDS carve: keep in step with vgxi_direct_lds_bytes().  Arrays whose size depends only on (PL, S) first.
    double *ld = (double *)smem;
    c.popRate = ld; ld += PL;   c.infect = ld; ld += PL;   c.immune = ld; ld += PL;   c.migRate = ld; ld += PL;
    c.maxEBM = ld; ld += PL;    c.cd = ld; ld += PL;       c.as = ld; ld += PL;       c.cum = ld; ld += PL;
    c.cumMig = ld; ld += PL;    c.sampMult = ld; ld += PL;
    c.kmig = ld; ld += PL;
    c.immSrc = ld; ld += PL * S;
    c.cumul = ld; ld += S;
    c.trans = ld; ld += S * S;
    int64_t *li = (int64_t *)ld;
    c.totalSus = li; li += PL;   c.totalInf = li; li += PL;   c.lockON = li; li += PL;
    c.sus = li; li += PL * S;
    c.cnt = li; li += 8;
    c.nocc = (int32_t *)li; li += (PL + 1) / 2;
    ld = (double *)li;
    c.birthC = ld; ld += PL * CB;
    c.xC = ld; ld += PL * CB * S;
    c.tE = ld; ld += C;   c.c_d = ld; ld += C;   c.c_s = ld; ld += C;   c.c_tm = ld; ld += C;
    c.cb_b = ld; ld += CB;
    c.cb_sigma = ld; ld += CB * S;
    int32_t *l4 = (int32_t *)ld;
    c.c_bidx = l4; l4 += C;
    c.c_stype = l4; l4 += C;

    double *gD = r.popD + (int64_t)rep * PD_COUNT * P;
    int64_t *gI = r.popI + (int64_t)rep * PI_COUNT * P;
    int32_t *gN = r.nocc + (int64_t)rep * P;
    for (int pn = lane; pn < P; pn += LANES) {
        c.popRate[pn] = 0.0; c.infect[pn] = 0.0; c.immune[pn] = 0.0; c.migRate[pn] = 0.0; c.maxEBM[pn] = 0.0;
        c.cum[pn] = 0.0; c.cumMig[pn] = 0.0; c.kmig[pn] = 0.0;
        c.cd[pn] = gD[PD_CD * P + pn];
        c.as[pn] = p.actualSizes[pn];
        c.sampMult[pn] = p.sampMult[pn];
        c.totalSus[pn] = gI[PI_TOTSUS * P + pn];
        c.totalInf[pn] = gI[PI_TOTINF * P + pn];
        c.lockON[pn] = gI[PI_LOCK * P + pn];
        c.nocc[pn] = gN[pn];
    }
    for (int i = lane; i < P * S; i += LANES) { c.sus[i] = r.sus[(int64_t)rep * P * S + i]; c.immSrc[i] = 0.0; }
    for (int i = lane; i < P * CB; i += LANES) c.birthC[i] = 0.0;
    for (int i = lane; i < P * CB * S; i += LANES) c.xC[i] = 0.0;
    for (int i = lane; i < C; i += LANES) {
        c.tE[i] = 0.0; c.c_d[i] = p.c_d[i]; c.c_s[i] = p.c_s[i]; c.c_tm[i] = p.c_tm[i];
        c.c_bidx[i] = p.c_bidx[i]; c.c_stype[i] = p.c_stype[i];
    }
    for (int i = lane; i < CB; i += LANES) c.cb_b[i] = p.cb_b[i];
    for (int i = lane; i < CB * S; i += LANES) c.cb_sigma[i] = p.cb_sigma[i];
    for (int i = lane; i < S; i += LANES) c.cumul[i] = p.suscepCumul[i];
    for (int i = lane; i < S * S; i += LANES) c.trans[i] = p.suscepTransition[i];

    c.effMig = r.effMig + (int64_t)rep * P * P;
    c.cap = r.cap;
    c.lhap = r.lhap + (int64_t)rep * P * r.cap;
    c.lcls = r.lcls + (int64_t)rep * P * r.cap;
    c.lcnt = r.lcnt + (int64_t)rep * P * r.cap;
    c.capT = r.capT;
    c.ltsum = r.ltsum + (int64_t)rep * P * r.capT;
    c.evcap = r.evcap; c.ev_base = r.ev_base;
    c.ev_rate = r.ev_rate + (int64_t)rep * r.evcap;
    c.ev_cols = r.ev_cols + (int64_t)rep * r.evcap * VGX_EV_COLS;
    c.loc_cap = r.loc_cap;
    c.loc_rec = r.loc_rec + (int64_t)rep * r.loc_cap * 2;
    c.loc_time = r.loc_time + (int64_t)rep * r.loc_cap;
    c.loc_iter = r.loc_iter + (int64_t)rep * r.loc_cap;
    c.den = 0.0; c.iter_key = 0;
    c.traj_points = r.traj_points; c.traj_t0 = r.traj_t0; c.traj_dt = r.traj_dt;
    c.traj = r.traj ? r.traj + (int64_t)rep * r.traj_points * P * 2 : nullptr;
    c.record_events = a.record_events;
    VgxRepScalars *sc = r.sc + rep;
    c.currentTime = sc->currentTime; c.totalRate = 0.0; c.totalMig = 0.0; c.rn = 0.0;
    c.gI = sc->globalInfectious;
    if (lane == 0) {
        c.cnt[CNT_B] = sc->bCounter; c.cnt[CNT_D] = sc->dCounter; c.cnt[CNT_S] = sc->sCounter; c.cnt[CNT_M] = sc->mCounter;
        c.cnt[CNT_I] = sc->iCounter; c.cnt[CNT_SWAP] = sc->swapLockdown; c.cnt[CNT_MIGP] = sc->migPlus;
        c.cnt[CNT_MIGN] = sc->migNonPlus;
    }
    c.ev_ptr = sc->ev_ptr; c.ev_size = a.ev_size; c.loc_n = 0; c.error = 0;
    c.traj_next = 0;
    c.rec = r.rec ? r.rec + rep * r.rec_cap * 5 : nullptr; c.rec_cap = r.rec_cap; c.rec_n = 0;
    c.has_mig = true;
    c.mig_cum_ok = false;
    WSYNC();
    {   // a population can switch on only if its threshold lies below its size (S + I of a population is conserved,
        // so totalInfectious[pn] <= sizes[pn]) and off only if it is on
        bool possible = false;
        for (int base = 0; base < P; base += LANES) {
            int pn = base + lane;
            bool pp = pn < P && (p.startLD[pn] * (double)p.sizes[pn] < (double)p.sizes[pn] || c.lockON[pn] != 0);
            possible = possible || (__ballot(pp) != 0ull);
        }
        c.ld_any = possible;
    }

    const double tlimit = (double)a.time;
    const bool has_tlimit = !(a.time == -1.0f);
    const int64_t seed = r.seeds[rep];
    int64_t loops = 0, restarts = 0, good_attempt = sc->good_attempt;
    int64_t last_att = -1, last_att_loops = 0;
    int64_t att = 0;
    int64_t att_ev0 = sc->ev_ptr, att_loc0 = 0, fa_n = 0;   // first log index / lockdown record of the current attempt
    RngBatch g;
    rng_init_lane(g, lane);
    g.philox = a.rng_philox;
    g.sh = g.sl = g.ih = g.il = 0;
    g.val = 0.0;

    // The reference's control flow (pyx:399-418) as one loop with one call site per large routine:
    //   rebuild : [CheckLockdown for all populations] + UpdateAllRates   (PrepareParameters tail, Restart tail)
    //   event   : SampleTime, GenerateEvent -> list ops -> UpdateRates -> extinction test -> CheckLockdown(pi)
    //             -> UpdateAllRates if a lockdown switched
    bool rebuild = true;       // start of the call / after a Restart
    bool attempt_open = false; // RNG seeded and first-iteration guard (pyx:404) passed for attempt `att`
    bool finished = false;
#ifdef VGX_PROFILE
    for (int i = 0; i < VGX_PROF_SLOTS; ++i) c.prof_acc[i] = 0;
    c.prof_t0 = __builtin_readcyclecounter();
#endif
    while (!finished && !c.error) {
        PROF(0);
        Tile t;
        t.valid = false; t.pi = -1; t.n = 0; t.hap = 0; t.cls = 0; t.cnt = 0; t.mrow0 = 0.0;
        UpdReq q;
        q.any = false; q.infect = false; q.immune = false; q.migration = false; q.full = false; q.lo = 0; q.hi = 0;
        ListOp op0, op1;
        op0.pi = 0; op0.hap = 0; op0.delta = 0; op1 = op0;
        int n_ops = 0;
        EvRec ev;
        ev.type = -1; ev.hap = 0; ev.pop = 0; ev.nh = 0; ev.np = 0;
        int lk_lo = 0, lk_hi = 0;
        bool end_attempt = false;

        if (rebuild) {
            lk_lo = 0; lk_hi = P;
            c.iter_key = att << 40;   // CheckLockdown of PrepareParameters / Restart: before the attempt's first iteration
        } else {
            if (!attempt_open) {
                if (att >= a.attempts) { finished = true; continue; }
                rng_seed(g, (uint64_t)seed, (uint32_t)att);
                attempt_open = true;
                last_att = att; last_att_loops = 0;
                if (!(c.totalRate + c.totalMig != 0.0 && c.gI != 0)) end_attempt = true;  // pyx:404
            }
            if (!end_attempt && !(c.ev_ptr < c.ev_size && (a.sample_size == -1 || c.cnt[CNT_S] <= a.sample_size) &&
                                  (!has_tlimit || c.currentTime < tlimit)))
                end_attempt = true;  // pyx:405-407
            if (!end_attempt) {
                if (loops >= a.max_loop) { c.error = ERR_LOOP_GUARD; continue; }
                loops += 1;
                last_att_loops += 1;
                if (g.pos == 32) rng_refill(g, lane);
                double nlog = bcast(g.val, 2 * g.pos), u2 = bcast(g.val, 2 * g.pos + 1);
                g.pos += 1;
                c.den = c.totalRate + c.totalMig;
                c.iter_key = (att << 40) | last_att_loops;
                double t_new = c.currentTime + (nlog / c.den);  // SampleTime pyx:476-478
                if (c.traj) traj_emit(c, t_new, false);
                c.currentTime = t_new;
                PROF(1);
                int pi = generate_event(c, u2, t, q, op0, op1, n_ops, ev);
                if (c.error) continue;
                lk_lo = pi; lk_hi = pi + 1;
            }
        }

        if (!end_attempt) {
            for (int i = 0; i < n_ops; ++i) {  // the one call site of the list routines
                list_add(c, i == 0 ? op0.pi : op1.pi, i == 0 ? op0.hap : op1.hap, i == 0 ? op0.delta : op1.delta);
                if (c.error) break;
            }
            if (c.error) continue;
            PROF(6);
            if (ev.type >= 0) add_event(c, ev);
            PROF(7);
            for (int round = 0; round < 2; ++round) {  // the one call site of the rate routines
                if (round == 1) {
                    if (!rebuild && (c.totalRate == 0.0 || c.gI == 0)) { end_attempt = true; break; }  // pyx:410-411
                    bool switched = c.ld_any && check_lockdowns(c, lk_lo, lk_hi);  // pyx:412 / pyx:449-450 / pyx:736-737
                    if (!switched && !rebuild) break;
                    q.any = true; q.full = true; q.infect = true; q.immune = true; q.migration = true;
                    q.lo = 0; q.hi = P;
                    t.valid = false;
                }
                if (q.any) update(c, q, t);
            }
            PROF(15);
            rebuild = false;
            if (!end_attempt) continue;
        }

        // end of an attempt (pyx:414-418)
        attempt_open = false;
        if (c.ev_ptr <= 100 && a.iterations > 100) {
            // the lockdown records of this failed attempt stay in the log (pyx:714-738): keep what the host clock needs
            if (c.loc_n > att_loc0 && c.record_events && r.fa_cap > 0) {
                const int64_t n = c.ev_ptr - att_ev0;
                for (int64_t k = lane; k < n; k += LANES) {
                    const int64_t slot = att_ev0 + k - c.ev_base;
                    if (fa_n + k < r.fa_cap && slot >= 0 && slot < c.evcap) {
                        r.fa_rate[(int64_t)rep * r.fa_cap + fa_n + k] = c.ev_rate[slot];
                        r.fa_key[(int64_t)rep * r.fa_cap + fa_n + k] = (att << 40) | (int64_t)(uint32_t)c.ev_cols[slot * VGX_EV_COLS + 5];
                    }
                }
                fa_n += n;
                WSYNC();
            }
            restart_state(c, r);
            att_ev0 = 0;
            att_loc0 = c.loc_n;
            restarts += 1;
            att += 1;
            rebuild = true;  // CheckLockdown for all + UpdateAllRates, also after the last attempt
        } else {
            good_attempt = att + 1;
            finished = true;
        }
    }
    if (c.traj) traj_emit(c, 0.0, true);

    WSYNC();
    for (int pn = lane; pn < P; pn += LANES) {
        gD[PD_POPRATE * P + pn] = c.popRate[pn];
        gD[PD_INFECT * P + pn] = c.infect[pn];
        gD[PD_IMMUNE * P + pn] = c.immune[pn];
        gD[PD_MIG * P + pn] = c.migRate[pn];
        gD[PD_MAXEBM * P + pn] = c.maxEBM[pn];
        gD[PD_CD * P + pn] = c.cd[pn];
        gI[PI_TOTSUS * P + pn] = c.totalSus[pn];
        gI[PI_TOTINF * P + pn] = c.totalInf[pn];
        gI[PI_LOCK * P + pn] = c.lockON[pn];
        gN[pn] = c.nocc[pn];
    }
    for (int i = lane; i < P * S; i += LANES) {
        r.sus[(int64_t)rep * P * S + i] = c.sus[i];
        r.immSrc[(int64_t)rep * P * S + i] = c.immSrc[i];
    }
#ifdef VGX_PROFILE
    if (lane == 0 && r.prof)
        for (int i = 0; i < VGX_PROF_SLOTS; ++i) r.prof[(int64_t)rep * VGX_PROF_SLOTS + i] = c.prof_acc[i];
#endif
    if (lane == 0) {
        sc->currentTime = c.currentTime; sc->totalRate = c.totalRate; sc->totalMig = c.totalMig;
        sc->globalInfectious = c.gI;
        sc->bCounter = c.cnt[CNT_B]; sc->dCounter = c.cnt[CNT_D]; sc->sCounter = c.cnt[CNT_S]; sc->mCounter = c.cnt[CNT_M];
        sc->iCounter = c.cnt[CNT_I]; sc->swapLockdown = c.cnt[CNT_SWAP]; sc->migPlus = c.cnt[CNT_MIGP];
        sc->migNonPlus = c.cnt[CNT_MIGN];
        sc->good_attempt = good_attempt;
        sc->ev_ptr = c.ev_ptr; sc->loop_iterations = loops; sc->restarts = restarts;
        sc->loc_n = c.loc_n; sc->error = c.error; sc->traj_next = c.traj_next;
        sc->last_attempt = last_att; sc->last_attempt_loops = last_att_loops;
        sc->fa_n = fa_n;
        sc->rec_n = c.rec_n;
    }
}

extern "C" __global__ void __launch_bounds__(LANES) vgx_direct_kernel(VgxDirectArgs a) { direct_body<0, 0, 0, 0>(a); }
// popNum <= 64: constant LDS addresses
extern "C" __global__ void __launch_bounds__(LANES) vgx_direct_kernel_p64(VgxDirectArgs a) { direct_body<64, 0, 0, 0>(a); }
// popNum <= 64, one susceptibility group, one rate class (e.g. BASELINE configs 2 and 3)
extern "C" __global__ void __launch_bounds__(LANES) vgx_direct_kernel_p64s1c1(VgxDirectArgs a) { direct_body<64, 1, 1, 0>(a); }
// FAST mode (vgx_run_opts.mode = 1): order-free sums, same random stream and event semantics
extern "C" __global__ void __launch_bounds__(LANES) vgx_direct_fast_kernel(VgxDirectArgs a) { direct_body<0, 0, 0, 1>(a); }
extern "C" __global__ void __launch_bounds__(LANES) vgx_direct_fast_kernel_p64s1(VgxDirectArgs a) { direct_body<64, 1, 1, 1>(a); }

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
        int32_t *l3 = r.lcnt32 ? r.lcnt32 + (rep * P + pn) * r.cap : nullptr;
        int64_t *lt = r.ltsum + (rep * P + pn) * r.capT;   // zero-filled by the host beyond the list
        for (int base = 0; base < n; base += LANES) {
            int k = base + lane;
            int64_t ct = 0;
            if (k < n) {
                ct = s_cnt[(int64_t)pn * s_cap + k];
                lh[k] = s_hap[(int64_t)pn * s_cap + k];
                lc[k] = s_cls[(int64_t)pn * s_cap + k];
                ln[k] = ct;
                if (l3) l3[k] = (int32_t)ct;
            }
            int64_t tsum = bcast_i64(iscan(ct, lane), LANES - 1);
            if (lane == 0) lt[base / LANES] = tsum;
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

// ---- host-side launchers (this translation unit owns its kernels; no relocatable device code needed) ----
extern "C" __attribute__((visibility("hidden"))) size_t vgxi_direct_lds_bytes(int P, int S, int C, int CB) {
    size_t PL = P <= 64 ? 64 : (size_t)P;
    size_t f64 = 11 * PL + PL * S + S + (size_t)S * S;
    size_t i64 = 3 * PL + PL * S + 8 + (PL + 1) / 2;
    size_t f64b = PL * CB + PL * CB * S + 4 * (size_t)C + CB + (size_t)CB * S;
    size_t i32 = 2 * (size_t)C;
    return (f64 + i64 + f64b) * 8 + ((i32 * 4 + 15) / 16) * 16;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_direct(const VgxDirectArgs *a, size_t lds,
                                                                              hipStream_t stream) {
    void (*k)(VgxDirectArgs) = vgx_direct_kernel;
    if (a->fast) {
        k = (a->p.P <= 64 && a->p.S == 1 && a->p.C == 1 && a->p.CB == 1) ? vgx_direct_fast_kernel_p64s1 : vgx_direct_fast_kernel;
    } else if (a->p.P <= 64) {
        k = (a->p.S == 1 && a->p.C == 1 && a->p.CB == 1) ? vgx_direct_kernel_p64s1c1 : vgx_direct_kernel_p64;
    }
    hipError_t err = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k, dim3((unsigned)a->n_replicates), dim3(LANES), lds, stream, *a);
    return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_init_reps(
    const VgxDevRep *r, int P, int S, int64_t R, const int32_t *s_nocc, const int32_t *s_hap, const int32_t *s_cls,
    const int64_t *s_cnt, int64_t s_cap, const int64_t *s_sus, const double *s_cd, const int64_t *s_tot,
    hipStream_t stream) {
    hipLaunchKernelGGL(vgx_init_reps_kernel, dim3((unsigned)R), dim3(LANES), 0, stream, *r, P, S, R, s_nocc, s_hap,
                       s_cls, s_cnt, s_cap, s_sus, s_cd, s_tot);
    return hipGetLastError();
}
