// vgx_quadf.hip — FAST mode of the direct path on the row-per-replicate layout (four replicates per wavefront, one per 16-lane
// DPP row): the same event semantics and the same PCG64 stream as vgx_quad.hip (vgx_run_opts.mode = 1; mode = 2: the counter-based
// Philox stream), ORDER-FREE sums.  With one rate class the infection rate of a population is tEvent x totalInfectious — no pass over the occupancy list —
// and the haplotype is chosen by an integer prefix search over the counts (fast_choose.pxi:18-31 on int64 weights, tile sums for
// lists longer than a tile); BirthRate (pyx:382-392) is factored through sum_pn m^2 cd / as, a constant of the model where no
// lockdown can switch; the prefix sums of popRate over the populations (what fastChoose accumulates, pyx:537-539) are formed
// by a tree (lane l holds populations 4l .. 4l+3: three local additions and a 4-step row scan) after every update and kept in
// registers for the next event's choice; totalMigrationRate likewise.  An iteration is about a fifth of the exact kernel's
// instructions.  Same scope as vgx_quad.hip (popNum <= 64, one susceptibility group, one rate class, no possible lockdown switch,
// no recombination).  On the same seed the integer columns of the log, the counters and the compartments equal the exact
// mode's (rates differ at the 1e-16 level), times agree within 1e-9: tests/test_hip_quadf.py.
// The lists may hold zero-count entries while the kernel runs (vgx_rowlist.h); vgx_lists_settle_kernel squeezes them out afterwards.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_rowprim.h"
#include "vgx_rowlist.h"

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
// LDS (bytes), one wavefront per workgroup: model constants [64] f64: kmig (sum_pn m^2 cd / as), smul, maxEBM, spare   2048;
// PCG64 jump-ahead 512; per replicate infect[64], birthC[64] f64, totS[64], totI[64] i64, nocc[64] i32, counters       4 x 2400
#define Q_CONST_BYTES 2048
#define Q_RNG_BYTES 512
#define Q_REP_BYTES 2400
#define Q_LDS_BYTES (Q_CONST_BYTES + Q_RNG_BYTES + 4 * Q_REP_BYTES)
enum { QC_B = 0, QC_D, QC_M, QC_MIGP, QC_MIGN };

struct QFArgs {          // what vgx_quad_prep_kernel leaves for all replicates
    const double *effMig;   // [P][P]
    const double *maxEBM;   // [P]
    const int32_t *has_mig; // [1]
};

// inclusive f64 prefix inside each row, tree order (lanes without a source receive 0)
static __device__ __forceinline__ double row_fscan(double v) {
    VGX_SCAN_STEPS(VGX_F64_STEP)
    return v;
}
// the value of the previous lane of the row (0 for the row's first lane)
static __device__ __forceinline__ double row_prev_f64(double v) {
    const int lo = VGX_DPP_SHR(__double2loint(v), 1), hi = VGX_DPP_SHR(__double2hiint(v), 1);
    return __hiloint2double(hi, lo);
}

struct QCount { int k, hap; int64_t total, wi, tsum; bool none, tk; };   // tk: tsum = the sum of the hit's tile
// fastChoose over int64 weights = the counts of an occupancy list (fast_choose.pxi:18-31): first entry whose running count reaches
// rr.  Integer prefix sums are order-free: the tile sums pick the 64-entry tile (lane l looks at four consecutive tiles: 64 tiles
// per step), then lane l looks at entries 4l .. 4l+3 of the tile's 4-byte counts (one 16-byte load, the haplotypes with it): one
// dependent memory step for lists of up to a tile, two beyond.  `on`: rows that make the choice.  k = -1: nothing reached rr
// (total = the list's sum).
static __device__ __forceinline__ QCount q_count_select(const int32_t *lh2, const int32_t *l32, const int64_t *lt2, int n, double rr_, bool on) {
    const int rl = threadIdx.x & 15;
    QCount o;
    int64_t before = 0, tsum = 0;
    int base = 0;
    bool none = false, tk = false;
    const int maxn2 = rows_max(n);
    if (__builtin_expect(maxn2 > 64, 0)) {
        const int nt = n > 64 ? (n + 63) >> 6 : 0;     // tile sums exist only for lists longer than a tile
        const int maxt = rows_max(nt);
        int jt = -1;
        int64_t carry = 0;
        for (int tb = 0; tb < maxt; tb += 64) {
            const int j0 = tb + 4 * rl;
            int64_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = lt2[min(j0 + k, max(nt - 1, 0))];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j0 + k >= nt) w[k] = 0;
            const int64_t q0 = w[0], q1 = q0 + w[1], q2 = q1 + w[2], q3 = q2 + w[3];
            const int64_t incl = row_iscan(q3), ex = incl - q3 + carry;
            int kk = 4;
            if (j0 + 3 < nt && !((double)(ex + q3) < rr_)) kk = 3;
            if (j0 + 2 < nt && !((double)(ex + q2) < rr_)) kk = 2;
            if (j0 + 1 < nt && !((double)(ex + q1) < rr_)) kk = 1;
            if (j0 + 0 < nt && !((double)(ex + q0) < rr_)) kk = 0;
            const int q = row_min(jt < 0 && kk < 4 ? 4 * rl + kk : 64);
            if (jt < 0 && q < 64) {
                jt = tb + q;
                const int k = q & 3;
                const int64_t pk = ex + (k == 0 ? q0 : k == 1 ? q1 : k == 2 ? q2 : q3), wk = k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3];
                before = rowget_i64(pk - wk, q >> 2);
                tsum = rowget_i64(wk, q >> 2); tk = true;
            }
            carry += rowget_i64(incl, 15);
            if (!__ballot(on && jt < 0 && tb + 64 < nt)) break;
        }
        if (nt > 0) { if (jt < 0) { none = true; before = carry; } else base = jt * 64; }
    }
    // the tile at `base` (every list is followed by one tile of padding; a row without a choice reads its tile 0)
    const QV4 c = *(const QV4 *)(l32 + base + 4 * rl);
    const QV4 hh = *(const QV4 *)(lh2 + base + 4 * rl);
    const int e0 = base + 4 * rl;
    const int c0 = (!none && e0 + 0 < n) ? c.x : 0, c1 = (!none && e0 + 1 < n) ? c.y : 0, c2 = (!none && e0 + 2 < n) ? c.z : 0, c3 = (!none && e0 + 3 < n) ? c.w : 0;
    const int q0 = c0, q1 = q0 + c1, q2 = q1 + c2, q3 = q2 + c3;
    const int incl = row_iscan32(q3), ex = incl - q3;
    int kk = 4;
    if (c3 != 0 && !((double)(before + ex + q3) < rr_)) kk = 3;
    if (c2 != 0 && !((double)(before + ex + q2) < rr_)) kk = 2;
    if (c1 != 0 && !((double)(before + ex + q1) < rr_)) kk = 1;
    if (c0 != 0 && !((double)(before + ex + q0) < rr_)) kk = 0;
    const int q = row_min(kk < 4 ? 4 * rl + kk : 64);
    if (q < 64) {
        const int k = q & 3;
        o.k = base + q;
        o.total = before + rowget_i32(ex + (k == 0 ? q0 : k == 1 ? q1 : k == 2 ? q2 : q3), q >> 2);
        o.wi = rowget_i32(k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3, q >> 2);
        o.hap = rowget_i32(k == 0 ? hh.x : k == 1 ? hh.y : k == 2 ? hh.z : hh.w, q >> 2);
    } else {
        o.k = -1; o.total = before + rowget_i32(incl, 15); o.wi = 0; o.hap = 0;
    }
    o.none = none; o.tsum = tsum; o.tk = tk;
    return o;
}

}  // namespace

static __device__ __forceinline__ void quadf_body(const VgxDirectArgs &a, const QFArgs &qa) {
    const int lane = threadIdx.x, row = lane >> 4, rl = lane & 15;
    const VgxDevParams &p = a.p;
    const VgxDevRep &r = a.r;
    const int P = p.P, sites = p.sites, H = p.H;
    const int64_t R = a.n_replicates;
    const int64_t rep_raw = (int64_t)blockIdx.x * 4 + row;
    const bool live = rep_raw < R;
    const int64_t rep = live ? rep_raw : R - 1;   // idle rows shadow the last replicate read-only
    const int nslot = (P + 15) >> 4;
    const bool has_mig = qa.has_mig[0] != 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *k_kmig = (double *)smem, *k_smul = k_kmig + 128, *k_mebm = k_smul + 64;
    uint64_t *k_jump = (uint64_t *)(smem + Q_CONST_BYTES) + rl * 4;
    unsigned char *blk = smem + Q_CONST_BYTES + Q_RNG_BYTES + row * Q_REP_BYTES;
    double *s_inf = (double *)blk, *s_bc = s_inf + 64;
    int64_t *s_ts = (int64_t *)(s_bc + 64), *s_ti = s_ts + 64;
    int32_t *s_nocc = (int32_t *)(s_ti + 64);
    double *s_cc = (double *)(s_nocc + 64);
    int64_t *s_cnt = (int64_t *)(s_cc + 4);
    uint64_t *s_inc = (uint64_t *)(s_cnt + 6);
#define QBUMP(i) do { if (rl == 0) s_cnt[i] += 1; } while (0)

    // the single rate class
    const double c_b = p.cb_b[0], c_sig = p.cb_sigma[0], c_d = p.c_d[0], c_s = p.c_s[0], c_tm = p.c_tm[0];

    // ---- load state ----
    {
        const double *gD = r.popD + rep * PD_COUNT * P;
        const int64_t *gI64 = r.popI + rep * PI_COUNT * P;
        const int32_t *gN = r.nocc + rep * P;
        const int pn = lane;
        // BirthRate's sum over the source populations, factored (constant: no lockdown can switch): tree order
        double km = 0.0;
        if (pn < P) {
            const double *mrow = p.mig + (int64_t)pn * P;
            for (int q = 0; q < P; ++q) km += mrow[q] * mrow[q] * gD[PD_CD * P + q] / p.actualSizes[q];
        }
        k_kmig[pn] = km;
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
        }
    }
    WSYNC();

    const int64_t cap = r.cap, capT = r.capT;
    int32_t *lhap = r.lhap + rep * P * cap;
    int32_t *lcls = r.lcls + rep * P * cap;
    int64_t *lcnt = r.lcnt + rep * P * cap;
    int32_t *lcnt32 = r.lcnt32 + rep * P * cap;
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
    int att = 0, restarts = 0, last_att = -1, traj_next = 0;
    int st = live ? ST_REBUILD : ST_DONE, err = 0;
    bool open = false;
    const double tlimit = (double)a.time;
    const bool has_tl = !(a.time == -1.0f);

    // random stream of the row (see vgx_quad.hip): 16 outputs per refill by lane-parallel jump-ahead
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
    uint64_t g_sh = 0, g_sl = 0;
    double g_val = 0.0;
    int pos = 8;
    // prefix sums of popRate over the populations 4 rl .. 4 rl + 3 of this lane (tree order), refreshed by every update
    double pre0 = 0.0, pre1 = 0.0, pre2 = 0.0, pre3 = 0.0;

#ifdef VGX_PROFILE
    unsigned long long prof_acc[VGX_PROF_SLOTS], prof_t0 = __builtin_readcyclecounter();
    for (int i = 0; i < VGX_PROF_SLOTS; ++i) prof_acc[i] = 0;
#endif
    while (true) {
        const bool run = st != ST_DONE;
        if (!__ballot(run)) break;
        QPROF(0);
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

        int u_lo = 0, u_hi = rebuild ? P : 0;
        int op_n = 0, op_pi = 0, op_h0 = 0;      // a deferred infectious[op_pi, op_h0] += 1 (mutation target, migrant)
        int e_type = -1, e_hap = 0, e_pop = 0, e_nh = 0, e_np = 0;
        double den = 0.0;

        if (__ballot(ev)) {
            // ---- random numbers: refill the batch of every row that ran dry ----
            if (__builtin_expect(__ballot(ev && pos == 8) != 0, 0)) {   // (every eighth iteration while the rows stay in step)
                const bool fill = ev && pos == 8;
                if (a.rng_philox) {
                    // the counter-based stream (vgx_run_opts.mode = 2): iteration i of the attempt takes outputs 2 i (time) and 2 i + 1
                    // (event) of the stream of (seed, attempt); every lane forms its own output, no state is carried
                    const double u = vgx_philox_stream_double((uint64_t)r.seeds[rep], (uint32_t)last_att, 2 * (uint64_t)att_loops + (uint64_t)rl);
                    const double v = (rl & 1) ? u : -vgx_log(u);
                    if (fill) { g_val = v; pos = 0; }
                } else {
                    uint64_t h, l, ch, cl;
                    vgx_mul128(k_jump[0], k_jump[1], g_sh, g_sl, h, l);
                    vgx_mul128(k_jump[2], k_jump[3], s_inc[0], s_inc[1], ch, cl);
                    vgx_add128(h, l, ch, cl);
                    const double u = vgx_pcg64_output_double(h, l);
                    const double v = (rl & 1) ? u : -vgx_log(u);
                    const uint64_t nh = (uint64_t)rowget_i64((int64_t)h, 15), nl = (uint64_t)rowget_i64((int64_t)l, 15);
                    if (fill) { g_val = v; g_sh = nh; g_sl = nl; pos = 0; }
                }
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

            // ================= GenerateEvent (pyx:483-512) =================
            double rn = u2;
            const double choose0 = rn * den;
            double choose = choose0;
            const bool evn = ev && (totalRate > choose);   // an event inside a population
            const bool evm = ev && !evn;                   // a migration attempt

            // ---- population by fastChoose over popRate = infectPopRate: the lane's cached prefix sums ----
            int pi = 0;
            {
                rn = choose / totalRate;
                const double rr_ = totalRate * rn;
                const int jl = !(pre0 < rr_) ? 0 : !(pre1 < rr_) ? 1 : !(pre2 < rr_) ? 2 : !(pre3 < rr_) ? 3 : 4;
                const int q = row_min(jl < 4 && 4 * rl + jl < P ? 4 * rl + jl : 64);
                double total, wi;
                if (q < 64) {
                    pi = q;
                    const int j = q & 3;
                    total = rowget_f64(j == 0 ? pre0 : j == 1 ? pre1 : j == 2 ? pre2 : pre3, q >> 2);
                    wi = s_inf[pi];
                } else { pi = P - 1; total = totalRate; wi = s_inf[P - 1]; }       // clamp at n-1 (fc:26)
                if (evn && wi == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 1;
                rn = (rr_ - (total - wi)) / wi;
                choose = rn * wi;
            }
            const double infect_pi = s_inf[pi];
            rn = (choose - 0.0) / infect_pi;               // immunePopRate = +0.0 (one susceptibility group)
            const double bC = s_bc[pi];
            const double smul = k_smul[pi];
            const double tE = ((bC + c_d) + smul) + c_tm;
            const int n_sel = evn ? s_nocc[pi] : 0;
            int32_t *l3 = lcnt32 + (int64_t)pi * cap;
            int64_t *lt = ltsum + (int64_t)pi * capT;

            // ---- haplotype: one rate class, so hapPopRate is proportional to the counts: integer prefix search ----
            const int64_t ti_sel = s_ti[pi];
            const double r2 = (double)ti_sel * rn;
            QPROF(3);
            const QCount hs = q_count_select(lhap + (int64_t)pi * cap, l3, lt, n_sel, r2, evn);
            QPROF(4);
            int k_hit = hs.k, hap_hit = hs.hap;
            int64_t cnt_hit = hs.wi, tot_hit = hs.total;
            const int64_t ts_hit = hs.tsum;
            bool tk_hit = hs.tk;
            if (evn && err == 0 && k_hit < 0) {
                // nothing reached r: the dense loop runs on to index H-1 (fc:26), a valid pick only if that haplotype is occupied
                const int32_t *lh = lhap + (int64_t)pi * cap;
                if (n_sel > 0 && lh[n_sel - 1] == H - 1 && l3[n_sel - 1] != 0) { k_hit = n_sel - 1; cnt_hit = l3[n_sel - 1]; hap_hit = H - 1; tk_hit = false; }
                else err = Q_ERR_ZERO_WEIGHT + 256 * 2;
            }
            if (k_hit < 0) { k_hit = 0; cnt_hit = 1; }
            const bool evn_ok = evn && err == 0;
            rn = (r2 - (double)(tot_hit - cnt_hit)) / (double)cnt_hit;

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
                if ((double)ts_pi * c_sig == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 6;
                if (rl == 0) { s_ts[pi] = ts_pi - 1; s_ti[pi] = ti_pi + 1; }
                gI += 1; QBUMP(QC_B);
                if (live && rl == 0) { l3[k_hit] = (int32_t)(cnt_hit + 1); if (n_sel > 64) lt[k_hit >> 6] = (tk_hit ? ts_hit : lt[k_hit >> 6]) + 1; }
                e_type = QEV_BIRTH; e_hap = hap_hit; e_pop = pi; e_nh = 0; e_np = H;
                u_lo = pi; u_hi = pi + 1;
            }
            if (isD) {
                if (rl == 0) { s_ts[pi] = ts_pi + 1; s_ti[pi] = ti_pi - 1; }
                gI -= 1;
                if (ei == 2) { cS += 1; e_type = QEV_SAMPLING; } else { QBUMP(QC_D); e_type = QEV_DEATH; }
                if (live && rl == 0)   // (a count of 0 stays in the list, see the list operations below)
                    { l3[k_hit] = (int32_t)(cnt_hit - 1); if (n_sel > 64) lt[k_hit >> 6] = (tk_hit ? ts_hit : lt[k_hit >> 6]) - 1; }
                e_hap = hap_hit; e_pop = pi; e_nh = 0; e_np = 0;
                u_lo = pi; u_hi = pi + 1;
            }
            QPROF(5);
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
                    if (live && rl == 0) { l3[k_hit] = (int32_t)(cnt_hit - 1); if (n_sel > 64) lt[k_hit >> 6] = (tk_hit ? ts_hit : lt[k_hit >> 6]) - 1; }
                    op_n = 1; op_pi = pi; op_h0 = nhi;
                    QBUMP(QC_M);
                    e_type = QEV_MUTATION; e_hap = hap_hit; e_pop = pi; e_nh = nhi; e_np = 0;
                    u_lo = pi; u_hi = pi + 1;
                }
            }
            if (__builtin_expect(__ballot(evm) != 0, 0)) {
                // ================= GenerateMigration (pyx:672-694) =================
                double rm = (choose0 - totalRate) / totalMig;
                // target population by fastChoose over migPopRate (tree prefix: the order totalMigrationRate was summed in)
                int tpi = 0;
                {
                    const double rr_ = totalMig * rm;
                    double w4[4], q0, q1, q2, q3;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pn = 4 * rl + j;
                        w4[j] = pn < P ? k_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]) : 0.0;
                    }
                    q0 = w4[0]; q1 = q0 + w4[1]; q2 = q1 + w4[2]; q3 = q2 + w4[3];
                    const double ex = row_prev_f64(row_fscan(q3));
                    q0 += ex; q1 += ex; q2 += ex; q3 += ex;
                    const int jl = !(q0 < rr_) ? 0 : !(q1 < rr_) ? 1 : !(q2 < rr_) ? 2 : !(q3 < rr_) ? 3 : 4;
                    const int q = row_min(jl < 4 && 4 * rl + jl < P ? 4 * rl + jl : 64);
                    tpi = q < 64 ? q : P - 1;
                    const int j = tpi & 3;
                    const double tot_hit2 = q < 64 ? rowget_f64(j == 0 ? q0 : j == 1 ? q1 : j == 2 ? q2 : q3, tpi >> 2) : totalMig;
                    const double w_h = rowget_f64(j == 0 ? w4[0] : j == 1 ? w4[1] : j == 2 ? w4[2] : w4[3], tpi >> 2);
                    if (evm && w_h == 0.0) err = Q_ERR_ZERO_WEIGHT + 256 * 9;
                    rm = (rr_ - (tot_hit2 - w_h)) / w_h;
                }
                // source population: fastChoose_skip(totalInfectious, globalInfectious - totalInfectious[tpi], rn, skip = tpi)
                int spi = -1;
                {
                    const double rr_ = (double)(gI - s_ti[tpi]) * rm;
                    const int start = tpi == 0 ? 1 : 0;
                    int64_t w4[4], q0, q1, q2, q3;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pn = 4 * rl + j;
                        w4[j] = (pn < P && pn != tpi && pn >= start) ? s_ti[pn] : 0;
                    }
                    q0 = w4[0]; q1 = q0 + w4[1]; q2 = q1 + w4[2]; q3 = q2 + w4[3];
                    const int64_t incl = row_iscan(q3);
                    const int64_t ex = incl - q3;
                    q0 += ex; q1 += ex; q2 += ex; q3 += ex;
                    int jl = 4;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const int pn = 4 * rl + j;
                        const int64_t qv = j == 0 ? q0 : j == 1 ? q1 : j == 2 ? q2 : q3;
                        if (pn < P && pn != tpi && pn >= start && !((double)qv < rr_)) jl = j;
                    }
                    const int q = row_min(jl < 4 ? 4 * rl + jl : 64);
                    int64_t total;
                    if (q < 64) { spi = q; const int j = q & 3; total = rowget_i64(j == 0 ? q0 : j == 1 ? q1 : j == 2 ? q2 : q3, q >> 2); }
                    else { spi = P - 1; total = rowget_i64(incl, 15); }   // clamp at n-1 (may equal skip only then)
                    const int64_t wi = s_ti[spi];
                    if (evm && wi == 0) err = Q_ERR_ZERO_WEIGHT + 256 * 10;
                    rm = (rr_ - (double)(total - wi)) / (double)wi;
                }
                // haplotype by fastChoose(infectious[spi], totalInfectious[spi], rn): int64 weights over the occupancy list
                int hi = 0;
                {
                    const int n = (evm && err == 0) ? s_nocc[spi] : 0;
                    const int32_t *lh2 = lhap + (int64_t)spi * cap;
                    const int32_t *ln2 = lcnt32 + (int64_t)spi * cap;
                    const double rr_ = (double)s_ti[spi] * rm;
                    const QCount ms = q_count_select(lh2, lcnt32 + (int64_t)spi * cap, ltsum + (int64_t)spi * capT, n, rr_, evm);
                    int kq = ms.k;
                    int64_t total = ms.total, wi = ms.wi;
                    if (evm && err == 0 && kq < 0) {
                        if (n > 0 && lh2[n - 1] == H - 1 && ln2[n - 1] != 0) { kq = n - 1; wi = ln2[n - 1]; }
                        else { err = Q_ERR_ZERO_WEIGHT + 256 * 11; kq = 0; wi = 1; }
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
                        op_n = 1; op_pi = tpi; op_h0 = hi;
                        e_type = QEV_MIGRATION; e_hap = hi; e_pop = spi; e_nh = 0; e_np = tpi;
                        u_lo = tpi; u_hi = tpi + 1;
                    } else {
                        QBUMP(QC_MIGN);
                    }
                }
            }
            QPROF(6);
            WSYNC();
            QPROF(7);
        }

        // ================= deferred list operations: infectious[op_pi, hap] += delta, list kept ordered =================
        if (err != 0) op_n = 0;
        if (__builtin_expect(__ballot(live && op_n > 0) != 0, 0)) {
            const bool act = live && op_n > 0;
            const int hap = op_h0;
            int n = act ? s_nocc[op_pi] : 0;
            const int n_was = n;
            int zeros_unused = 0;
            const bool fits = q_list_add_one<false>(act, hap, lhap + (int64_t)op_pi * cap, lcls + (int64_t)op_pi * cap, lcnt32 + (int64_t)op_pi * cap,
                                                    ltsum + (int64_t)op_pi * capT, nullptr, n, (int)cap, H, zeros_unused);
            if (act && !fits) err = Q_ERR_CAPACITY;
            if (act && n != n_was && rl == 0) s_nocc[op_pi] = n;
            WSYNC();
            QPROF(9);
        }

        QPROF(10);
        // ================= Events.AddEvent (events.pxi:37-44) =================
        if (err == 0 && e_type >= 0) {
            if (a.record_events) {
                const int64_t slot = ev_ptr - r.ev_base;
                if (slot >= 0 && slot < r.evcap) {
                    if (live) {
                        if (rl < VGX_EV_COLS) {
                            const int v = rl == 0 ? e_type : rl == 1 ? e_hap : rl == 2 ? e_pop : rl == 3 ? e_nh : rl == 4 ? e_np
                                                                                                          : (int)(uint32_t)att_loops;
                            r.ev_cols[(rep * r.evcap + slot) * VGX_EV_COLS + rl] = v;
                        } else if (rl == VGX_EV_COLS) {
                            r.ev_rate[rep * r.evcap + slot] = den;
                        }
                    }
                } else {
                    err = Q_ERR_CAPACITY;
                }
            }
            ev_ptr += 1;
        }

        QPROF(11);
        // ================= UpdateRates for [u_lo, u_hi) (pyx:516-546) / UpdateAllRates (pyx:279-351), order-free =================
        if (err != 0) u_hi = u_lo;
        const int maxu = rows_max(u_hi - u_lo);
        if (maxu > 0) {
            for (int us = 0; us < maxu; ++us) {
                const int pn0 = u_lo + us;
                const bool act = pn0 < u_hi;
                const int pu = act ? pn0 : 0;
                // BirthRate of the class, factored; tEventHapPopRate; infectPopRate = tEvent x totalInfectious (one rate class)
                const double bCu = c_b * (((double)s_ts[pu] * c_sig) * k_kmig[pu]);
                const double tEu = ((bCu + c_d) + k_smul[pu]) + c_tm;
                const double inf_u = tEu * (double)s_ti[pu];
                if (act && rl == 0) { s_bc[pu] = bCu; s_inf[pu] = inf_u; }
            }
            WSYNC();
            // totalRate and the prefix sums of popRate over the populations (pyx:537-539) in tree order
            {
                const double w0 = s_inf[4 * rl], w1 = s_inf[4 * rl + 1], w2 = s_inf[4 * rl + 2], w3 = s_inf[4 * rl + 3];   // (+0.0 beyond P)
                double q0 = w0, q1 = q0 + w1, q2 = q1 + w2, q3 = q2 + w3;
                const double incl = row_fscan(q3), ex = row_prev_f64(incl);
                q0 += ex; q1 += ex; q2 += ex; q3 += ex;
                const double tot = rowget_f64(q3, 15);
                if (u_hi > u_lo) { pre0 = q0; pre1 = q1; pre2 = q2; pre3 = q3; totalRate = tot; }
            }
            // totalMigrationRate = sum of maxEffectiveBirthMigration * totalSusceptible * (globalInfectious - totalInfectious)
            if (has_mig) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pn = 4 * rl + j;
                    if (pn < P) acc += k_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]);
                }
                const double tot = rowget_f64(row_fscan(acc), 15);
                if (u_hi > u_lo) totalMig = tot;
            }
        }

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
                if (rl < 6) s_cnt[rl] = 0;
                t_now = 0.0; traj_next = 0;
                restarts += 1; att += 1;
                st = ST_REBUILD;
            } else {
                good_attempt = (int64_t)att + 1;
                st = ST_DONE;
            }
        }
        QPROF(13);
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
                            }
                        }
                        tsum += rowget_i64(row_iscan(ct), 15);
                    }
                    if (rs && rl == 0) ltsum[(int64_t)pn * capT + base / 64] = tsum;
                    ti += tsum;
                }
                if (rs)
                    for (int j = (n + 63) / 64 + rl; j <= n_old / 64 && j < capT; j += 16) ltsum[(int64_t)pn * capT + j] = 0;
                if (rs && rl == 0) { s_nocc[pn] = n; s_ts[pn] = r.i_sus[pn]; s_ti[pn] = ti; }
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

    // ---- state back to HBM ----
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
#ifdef VGX_PROFILE
            for (int i = 0; i < VGX_PROF_SLOTS; ++i) r.prof[rep * VGX_PROF_SLOTS + i] = prof_acc[i];
#endif
        }
    }
}


// After the event loop: zero-count entries leave the lists, the tile sums and the 8-byte counts (the loop keeps the 4-byte ones only)
// are rewritten; one list per 16-lane row.
extern "C" __global__ void __launch_bounds__(256) vgx_lists_settle_kernel(int32_t *lhap, int32_t *c32, int64_t *c64, int64_t *ltsum, int32_t *nocc,
                                                                           int64_t lists, int64_t cap, int64_t capT) {
    const int64_t li = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool on = li < lists;
    const int64_t l = on ? li : 0;
    const int n = on ? nocc[l] : 0;
    const int nn = q_compact_list<4>(lhap + l * cap, c32 + l * cap, ltsum + l * capT, n, on, c64 ? c64 + l * cap : nullptr);
    if (on && (threadIdx.x & 15) == 0) nocc[l] = nn;
}

extern "C" __global__ void __launch_bounds__(64, 3) vgx_quadf_kernel(VgxDirectArgs a, QFArgs qa) { quadf_body(a, qa); }

extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_quad_prep(const VgxDevParams *p, const double *cd, double *effMig,
                                                                                 double *maxEBM, int32_t *has_mig, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_lists_settle(const VgxDirectArgs *a, hipStream_t stream) {
    const int64_t lists = (int64_t)a->n_replicates * a->p.P;
    // (the 8-byte counts are rewritten on demand: vgxi_launch_counts64, the engine's counts64_valid)
    hipLaunchKernelGGL(vgx_lists_settle_kernel, dim3((unsigned)((lists + 15) / 16)), dim3(256), 0, stream, a->r.lhap, a->r.lcnt32, (int64_t *)nullptr,
                       a->r.ltsum, a->r.nocc, lists, (int64_t)a->r.cap, (int64_t)a->r.capT);
    return hipGetLastError();
}
// The 8-byte counts from the 4-byte ones, for the kernels and the copy-out that read them.
extern "C" __global__ void __launch_bounds__(256) vgx_counts64_kernel(const int32_t *c32, int64_t *c64, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c64[i] = c32[i];
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_counts64(const int32_t *c32, int64_t *c64, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(vgx_counts64_kernel, dim3(8192), dim3(256), 0, stream, c32, c64, n);
    return hipGetLastError();
}
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_quadf(const VgxDirectArgs *a, const double *cd, double *effMig,
                                                                             double *maxEBM, int32_t *has_mig, hipStream_t stream) {
    hipError_t err = vgxi_launch_quad_prep(&a->p, cd, effMig, maxEBM, has_mig, stream);
    if (err != hipSuccess) return err;
    QFArgs qa;
    qa.effMig = effMig; qa.maxEBM = maxEBM; qa.has_mig = has_mig;
    const unsigned grid = (unsigned)((a->n_replicates + 3) / 4);
    hipLaunchKernelGGL(vgx_quadf_kernel, dim3(grid), dim3(64), Q_LDS_BYTES, stream, *a, qa);
    return vgxi_launch_lists_settle(a, stream);
}
