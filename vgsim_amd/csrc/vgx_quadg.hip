// vgx_quadg.hip — the row-per-replicate direct-Gillespie kernel for GENERAL models: four replicates per wavefront (one per
// 16-lane DPP row, vgx_rowprim.h) with several susceptibility groups, several rate classes, populations that switch their
// lockdown state and up to 128 populations — everything vgx_quad.hip (one group, one class, no switch, <= 64 populations)
// leaves to the one-replicate-per-wavefront kernel of vgx_direct.hip.
//
// Same path, same contract as the other two: SimulatePopulation pyx:396-429 and everything it calls (SampleTime pyx:476,
// GenerateEvent pyx:483, UpdateRates pyx:516, ImmunityTransition pyx:550, Birth pyx:568 (the recombination branch pyx:575-596 in the
// *_rec instantiations),
// Death/Sampling pyx:616/630, Mutation pyx:640, GenerateMigration pyx:672, CheckLockdown pyx:698, Restart pyx:714,
// UpdateAllRates pyx:279, fastChoose / fastChoose_skip fast_choose.pxi:18/36, Events.AddEvent events.pxi:37); every sum the
// reference forms left to right is formed left to right, no contraction: event chains are bit-identical.
//
// Layout.  Lane (row, l) holds populations l, l+16, ... (NS slots of 16: P <= 16 NS) of its replicate and entry 16c + l of a list
// chunk.  Per replicate the hot per-population arrays (popRate, infectPopRate, immunePopRate, maxEffectiveBirthMigration,
// contactDensity, totals, list lengths, lockdown flags) live in LDS for the whole run; the COLD RECORD of a population —
// susceptible[S], immuneSourcePopRate[S], the susceptible counts as of the population's last infect-update[S] (which
// reproduce the stale susceptHapPopRate of pyx:384 bit for bit) and eventHapPopRate[., ., 0] per birth class[CB] — is W = 3S + CB
// 8-byte words in global memory (L2-resident), staged in LDS for the ONE population an iteration works on.
//
// BirthRate (pyx:382-392) is evaluated per birth class as a PROGRAM OF CHAIN SEGMENTS built by the host (vgx_api.hip):
// a segment adds, for one susceptibility group sn with a non-zero susceptibility sigma, the P terms
// ((S[pi,sn] sigma) m m cd) / as to the sum its parent segment left — the reference's own loop nest for that class, with the
// groups of zero susceptibility left out (they add +0.0) and common prefixes (sn, sigma) of different classes evaluated once:
// same operations on the same operands in the same order, so every class gets the reference's bits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "vgx_dev.h"
#include "vgx_rng.h"
#include "vgx_wave.h"
#include "vgx_rowprim.h"
#include "vgx_quadg.h"

#ifndef VGX_QUADG_WAVES
#define VGX_QUADG_WAVES 2
#endif

namespace {

enum { G_ERR_ZERO_WEIGHT = 3, G_ERR_CAPACITY = 4, G_ERR_LOOP_GUARD = 5 };
enum { GEV_BIRTH = 0, GEV_DEATH, GEV_SAMPLING, GEV_MUTATION, GEV_SUSCCHANGE, GEV_MIGRATION };
enum { GS_REBUILD = 0, GS_FULL = 1, GS_RUN = 2, GS_DONE = 3 };
enum { GC_B = 0, GC_D, GC_M, GC_I, GC_SWAP, GC_MIGP, GC_MIGN };

// row-uniform "any lane of the row": maximum of a 0/1 flag over the row
static __device__ __forceinline__ bool row_any(bool f) { return row_max(f ? 1 : 0) != 0; }

}  // namespace

// RECOMB: the instantiations with the recombination branch of Birth (pyx:575-596); the others do not carry its code.
// The kernel's arguments are read through the kernarg segment pointer (constant address space: scalar loads where and when a value is
// needed).  As by-value parameters whose fields are referenced all over the loop they were kept in scalar registers for the whole
// kernel — 300 to 470 of them spilled to lanes of vector registers, every reload an instruction in the event loop (vgx_solo.hip).
struct VgxQuadgKArgs { VgxDirectArgs a; VgxQuadgArgs qa; };
typedef const VgxQuadgKArgs __attribute__((address_space(4))) *QuadgKA;

template <int NS, bool RECOMB = false>
static __device__ __forceinline__ void quadg_body() {
    const QuadgKA ka = (QuadgKA)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &a = ka->a;
    const auto &qa = ka->qa;
    const int lane = threadIdx.x, row = lane >> 4, rl = lane & 15;
    const auto &p = a.p;
    const auto &r = a.r;
    const int P = p.P, S = p.S, C = p.C, CB = p.CB, sites = p.sites, H = p.H, W = qa.W, NSEG = qa.nseg;
    constexpr int PL = 16 * NS;
    const int64_t R = a.n_replicates;
    const int64_t rep_raw = (int64_t)blockIdx.x * 4 + row;
    const bool live = rep_raw < R;
    const int64_t rep = live ? rep_raw : R - 1;   // idle rows shadow the last replicate read-only

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgxQuadgLayout L = vgx_quadg_layout(PL, S, C, CB, NSEG, W);
    double *k_as = (double *)(smem + L.k_as), *k_thS = (double *)(smem + L.k_thS), *k_thE = (double *)(smem + L.k_thE);
    double *k_mult = (double *)(smem + L.k_mult), *k_d = (double *)(smem + L.k_d), *k_s = (double *)(smem + L.k_s);
    double *k_tm = (double *)(smem + L.k_tm), *k_cbb = (double *)(smem + L.k_cbb), *k_sig = (double *)(smem + L.k_sig);
    double *k_cumul = (double *)(smem + L.k_cumul), *k_trans = (double *)(smem + L.k_trans), *k_segsig = (double *)(smem + L.k_segsig);
    uint64_t *k_jump = (uint64_t *)(smem + L.k_jump) + rl * 4;
    int32_t *k_bidx = (int32_t *)(smem + L.k_bidx), *k_stype = (int32_t *)(smem + L.k_stype);
    int32_t *k_segpar = (int32_t *)(smem + L.k_segpar), *k_segsn = (int32_t *)(smem + L.k_segsn), *k_cbseg = (int32_t *)(smem + L.k_cbseg);
    unsigned char *blk = smem + L.rows + row * L.row_bytes;
    double *s_inf = (double *)(blk + L.s_inf), *s_imm = (double *)(blk + L.s_imm);
    // popRate[pn] = infectPopRate[pn] + immunePopRate[pn] (pyx:534) is formed where it is read: the same addition, and the block of
    // a wavefront at 97-112 populations stays within a fifth of a CU's LDS
#define S_POP(i) (s_inf[(i)] + s_imm[(i)])
    double *s_mebm = (double *)(blk + L.s_mebm), *s_cd = (double *)(blk + L.s_cd), *s_cc = (double *)(blk + L.s_cc);
    double *s_seg = (double *)(blk + L.s_seg);
    double *s_recd = (double *)(blk + L.s_rec);
    int64_t *s_reci = (int64_t *)(blk + L.s_rec);
    int64_t *s_ts = (int64_t *)(blk + L.s_ts), *s_ti = (int64_t *)(blk + L.s_ti), *s_cnt = (int64_t *)(blk + L.s_cnt);
    uint64_t *s_inc = (uint64_t *)(blk + L.s_inc);
    int32_t *s_nocc = (int32_t *)(blk + L.s_nocc), *s_lock = (int32_t *)(blk + L.s_lock);
#define GBUMP(i) do { if (rl == 0) s_cnt[i] += 1; } while (0)
    // In-kernel stamps (diagnostic build only, -DVGX_PROFILE): shader cycles per phase, summed per wavefront into r.prof of its first
    // replicate (tools/profile_quadg.py).  Stamp k closes the phase that precedes it.
#ifdef VGX_PROFILE
    unsigned long long prof_acc[VGX_PROF_SLOTS], prof_t0 = __builtin_readcyclecounter();
    for (int i = 0; i < VGX_PROF_SLOTS; ++i) prof_acc[i] = 0;
#define GPROF(i) do { unsigned long long prof_t1 = __builtin_readcyclecounter(); prof_acc[i] += prof_t1 - prof_t0; prof_t0 = prof_t1; } while (0)
#else
#define GPROF(i)
#endif
    // words of the staged cold record
#define R_SUS(j) s_reci[(j)]
#define R_IMS(j) s_recd[S + (j)]
#define R_SNP(j) s_reci[2 * S + (j)]
#define R_BC(j) s_recd[3 * S + (j)]

    // ---- tables shared by the four replicates ----
    for (int i = lane; i < PL; i += 64) {
        const bool ok = i < P;
        k_as[i] = ok ? p.actualSizes[i] : 1.0;
        k_thS[i] = ok ? p.startLD[i] * (double)p.sizes[i] : 0.0;   // thresholds of CheckLockdown (pyx:699, 705)
        k_thE[i] = ok ? p.endLD[i] * (double)p.sizes[i] : 0.0;
        k_mult[i] = ok ? p.sampMult[i] : 0.0;
    }
    for (int i = lane; i < C; i += 64) {
        k_d[i] = p.c_d[i]; k_s[i] = p.c_s[i]; k_tm[i] = p.c_tm[i]; k_bidx[i] = p.c_bidx[i]; k_stype[i] = p.c_stype[i];
    }
    for (int i = lane; i < CB; i += 64) { k_cbb[i] = p.cb_b[i]; k_cbseg[i] = qa.cb_seg[i]; }
    for (int i = lane; i < CB * S; i += 64) k_sig[i] = p.cb_sigma[i];
    for (int i = lane; i < S; i += 64) k_cumul[i] = p.suscepCumul[i];
    for (int i = lane; i < S * S; i += 64) k_trans[i] = p.suscepTransition[i];
    for (int i = lane; i < NSEG; i += 64) { k_segsig[i] = qa.seg_sig[i]; k_segpar[i] = qa.seg_par[i]; k_segsn[i] = qa.seg_sn[i]; }

    // ---- load state ----
    int64_t *cold = qa.cold + rep * (int64_t)P * W;
    bool eff_dirty = false;       // the replicate's contact densities differ from those effMig0 / mebm0 were computed for
    bool ld_any = false;          // some population can switch its lockdown state at all
    {
        const double *gD = r.popD + rep * PD_COUNT * P;
        const int64_t *gI64 = r.popI + rep * PI_COUNT * P;
        const int32_t *gN = r.nocc + rep * P;
        bool dd = false, la = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int pq = s * 16 + rl;
            const bool ok = pq < P;
            s_inf[pq] = 0.0; s_imm[pq] = 0.0;
            s_mebm[pq] = ok ? qa.mebm0[pq] : 0.0;
            const double cdv = ok ? gD[PD_CD * P + pq] : 0.0;
            s_cd[pq] = cdv;
            if (ok && cdv != qa.cd0[pq]) dd = true;
            s_ts[pq] = ok ? gI64[PI_TOTSUS * P + pq] : 0;
            s_ti[pq] = ok ? gI64[PI_TOTINF * P + pq] : 0;
            const int lk = ok ? (int)gI64[PI_LOCK * P + pq] : 0;
            s_lock[pq] = lk;
            s_nocc[pq] = ok ? gN[pq] : 0;
            // a population can switch on only if its threshold lies below its size, off only if it is on
            if (ok && (p.startLD[pq] * (double)p.sizes[pq] < (double)p.sizes[pq] || lk != 0)) la = true;
        }
        if (rl < 8) s_cc[rl] = 0.0;
        eff_dirty = row_any(dd);
        ld_any = row_any(la);
        // cold records: the susceptible counts; everything else in them is rebuilt by the first pass (UpdateAllRates)
        if (live) {
            for (int pn = 0; pn < P; ++pn) {
                for (int w = rl; w < W; w += 16) cold[(int64_t)pn * W + w] = w < S ? r.sus[(rep * P + pn) * S + w] : 0;
            }
        }
    }

    const int64_t cap = r.cap, capT = r.capT;
    int32_t *lhap = r.lhap + rep * P * cap;
    int32_t *lcls = r.lcls + rep * P * cap;
    int64_t *lcnt = r.lcnt + rep * P * cap;
    int64_t *ltsum = r.ltsum + rep * P * capT;
    double *effP = r.effMig + rep * (int64_t)P * P;    // the replicate's own effectiveMigration once its contact densities differ
    const bool has_traj = r.traj != nullptr;
    const VgxRepScalars *sc = r.sc + rep;

    double t_now = sc->currentTime, totalRate = 0.0, totalMig = 0.0;
    int64_t gI = sc->globalInfectious, ev_ptr = sc->ev_ptr;
    int64_t rec_n = 0;      // recombination records of this call (kept across Restarts like upstream's `rec`)
    int64_t cS = sc->sCounter;
    if (rl == 0) {
        s_cnt[GC_B] = sc->bCounter; s_cnt[GC_D] = sc->dCounter; s_cnt[GC_M] = sc->mCounter; s_cnt[GC_I] = sc->iCounter;
        s_cnt[GC_SWAP] = sc->swapLockdown; s_cnt[GC_MIGP] = sc->migPlus; s_cnt[GC_MIGN] = sc->migNonPlus;
    }
    int64_t loops = 0, att_loops = 0, good_attempt = sc->good_attempt;
    int64_t att_ev0 = sc->ev_ptr, fa_n = 0;
    int att = 0, restarts = 0, last_att = -1, traj_next = 0, loc_n = 0, att_loc0 = 0;
    int st = live ? GS_REBUILD : GS_DONE, err = 0;
    bool open = false, eff_private = false, has_mig = qa.has_mig0[0] != 0;
    const double tlimit = (double)a.time;
    const bool has_tl = !(a.time == -1.0f);
    const int loc_cap = (int)r.loc_cap;
    int32_t *loc_rec = r.loc_rec + rep * r.loc_cap * 2;
    double *loc_time = r.loc_time + rep * r.loc_cap;
    int64_t *loc_iter = r.loc_iter + rep * r.loc_cap;

    // random stream of the row: 16 outputs (8 loop iterations) per refill (see vgx_quad.hip)
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

    // the cold record of population `pn` into the row's stage / back (rows with `on` only)
    auto rec_load = [&](int pn, bool on) {
        for (int w = rl; w < W; w += 16) {
            const int64_t v = cold[(int64_t)pn * W + w];
            if (on) s_reci[w] = v;
        }
        WSYNC();
    };
    auto rec_store = [&](int pn, bool on) {
        WSYNC();
        if (on && live)
            for (int w = rl; w < W; w += 16) cold[(int64_t)pn * W + w] = s_reci[w];
        WSYNC();
    };
    // tEventHapPopRate of class `cl` in the staged population (pyx:522-526), from its cached birth rate
    auto tE_of = [&](int cl, double mult) -> double {
        const double e0 = R_BC(k_bidx[cl]);
        return ((e0 + k_d[cl]) + k_s[cl] * mult) + k_tm[cl];
    };

    while (true) {
        const bool run = st != GS_DONE;
        if (!__ballot(run)) break;
        const bool rebuild = st == GS_REBUILD;
        const bool full = rebuild || st == GS_FULL;

        GPROF(0);
        // ================= front: open the attempt, loop condition (pyx:402-407) =================
        bool end_attempt = false, ev = false;
        if (st == GS_RUN) {
            if (!open) {
                if (att >= a.attempts) {
                    st = GS_DONE;
                } else {
                    VgxPcg64 sd;
                    vgx_pcg64_seed(sd, (uint64_t)r.seeds[rep], (uint32_t)att);
                    g_sh = sd.sh; g_sl = sd.sl;
                    if (rl == 0) { s_inc[0] = sd.ih; s_inc[1] = sd.il; }
                    pos = 8;
                    open = true;
                    last_att = att; att_loops = 0;
                    if (!(totalRate + totalMig != 0.0 && gI != 0)) end_attempt = true;   // pyx:404
                }
            }
            if (st == GS_RUN && !end_attempt &&
                !(ev_ptr < a.ev_size && (a.sample_size == -1 || cS <= a.sample_size) && (!has_tl || t_now < tlimit)))
                end_attempt = true;
            if (st == GS_RUN && !end_attempt) {
                if (loops >= a.max_loop) { err = G_ERR_LOOP_GUARD; st = GS_DONE; }
                else ev = true;
            }
        }

        // what this pass refreshes: one population with UpdateRates' flags (pyx:516), or everything
        int u_pop = -1;
        bool u_inf = false, u_imm = false, u_mig = false;
        int lk_pop = -1;              // population CheckLockdown looks at after the event (pyx:412)
        int op_n = 0, op_pi = 0, op_h0 = 0, op_h1 = 0, op_d0 = 0;
        int e_type = -1, e_hap = 0, e_pop = 0, e_nh = 0, e_np = 0;
        double den = 0.0;
        // a one-chunk list read for the haplotype choice stays in registers for the rate refresh
        int64_t ch_cn = 0;
        int ch_cl = 0, ch_pi = -1;

        if (__ballot(ev)) {
            GPROF(1);
            // ---- random numbers ----
            if (__builtin_expect(__ballot(ev && pos == 8) != 0, 0)) {
                const bool fill = ev && pos == 8;
                if (a.rng_philox) {
                    // the counter-based stream (vgx_run_opts.mode = 2 on a general model: this kernel's exact arithmetic on other random
                    // numbers): iteration i of the attempt takes outputs 2 i (time) and 2 i + 1 (event) of the stream of (seed, attempt),
                    // as in vgx_quadf.hip and in the host clock
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
            den = totalRate + totalMig;
            const double t_new = t_now + (nlog / den);   // SampleTime pyx:476-478
            if (has_traj) {
                while (true) {
                    const double tg = r.traj_t0 + (double)traj_next * r.traj_dt;
                    const bool emit = ev && live && traj_next < r.traj_points && tg < t_new;
                    if (__builtin_expect(!__ballot(emit), 1)) break;
                    if (emit) {
                        double *o = r.traj + (rep * r.traj_points + traj_next) * (int64_t)P * 2;
                        for (int s = 0; s < NS; ++s) {
                            const int pn = s * 16 + rl;
                            if (pn < P) { o[pn * 2 + 0] = (double)s_ti[pn]; o[pn * 2 + 1] = (double)s_ts[pn]; }
                        }
                        traj_next += 1;
                    }
                }
            }
            if (ev) t_now = t_new;

            GPROF(2);
            // ================= GenerateEvent (pyx:483-512) =================
            double rn = u2;
            const double choose0 = rn * den;
            double choose = choose0;
            const bool evn = ev && (totalRate > choose);   // an event inside a population
            const bool evm = ev && !evn;                   // a migration attempt

            // ---- population by fastChoose over popRate (fc:18-31): cached serial prefix sums at the slot ends ----
            int pi = 0;
            {
                rn = choose / totalRate;
                const double rr_ = totalRate * rn;
                int slot = NS - 1;
                bool any = false;
                double cin = 0.0, clast = 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double cs = s_cc[s];        // slots beyond P repeat the total
                    if (!any && !(cs < rr_)) { any = true; slot = s; cin = clast; }
                    clast = cs;
                }
                const double w = S_POP(slot * 16 + rl);
                double tot_;
                const double pre = row_scan16(w, cin, tot_);
                const int q = row_min(any && slot * 16 + rl < P && !(pre < rr_) ? rl : 16);
                double total, wi;
                if (q < 16) { pi = slot * 16 + q; total = rowget_f64(pre, q); wi = rowget_f64(w, q); }
                else { pi = P - 1; total = clast; wi = S_POP(P - 1); }       // clamp at n-1 (fc:26)
                if (evn && wi == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 1;
                rn = (rr_ - (total - wi)) / wi;
                choose = rn * wi;                    // pyx:493: rn * popRate[pi]
            }
            const double imm_pi = s_imm[pi], inf_pi = s_inf[pi];
            const bool isI = evn && err == 0 && (imm_pi > choose);    // pyx:494: ImmunityTransition
            const bool isN = evn && err == 0 && !isI;                 // an event of an infectious host

            GPROF(3);
            // ---- migration, first half (pyx:676-678): target and source population from the LDS arrays ----
            int tpi = 0, spi = 0;
            double rm = 0.0;
            if (__builtin_expect(__ballot(evm) != 0, 0)) {
                rm = (choose0 - totalRate) / totalMig;
                {
                    const double rr_ = totalMig * rm;
                    double carry = 0.0, tot_hit = 0.0, w_h = 0.0;
                    int cand = 4096;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int pn = s * 16 + rl;
                        const double w = pn < P ? s_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]) : 0.0;
                        double tot_;
                        const double pre = row_scan16(w, carry, tot_);
                        const int q = row_min(cand == 4096 && pn < P && !(pre < rr_) ? rl : 16);
                        if (cand == 4096 && q < 16) { cand = s * 16 + q; tot_hit = rowget_f64(pre, q); w_h = rowget_f64(w, q); }
                        else if (s == ((P - 1) >> 4) && cand == 4096) {        // clamp at P-1
                            const int ql = (P - 1) & 15;
                            tot_hit = rowget_f64(pre, ql); w_h = rowget_f64(w, ql);
                        }
                        carry = tot_;
                    }
                    tpi = cand < 4096 ? cand : P - 1;
                    if (evm && w_h == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 9;
                    rm = (rr_ - (tot_hit - w_h)) / w_h;
                }
                {   // fastChoose_skip(totalInfectious, globalInfectious - totalInfectious[tpi], rn, skip = tpi)
                    const double rr_ = (double)(gI - s_ti[tpi]) * rm;
                    const int start = tpi == 0 ? 1 : 0;
                    int64_t carry = 0, total = 0;
                    spi = -1;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
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
                    if (evm && wi == 0 && err == 0) err = G_ERR_ZERO_WEIGHT + 256 * 10;
                    rm = (rr_ - (double)(total - wi)) / (double)wi;
                }
            }

            GPROF(4);
            // ---- the cold record of the population this iteration works on ----
            const int sp = evn ? pi : (evm ? tpi : 0);
            rec_load(sp, ev);
            lk_pop = ev ? sp : -1;

            GPROF(5);
            // ---- ImmunityTransition (pyx:550-564) ----
            if (__builtin_expect(__ballot(isI) != 0, 0)) {
                double ri = choose / imm_pi;
                int ssi = 0, tsi = 0;
                {
                    const double rq = imm_pi * ri;
                    double total = R_IMS(0), wi = total;
                    bool stop = false;
                    for (int j = 1; j < S; ++j) {
                        const double wj = R_IMS(j);
                        if (!stop && total < rq) { ssi = j; total += wj; wi = wj; } else stop = true;
                    }
                    if (isI && wi == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 13;
                    ri = (rq - (total - wi)) / wi;
                }
                {
                    const double *tr = k_trans + ssi * S;
                    const double rq = k_cumul[ssi] * ri;
                    double total = tr[0], wi = total;
                    bool stop = false;
                    for (int j = 1; j < S; ++j) {
                        const double wj = tr[j];
                        if (!stop && total < rq) { tsi = j; total += wj; wi = wj; } else stop = true;
                    }
                    if (isI && wi == 0.0 && err == 0) err = G_ERR_ZERO_WEIGHT + 256 * 14;
                }
                if (isI && err == 0) {
                    if (rl == 0) {
                        R_SUS(ssi) -= 1;
                        R_SUS(tsi) += 1;
                        R_IMS(ssi) = (double)R_SUS(ssi) * k_cumul[ssi];
                        R_IMS(tsi) = (double)R_SUS(tsi) * k_cumul[tsi];
                    }
                    GBUMP(GC_I);
                    e_type = GEV_SUSCCHANGE; e_hap = ssi; e_pop = pi; e_nh = tsi; e_np = 0;
                    u_pop = pi; u_imm = true;
                }
                WSYNC();
            }

            GPROF(6);
            // ---- an infectious host's event: haplotype, then event class (pyx:496-511) ----
            if (__ballot(isN)) {
                rn = (choose - imm_pi) / inf_pi;
                const double r2 = inf_pi * rn;
                const double mult = k_mult[pi];
                const int n_sel = isN ? s_nocc[pi] : 0;
                const int32_t *lh = lhap + (int64_t)pi * cap;
                const int32_t *lc = lcls + (int64_t)pi * cap;
                int64_t *ln = lcnt + (int64_t)pi * cap;
                int64_t *lt = ltsum + (int64_t)pi * capT;
                const int maxn = rows_max(n_sel);
                const int last = max(n_sel - 1, 0);
                int k_hit = -1, hap_hit = 0, cls_hit = 0;
                double pre_hit = 0.0, w_hit = 0.0;
                int64_t cnt_hit = 0;
                if (__builtin_expect(maxn <= 16, 1)) {
                    // one chunk: the list in registers
                    const int k = min(rl, last);
                    const int64_t cn = ln[k];
                    const int hp = lh[k], cl = lc[k];
                    const double w = rl < n_sel ? tE_of(cl, mult) * (double)cn : 0.0;
                    double tot_;
                    const double pre = row_scan16(w, 0.0, tot_);
                    const int q = row_min(rl < n_sel && !(pre < r2) ? rl : 16);
                    const int qq = q < 16 ? q : (last & 15);
                    pre_hit = q < 16 ? rowget_f64(pre, qq) : tot_;     // no hit: the total of the whole list
                    w_hit = rowget_f64(w, qq);
                    hap_hit = rowget_i32(hp, qq); cls_hit = rowget_i32(cl, qq); cnt_hit = rowget_i64(cn, qq);
                    if (q < 16) k_hit = q;
                    else if (isN) {
                        // nothing reached r: the dense loop runs on to index H-1 (fc:26), a valid pick only if that haplotype
                        // is occupied, otherwise the reference reports a zero weight
                        if (n_sel > 0 && hap_hit == H - 1) k_hit = n_sel - 1; else err = G_ERR_ZERO_WEIGHT + 256 * 2;
                    }
                    if (isN) { ch_cn = cn; ch_cl = cl; ch_pi = pi; }
                } else {
                    // longer lists: the running sum advances one chunk of 16 entries per step; the chunk in which it first
                    // reaches r is then scanned entry by entry — same additions, same order
                    double carry = 0.0, carry_hit = 0.0;
                    int c_hit = -1;
                    for (int cb0 = 0; cb0 * 16 < maxn; cb0 += 4) {
                        double w4[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int k = (cb0 + c) * 16 + rl;
                            const int kk = min(k, last);
                            const int64_t cn = ln[kk];
                            const int cl = lc[kk];
                            w4[c] = k < n_sel ? tE_of(cl, mult) * (double)cn : 0.0;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const double acc = row_sum16(w4[c], carry);
                            if (c_hit < 0 && (cb0 + c) * 16 < n_sel && !(acc < r2)) { c_hit = cb0 + c; carry_hit = carry; }
                            carry = acc;
                        }
                        if (!__ballot(c_hit < 0 && (cb0 + 4) * 16 < n_sel)) break;
                    }
                    const int cc = c_hit >= 0 ? c_hit : (last >> 4);    // rows without a hit: their last chunk (H-1 rule)
                    const int k = cc * 16 + rl;
                    const int kk = min(k, last);
                    const int64_t cn = ln[kk];
                    const int hp = lh[kk], cl = lc[kk];
                    const double w = k < n_sel ? tE_of(cl, mult) * (double)cn : 0.0;
                    double tot_;
                    const double pre = row_scan16(w, c_hit >= 0 ? carry_hit : 0.0, tot_);
                    const int q = row_min(c_hit >= 0 && k < n_sel && !(pre < r2) ? rl : 16);
                    const int qq = q < 16 ? q : (last & 15);
                    pre_hit = q < 16 ? rowget_f64(pre, qq) : carry;
                    w_hit = rowget_f64(w, qq);
                    hap_hit = rowget_i32(hp, qq); cls_hit = rowget_i32(cl, qq); cnt_hit = rowget_i64(cn, qq);
                    if (q < 16) k_hit = cc * 16 + q;
                    else if (isN) {
                        if (n_sel > 0 && hap_hit == H - 1) k_hit = n_sel - 1; else err = G_ERR_ZERO_WEIGHT + 256 * 3;
                    }
                }
                const bool isN_ok = isN && err == 0;
                if (isN_ok && w_hit == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 4;
                rn = (r2 - (pre_hit - w_hit)) / w_hit;

                GPROF(7);
                // ---- event class by fastChoose over (birth, death, sampling, mutation) rates (pyx:503-511) ----
                const int cbh = k_bidx[cls_hit];
                const double e0 = R_BC(cbh), e1 = k_d[cls_hit], e2 = k_s[cls_hit] * mult, e3 = k_tm[cls_hit];
                int ei = 0;
                {
                    const double tEv = ((e0 + e1) + e2) + e3;
                    const double r3 = tEv * rn;
                    double total = e0, wi = e0;
                    if (total < r3) { ei = 1; total += e1; wi = e1; }
                    if (ei == 1 && total < r3) { ei = 2; total += e2; wi = e2; }
                    if (ei == 2 && total < r3) { ei = 3; total += e3; wi = e3; }
                    if (isN && err == 0 && wi == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 5;
                    rn = (r3 - (total - wi)) / wi;
                }
                const bool go = isN && err == 0;
                const bool isB = go && ei == 0, isD = go && (ei == 1 || ei == 2), isM = go && ei == 3;
                if (__ballot(isB)) {
                    // ---- Birth (pyx:568-605): susceptibility group by fastChoose over susceptHapPopRate[pi, hi, :] ----
                    const double *sg = k_sig + cbh * S;
                    double ws = 0.0;
                    for (int sn = 0; sn < S; ++sn) ws += (double)R_SNP(sn) * sg[sn];
                    int si = 0;
                    double rnb = 0.0;       // (RECOMB) the random number after the group choice (fast_choose.pxi:31)
                    {
                        const double rq = ws * rn;
                        double total = (double)R_SNP(0) * sg[0], wi = total;
                        bool stop = false;
                        for (int j = 1; j < S; ++j) {
                            const double wj = (double)R_SNP(j) * sg[j];
                            if (!stop && total < rq) { si = j; total += wj; wi = wj; } else stop = true;
                        }
                        if (isB && wi == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 6;
                        if (RECOMB) rnb = (rq - (total - wi)) / wi;
                    }
                    bool isR = false;       // a recombinant birth: second parent hi2, the newborn carries nhi
                    int hi2 = 0, nhi = 0;
                    if (RECOMB) {
                        isR = isB && err == 0 && rnb < p.recombination && s_ti[pi] > 1;
                        if (__builtin_expect(__ballot(isR) != 0, 0)) {
                            // ---- pyx:575-596: the second parent by fastChoose over birthInf[hn] = eventHapPopRate[pi, hn, 0] *
                            // infectious[pi, hn], one host of hi set aside; over the list in its (haplotype) order: the same additions ----
                            double rr_ = rnb / p.recombination;
                            auto weight = [&](int k) -> double {
                                const int kk = min(k, last);
                                return (isR && k < n_sel) ? R_BC(k_bidx[lc[kk]]) * (double)(ln[kk] - (kk == k_hit ? 1 : 0)) : 0.0;
                            };
                            const int maxr = rows_max(isR ? n_sel : 0);
                            double hs = 0.0;
                            for (int cb0 = 0; cb0 * 16 < maxr; ++cb0) hs = row_sum16(weight(cb0 * 16 + rl), hs);
                            const double r_ = hs * rr_;
                            if (isR && !(0.0 < r_) && !(n_sel > 0 && lh[0] == 0)) err = G_ERR_ZERO_WEIGHT + 256 * 15;   // the dense loop stops at haplotype 0
                            double carry = 0.0, tot2 = 0.0, w2 = 0.0;
                            int kq = -1;
                            for (int cb0 = 0; cb0 * 16 < maxr; ++cb0) {
                                const int k = cb0 * 16 + rl;
                                const double w = weight(k);
                                double tot_;
                                const double pre = row_scan16(w, carry, tot_);
                                const int q = row_min(kq < 0 && isR && k < n_sel && !(pre < r_) ? rl : 16);
                                if (kq < 0 && q < 16) { kq = cb0 * 16 + q; tot2 = rowget_f64(pre, q); w2 = rowget_f64(w, q); }
                                carry = tot_;
                            }
                            if (isR && err == 0 && kq < 0) {     // rounding left the total below r: the dense loop ends at haplotype H - 1
                                if (n_sel > 0 && lh[last] == H - 1) { kq = last; w2 = rowget_f64(weight((last & ~15) + rl), last & 15); tot2 = carry; }
                                else err = G_ERR_ZERO_WEIGHT + 256 * 16;
                            }
                            if (kq < 0) kq = 0;
                            if (isR && err == 0 && w2 == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 17;
                            rr_ = (r_ - (tot2 - w2)) / w2;
                            hi2 = isR ? lh[min(kq, last)] : 0;
                            const int64_t posRecomb = (int64_t)((double)p.genome_length * rr_);
                            // pyx:586-591 as written: the recombinant carries only the last site of one parent
                            nhi = sites > 0 ? ((p.sitesPosition[sites - 1] < posRecomb ? hap_hit : hi2) % 4) : 0;
                            if (isR && err == 0 && live) {
                                if (r.rec) {
                                    if (rec_n < r.rec_cap) {
                                        if (rl == 0) {
                                            int64_t *o = r.rec + (rep * r.rec_cap + rec_n) * 5;
                                            o[0] = ev_ptr; o[1] = hap_hit; o[2] = hi2; o[3] = nhi; o[4] = posRecomb;
                                        }
                                    } else err = G_ERR_CAPACITY;
                                }
                                rec_n += 1;
                            }
                            isR = isR && err == 0;
                        }
                    }
                    WSYNC();
                    if (isB && err == 0) {
                        if (rl == 0) {     // NewInfections (pyx:246-251) + immuneSourcePopRate[pi, si] (pyx:602)
                            R_SUS(si) -= 1;
                            R_IMS(si) = k_cumul[si] * (double)R_SUS(si);
                            s_ts[pi] -= 1; s_ti[pi] += 1;
                            if (live && !isR) { ln[k_hit] = cnt_hit + 1; if (n_sel > 64) lt[k_hit >> 6] += 1; }
                        }
                        if (isR) { op_n = 1; op_pi = pi; op_h0 = nhi; op_d0 = +1; ch_pi = -1; }     // the newborn's haplotype: a list operation
                        else if (rl == k_hit) ch_cn += 1;
                        gI += 1; GBUMP(GC_B);
                        e_type = GEV_BIRTH; e_hap = hap_hit; e_pop = pi; e_nh = si; e_np = isR ? hi2 : H;
                        u_pop = pi; u_inf = true; u_imm = true; u_mig = true;
                    }
                }
                if (__ballot(isD)) {
                    // ---- Death / Sampling (pyx:616-635): recovery into the class's group suscType ----
                    const int sty = k_stype[cls_hit];
                    WSYNC();
                    if (isD) {
                        if (rl == 0) {
                            R_SUS(sty) += 1;
                            R_IMS(sty) = (double)R_SUS(sty) * k_cumul[sty];
                            s_ts[pi] += 1; s_ti[pi] -= 1;
                        }
                        gI -= 1;
                        if (ei == 2) { cS += 1; e_type = GEV_SAMPLING; } else { GBUMP(GC_D); e_type = GEV_DEATH; }
                        if (cnt_hit == 1) { op_n = 1; op_pi = pi; op_h0 = hap_hit; op_d0 = -1; ch_pi = -1; }
                        else {
                            if (live && rl == 0) { ln[k_hit] = cnt_hit - 1; if (n_sel > 64) lt[k_hit >> 6] -= 1; }
                            if (rl == k_hit) ch_cn -= 1;
                        }
                        e_hap = hap_hit; e_pop = pi; e_nh = sty; e_np = 0;
                        u_pop = pi; u_inf = true; u_imm = true; u_mig = true;
                    }
                }
                if (__builtin_expect(__ballot(isM) != 0, 0)) {
                    // ---- Mutation (pyx:640-667): site by mRate[h, :], derived state by hapMutType[h, site, :] ----
                    const double *mr = p.mRate + (int64_t)hap_hit * sites;
                    int mi = 0;
                    {
                        const double rq = e3 * rn;
                        double total = isM ? mr[0] : 1.0, wi = total;
                        bool stop = false;
                        for (int i = 1; i < sites; ++i) {
                            if (isM && !stop && total < rq) { mi = i; wi = mr[i]; total += wi; } else stop = true;
                        }
                        if (isM && wi == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 7;
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
                        if (wi == 0.0) err = G_ERR_ZERO_WEIGHT + 256 * 8;
                    }
                    if (isM && err == 0) {
                        const int digit4 = 1 << (2 * (sites - mi - 1));     // Mutate (pyx:2420-2427)
                        const int AS = (hap_hit / digit4) % 4;
                        if (DS >= AS) DS += 1;
                        const int nhi = hap_hit + (DS - AS) * digit4;
                        op_n = 2; op_pi = pi; op_h0 = nhi; op_d0 = +1; op_h1 = hap_hit; ch_pi = -1;
                        GBUMP(GC_M);
                        e_type = GEV_MUTATION; e_hap = hap_hit; e_pop = pi; e_nh = nhi; e_np = 0;
                        u_pop = pi; u_inf = true;
                    }
                }
                WSYNC();
            }

            GPROF(8);
            // ---- migration, second half (pyx:679-692): haplotype of the source, group of the target, thinning ----
            if (__builtin_expect(__ballot(evm) != 0, 0)) {
                int hi = 0;
                {   // fastChoose(infectious[spi], totalInfectious[spi], rn): int64 weights over the occupancy list (order-free)
                    const int n = (evm && err == 0) ? s_nocc[spi] : 0;
                    const int32_t *lh2 = lhap + (int64_t)spi * cap;
                    const int64_t *ln2 = lcnt + (int64_t)spi * cap;
                    const int64_t *lt2 = ltsum + (int64_t)spi * capT;
                    const double rr_ = (double)s_ti[spi] * rm;
                    int64_t before = 0;
                    int base = 0;
                    bool none = false;
                    const int maxn2 = rows_max(n);
                    if (__builtin_expect(maxn2 > 64, 0)) {
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
                            const int64_t w = in ? ln2[k] : 0;
                            const int64_t pre = row_iscan(w) + carry;
                            const int q = row_min(kq < 0 && in && !((double)pre < rr_) ? rl : 16);
                            if (kq < 0 && q < 16) { kq = k - rl + q; total = rowget_i64(pre, q); wi = rowget_i64(w, q); }
                            carry = rowget_i64(pre, 15);
                            if (!__ballot(evm && kq < 0 && !none && base + (c4 + 1) * 16 < n)) break;
                        }
                        if (kq < 0) total = carry;
                    }
                    if (evm && err == 0 && kq < 0) {
                        if (n > 0 && lh2[n - 1] == H - 1) { kq = n - 1; wi = ln2[n - 1]; }
                        else { err = G_ERR_ZERO_WEIGHT + 256 * 11; kq = 0; wi = 1; }
                    }
                    if (kq < 0) { kq = 0; wi = 1; }
                    rm = (rr_ - (double)(total - wi)) / (double)wi;
                    hi = (evm && n > 0) ? lh2[kq] : 0;
                }
                int si = 0;
                {   // fastChoose(susceptible[tpi, :], totalSusceptible[tpi], rn): int64 weights (fc:18-31)
                    const double rq = (double)s_ts[tpi] * rm;
                    int64_t total = R_SUS(0), wi = total;
                    bool stop = false;
                    for (int j = 1; j < S; ++j) {
                        const int64_t wj = R_SUS(j);
                        if (!stop && (double)total < rq) { si = j; total += wj; wi = wj; } else stop = true;
                    }
                    if (evm && wi == 0 && err == 0) err = G_ERR_ZERO_WEIGHT + 256 * 12;
                    rm = (rq - (double)(total - wi)) / (double)wi;
                }
                const bool mgo = evm && err == 0;
                WSYNC();
                if (mgo) {
                    const double em = eff_private ? effP[(int64_t)spi * P + tpi] : qa.effMig0[(int64_t)spi * P + tpi];
                    const double p_accept = em * p.bRate[hi] * p.susc[(int64_t)hi * S + si] / s_mebm[tpi];
                    if (rm < p_accept) {
                        if (rl == 0) { R_SUS(si) -= 1; s_ts[tpi] -= 1; s_ti[tpi] += 1; }     // NewInfections (pyx:246-251)
                        gI += 1; GBUMP(GC_MIGP);
                        op_n = 1; op_pi = tpi; op_h0 = hi; op_d0 = +1;
                        e_type = GEV_MIGRATION; e_hap = hi; e_pop = spi; e_nh = si; e_np = tpi;
                        u_pop = tpi; u_inf = true; u_imm = true; u_mig = true;
                    } else {
                        GBUMP(GC_MIGN);
                    }
                }
                WSYNC();
            }
            WSYNC();
        }

        GPROF(9);
        // ================= deferred list operations: infectious[op_pi, hap] += delta, list kept ordered =================
        if (err != 0) op_n = 0;
        for (int oi = 0; oi < 2; ++oi) {
            const bool act = live && oi < op_n;
            if (!__ballot(act)) break;
            const int hap = oi == 0 ? op_h0 : op_h1;
            const int delta = oi == 0 ? op_d0 : -1;
            const int n = act ? s_nocc[op_pi] : 0;
            int32_t *lh = lhap + (int64_t)op_pi * cap;
            int32_t *lc = lcls + (int64_t)op_pi * cap;
            int64_t *ln = lcnt + (int64_t)op_pi * cap;
            int64_t *lt = ltsum + (int64_t)op_pi * capT;
            // ---- lower bound: first index whose haplotype is >= hap (16-ary descent over the sorted list) ----
            int posn = 0;
            bool found = false;
            int64_t cur = 0;
            {
                int lo = 0;
                const int maxn = rows_max(n);
                for (int stride = 1 << 20; stride >= 1; stride >>= 4) {
                    if (stride >= 16 && maxn <= stride) continue;
                    const int k = lo + rl * stride;
                    const int h = (act && k < n) ? lh[k] : 0x7fffffff;
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
            if (ins && n >= cap) { err = G_ERR_CAPACITY; }
            const bool ins_ok = ins && err == 0;
            const int ins_cls = ins_ok ? p.cls[hap] : 0;
            if (bump && rl == 0) { ln[posn] = cur + delta; if (n > 64) lt[posn >> 6] += delta; }
            // ---- tile sums of lists longer than one tile ----
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
            if (__ballot(ins_ok)) {
                enum { SU = 4 };
                int hi_ = ins_ok ? n : 0;
                const int lo_ = ins_ok ? posn : 0;
                while (__ballot(hi_ > lo_)) {
                    const int blo = max(lo_, hi_ - SU * 16);
                    int h[SU], cl[SU];
                    int64_t ct[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = blo + u * 16 + rl;
                        h[u] = 0; ct[u] = 0; cl[u] = 0;
                        if (k < hi_) { h[u] = lh[k]; cl[u] = lc[k]; ct[u] = ln[k]; }
                    }
                    WSYNC();
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = blo + u * 16 + rl;
                        if (k < hi_) { lh[k + 1] = h[u]; lc[k + 1] = cl[u]; ln[k + 1] = ct[u]; }
                    }
                    WSYNC();
                    hi_ = blo;
                }
                if (ins_ok && rl == 0) { lh[posn] = hap; lc[posn] = ins_cls; ln[posn] = delta; s_nocc[op_pi] = n + 1; }
                WSYNC();
                if (ins_ok && n == 64) {   // the list outgrows one tile: start its tile sums
                    int64_t s0 = 0;
                    for (int c4 = 0; c4 < 4; ++c4) s0 += rowget_i64(row_iscan(ln[c4 * 16 + rl]), 15);
                    if (rl == 0) { lt[0] = s0; lt[1] = ln[64]; }
                }
                WSYNC();
            }
            if (__ballot(rem)) {
                enum { SU = 4 };
                int lo_ = rem ? posn + 1 : 0;
                const int hi_ = rem ? n : 0;
                while (__ballot(lo_ < hi_)) {
                    int h[SU], cl[SU];
                    int64_t ct[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = lo_ + u * 16 + rl;
                        h[u] = 0; ct[u] = 0; cl[u] = 0;
                        if (k < hi_) { h[u] = lh[k]; cl[u] = lc[k]; ct[u] = ln[k]; }
                    }
                    WSYNC();
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int k = lo_ + u * 16 + rl;
                        if (k < hi_) { lh[k - 1] = h[u]; lc[k - 1] = cl[u]; ln[k - 1] = ct[u]; }
                    }
                    WSYNC();
                    lo_ += SU * 16;
                }
                if (rem && rl == 0) s_nocc[op_pi] = n - 1;
                WSYNC();
            }
        }

        GPROF(10);
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
                    err = G_ERR_CAPACITY;
                }
            }
            ev_ptr += 1;
        }

        GPROF(11);
        // ================= CheckLockdown for every population (PrepareParameters pyx:449-450, Restart pyx:736-737) =================
        // UpdateAllRates is a pure function of the state: one rebuild after the last switch leaves what the reference's call
        // after every switch leaves.
        if (__builtin_expect(__ballot(rebuild && ld_any) != 0, 0)) {
            bool cand = false;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int pn = s * 16 + rl;
                if (rebuild && pn < P) {
                    const double ti = (double)s_ti[pn];
                    const int lk = s_lock[pn];
                    if ((ti > k_thS[pn] && lk == 0) || (ti < k_thE[pn])) cand = true;
                }
            }
            if (__ballot(cand)) {
                const bool rc_ = rebuild && row_any(cand);
                for (int pn = 0; pn < P; ++pn) {
                    for (int pass = 0; pass < 2; ++pass) {
                        const double ti = (double)s_ti[pn];
                        const int lk = s_lock[pn];
                        const bool flip = rc_ && err == 0 && (pass == 0 ? (ti > k_thS[pn] && lk == 0) : (ti < k_thE[pn] && lk == 1));
                        if (!__ballot(flip)) continue;
                        WSYNC();
                        if (flip) {
                            if (rl == 0) {
                                s_cd[pn] = pass == 0 ? p.cdAfter[pn] : p.cdBefore[pn];
                                s_lock[pn] = pass == 0 ? 1 : 0;
                                if (loc_n < loc_cap && live) {
                                    loc_rec[loc_n * 2 + 0] = pass == 0 ? 1 : 0;
                                    loc_rec[loc_n * 2 + 1] = pn;
                                    loc_time[loc_n] = t_now;
                                    loc_iter[loc_n] = (int64_t)att << 40;
                                }
                            }
                            if (loc_n >= loc_cap) err = G_ERR_CAPACITY;
                            GBUMP(GC_SWAP);
                            loc_n += 1;
                            eff_dirty = true;
                        }
                        WSYNC();
                    }
                }
            }
        }

        // ================= effectiveMigration, maxEffectiveBirthMigration of a replicate whose contact densities changed
        // (pyx:327-338, 346-348): lane <-> target population, serial over the sources and the inner sum =================
        if (__builtin_expect(__ballot(full && eff_dirty && err == 0) != 0, 0)) {
            const bool on = full && eff_dirty && err == 0;
            bool anym = false;
            for (int s = 0; s < NS; ++s) {
                const int pn2 = s * 16 + rl;
                double mx = 0.0;
                if (on && pn2 < P) {
                    const double *m2 = p.mig + (int64_t)pn2 * P;
                    for (int pn1 = 0; pn1 < P; ++pn1) {
                        if (pn1 == pn2) continue;
                        const double *m1 = p.mig + (int64_t)pn1 * P;
                        double e = 0.0;
                        for (int pn3 = 0; pn3 < P; ++pn3) e += m1[pn3] * m2[pn3] * s_cd[pn3] / k_as[pn3];
                        if (live) effP[(int64_t)pn1 * P + pn2] = e;
                        if (e > mx) mx = e;
                    }
                    s_mebm[pn2] = mx * p.maxEffectiveBirth;
                    if (mx * p.maxEffectiveBirth != 0.0) anym = true;
                }
            }
            const bool ra = row_any(anym);
            if (on) { has_mig = ra; eff_private = true; eff_dirty = false; }
            WSYNC();
        }

        GPROF(12);
        // ================= UpdateRates(u_pop, ...) (pyx:516-546) / UpdateAllRates (pyx:279-351) =================
        if (err != 0) { u_pop = -1; }
        const bool want = err == 0 && (full || u_pop >= 0);
        const int maxu = rows_max(err != 0 ? 0 : (full ? P : (u_pop >= 0 ? 1 : 0)));
        for (int us = 0; us < maxu; ++us) {
            const bool act = want && (full ? us < P : us == 0);
            const int pu = act ? (full ? us : u_pop) : 0;
            const bool d_inf = act && (full || u_inf), d_imm = act && (full || u_imm);
            if (__ballot(full && act)) {
                rec_load(pu, full && act);
                if (full && act && rl < S) R_IMS(rl) = k_cumul[rl] * (double)R_SUS(rl);    // pyx:320-322
                WSYNC();
            }
            if (__ballot(d_inf)) {
                // ---- BirthRate per birth class (pyx:382-392) as the program of chain segments ----
                if (d_inf && rl < S) R_SNP(rl) = R_SUS(rl);      // susceptHapPopRate of this update = these counts x sigma
                WSYNC();
                const double *mrow = p.mig + (int64_t)pu * P;
                double mm[NS], cdv[NS], asv[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int pn = s * 16 + rl;
                    mm[s] = mrow[min(pn, P - 1)];
                    cdv[s] = s_cd[pn];      // lanes beyond P: cd = +0.0, as = 1.0
                    asv[s] = k_as[pn];
                }
                for (int sg = 0; sg < NSEG; ++sg) {
                    const int par = k_segpar[sg];
                    const double x = (double)R_SUS(k_segsn[sg]) * k_segsig[sg];
                    double ps = par < 0 ? 0.0 : s_seg[par];
                    if (__ballot(d_inf && x != 0.0)) {      // (a zero x adds +0.0 P times)
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            const int pn = s * 16 + rl;
                            double tv = x * mm[s] * mm[s] * cdv[s] / asv[s];
                            if (pn >= P) tv = 0.0;
                            if (s * 16 < P) ps = row_sum16(tv, ps);
                        }
                    }
                    if (rl == 0) s_seg[sg] = ps;
                    WSYNC();
                }
                if (d_inf && rl < CB) {
                    const int sgi = k_cbseg[rl];
                    R_BC(rl) = k_cbb[rl] * (sgi < 0 ? 0.0 : s_seg[sgi]);
                }
                WSYNC();
                GPROF(13);
                // ---- infectPopRate[pu]: tEvent * infectious over the occupied haplotypes, in haplotype order (pyx:519-528) ----
                const int n = d_inf ? s_nocc[pu] : 0;
                const int32_t *lc = lcls + (int64_t)pu * cap;
                const int64_t *ln = lcnt + (int64_t)pu * cap;
                const int maxn = rows_max(n);
                const double mult = k_mult[pu];
                double acc = 0.0;
                if (__builtin_expect(maxn <= 16, 1)) {
                    const bool chave = d_inf && pu == ch_pi;       // the list read for the haplotype choice, event applied
                    int64_t cn = ch_cn;
                    int cl = ch_cl;
                    if (__ballot(d_inf && !chave)) {
                        const int k = min(rl, max(n - 1, 0));
                        const int64_t cn2 = ln[k];
                        const int cl2 = lc[k];
                        if (!chave) { cn = cn2; cl = cl2; }
                    }
                    acc = row_sum16(rl < n ? tE_of(cl, mult) * (double)cn : 0.0, 0.0);
                } else {
                    const int last = max(n - 1, 0);
                    for (int cb0 = 0; cb0 * 16 < maxn; cb0 += 4) {
                        double w4[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int k = (cb0 + c) * 16 + rl;
                            const int kk = min(k, last);
                            const int64_t cn = ln[kk];
                            const int cl = lc[kk];
                            w4[c] = k < n ? tE_of(cl, mult) * (double)cn : 0.0;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if ((cb0 + c) * 16 < maxn) acc = row_sum16(w4[c], acc);
                    }
                }
                if (d_inf && rl == 0) s_inf[pu] = acc;
            }
            if (__ballot(d_imm)) {
                double v = 0.0;
                for (int sn = 0; sn < S; ++sn) v += R_IMS(sn);        // pyx:530-533
                if (d_imm && rl == 0) s_imm[pu] = v;
            }
            WSYNC();
            rec_store(pu, act);
        }
        if (maxu > 0) {
            // totalRate and the serial prefix sums of popRate at the slot ends (pyx:537-539)
            {
                double carry = 0.0, cend[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double w = S_POP(s * 16 + rl);      // lanes beyond P hold +0.0
                    if (s * 16 < P) carry = row_sum16(w, carry);
                    cend[s] = carry;
                }
                if (want) {
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (rl == s) s_cc[s] = cend[s];
                    totalRate = carry;
                }
                WSYNC();
            }
            // totalMigrationRate = sum of maxEffectiveBirthMigration * totalSusceptible * (globalInfectious - totalInfectious)
            if (__ballot(want && (full || u_mig) && has_mig)) {
                double acc = 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int pn = s * 16 + rl;
                    double w = s_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]);
                    if (pn >= P) w = 0.0;
                    if (s * 16 < P) acc = row_sum16(w, acc);
                }
                if (want && (full || u_mig)) totalMig = has_mig ? acc : 0.0;
            } else if (want && (full || u_mig)) {
                totalMig = 0.0;
            }
        }

        GPROF(14);
        // ================= after the pass =================
        if (full && st != GS_DONE) st = err ? GS_DONE : GS_RUN;
        if (err != 0) st = GS_DONE;
        if (ev && st == GS_RUN && (totalRate == 0.0 || gI == 0)) end_attempt = true;   // pyx:410-411
        // CheckLockdown(pi) (pyx:412, 698-710)
        if (__builtin_expect(__ballot(ev && st == GS_RUN && !end_attempt && ld_any && lk_pop >= 0) != 0, 0)) {
            const bool chk = ev && st == GS_RUN && !end_attempt && ld_any && lk_pop >= 0;
            const int pn = chk ? lk_pop : 0;
            for (int pass = 0; pass < 2; ++pass) {
                const double ti = (double)s_ti[pn];
                const int lk = s_lock[pn];
                const bool flip = chk && err == 0 && (pass == 0 ? (ti > k_thS[pn] && lk == 0) : (ti < k_thE[pn] && lk == 1));
                if (__builtin_expect(!__ballot(flip), 1)) continue;
                WSYNC();
                if (flip) {
                    if (rl == 0) {
                        s_cd[pn] = pass == 0 ? p.cdAfter[pn] : p.cdBefore[pn];
                        s_lock[pn] = pass == 0 ? 1 : 0;
                        if (loc_n < loc_cap && live) {
                            loc_rec[loc_n * 2 + 0] = pass == 0 ? 1 : 0;
                            loc_rec[loc_n * 2 + 1] = pn;
                            loc_time[loc_n] = t_now;
                            loc_iter[loc_n] = ((int64_t)att << 40) | att_loops;
                        }
                    }
                    if (loc_n >= loc_cap) { err = G_ERR_CAPACITY; st = GS_DONE; }
                    GBUMP(GC_SWAP);
                    loc_n += 1;
                    eff_dirty = true;
                    if (st == GS_RUN) st = GS_FULL;      // UpdateAllRates in the next pass (pyx:703, 709)
                }
                WSYNC();
            }
        }
        if ((st == GS_RUN || st == GS_FULL) && end_attempt) {
            // end of an attempt (pyx:414-418)
            open = false;
            if (ev_ptr <= 100 && a.iterations > 100) {
                // the lockdown records of this failed attempt stay in the log (pyx:714-738): keep what the host clock needs
                if (__builtin_expect(__ballot(loc_n > att_loc0 && a.record_events && r.fa_cap > 0) != 0, 0)) {
                    if (loc_n > att_loc0 && a.record_events && r.fa_cap > 0 && live) {
                        const int64_t n = ev_ptr - att_ev0;
                        for (int64_t k = rl; k < n; k += 16) {
                            const int64_t slot = att_ev0 + k - r.ev_base;
                            if (fa_n + k < r.fa_cap && slot >= 0 && slot < r.evcap) {
                                r.fa_rate[rep * r.fa_cap + fa_n + k] = r.ev_rate[rep * r.evcap + slot];
                                r.fa_key[rep * r.fa_cap + fa_n + k] =
                                    ((int64_t)att << 40) | (int64_t)(uint32_t)r.ev_cols[(rep * r.evcap + slot) * VGX_EV_COLS + 5];
                            }
                        }
                        fa_n += n;
                    }
                    WSYNC();
                }
                // Restart (pyx:714-738): compartments back to the initial snapshot, then CheckLockdown for all + UpdateAllRates
                ev_ptr = 0; cS = 0;
                if (rl < 8 && rl != GC_SWAP) s_cnt[rl] = 0;      // swapLockdown survives a Restart
                t_now = 0.0; traj_next = 0;
                att_ev0 = 0; att_loc0 = loc_n;
                restarts += 1; att += 1;
                st = GS_REBUILD;
            } else {
                good_attempt = (int64_t)att + 1;
                st = GS_DONE;
            }
        }
        if (__builtin_expect(__ballot(st == GS_REBUILD && restarts > 0 && !rebuild) != 0, 0)) {
            const bool rs = st == GS_REBUILD && restarts > 0 && !rebuild && live;
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
                            }
                        }
                        tsum += rowget_i64(row_iscan(ct), 15);
                    }
                    if (rs && rl == 0) ltsum[(int64_t)pn * capT + base / 64] = tsum;
                    ti += tsum;
                }
                if (rs)
                    for (int j = (n + 63) / 64 + rl; j <= n_old / 64 && j < capT; j += 16) ltsum[(int64_t)pn * capT + j] = 0;
                int64_t ts = 0;
                for (int sn = 0; sn < S; ++sn) ts += r.i_sus[pn * S + sn];
                if (rs) {
                    if (rl < S) cold[(int64_t)pn * W + rl] = r.i_sus[pn * S + rl];
                    if (rl == 0) { s_nocc[pn] = n; s_ts[pn] = ts; s_ti[pn] = ti; }
                }
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
                for (int s = 0; s < NS; ++s) {
                    const int pn = s * 16 + rl;
                    if (pn < P) { o[pn * 2 + 0] = (double)s_ti[pn]; o[pn * 2 + 1] = (double)s_ts[pn]; }
                }
                traj_next += 1;
            }
        }
    }

    // ---- state back to HBM ----
    WSYNC();
#ifdef VGX_PROFILE
    if (live && rl == 0 && r.prof)
        for (int i = 0; i < VGX_PROF_SLOTS; ++i) r.prof[rep * VGX_PROF_SLOTS + i] = prof_acc[i];
#endif
    if (live) {
        double *gD = r.popD + rep * PD_COUNT * P;
        int64_t *gI64 = r.popI + rep * PI_COUNT * P;
        int32_t *gN = r.nocc + rep * P;
        VgxRepScalars *sco = r.sc + rep;
        for (int s = 0; s < NS; ++s) {
            const int pn = s * 16 + rl;
            if (pn < P) {
                gD[PD_POPRATE * P + pn] = S_POP(pn);
                gD[PD_INFECT * P + pn] = s_inf[pn];
                gD[PD_IMMUNE * P + pn] = s_imm[pn];
                gD[PD_MIG * P + pn] = s_mebm[pn] * (double)s_ts[pn] * (double)(gI - s_ti[pn]);
                gD[PD_MAXEBM * P + pn] = s_mebm[pn];
                gD[PD_CD * P + pn] = s_cd[pn];
                gI64[PI_TOTSUS * P + pn] = s_ts[pn];
                gI64[PI_TOTINF * P + pn] = s_ti[pn];
                gI64[PI_LOCK * P + pn] = s_lock[pn];
                gN[pn] = s_nocc[pn];
            }
        }
        // cold records -> the arrays the other kernels and the host read
        for (int pn = 0; pn < P; ++pn) {
            const int64_t *cr = cold + (int64_t)pn * W;
            if (rl < S) {
                r.sus[(rep * P + pn) * S + rl] = cr[rl];
                r.immSrc[(rep * P + pn) * S + rl] = __longlong_as_double(cr[S + rl]);
            }
            if (rl < CB) r.birthC[(rep * P + pn) * CB + rl] = __longlong_as_double(cr[3 * S + rl]);
            for (int i = rl; i < CB * S; i += 16)
                r.xC[((rep * P + pn) * CB) * (int64_t)S + i] = (double)cr[2 * S + (i % S)] * p.cb_sigma[i];
        }
        if (!eff_private)
            for (int64_t i = rl; i < (int64_t)P * P; i += 16) effP[i] = qa.effMig0[i];
        if (rl == 0) {
            sco->currentTime = t_now; sco->totalRate = totalRate; sco->totalMig = totalMig;
            sco->globalInfectious = gI;
            sco->bCounter = s_cnt[GC_B]; sco->dCounter = s_cnt[GC_D]; sco->sCounter = cS; sco->mCounter = s_cnt[GC_M];
            sco->iCounter = s_cnt[GC_I]; sco->swapLockdown = s_cnt[GC_SWAP];
            sco->migPlus = s_cnt[GC_MIGP]; sco->migNonPlus = s_cnt[GC_MIGN];
            sco->good_attempt = good_attempt;
            sco->ev_ptr = ev_ptr; sco->loop_iterations = loops; sco->restarts = restarts;
            sco->loc_n = loc_n; sco->error = err; sco->traj_next = traj_next;
            sco->last_attempt = last_att; sco->last_attempt_loops = att_loops;
            sco->fa_n = fa_n;
            sco->rec_n = rec_n;
        }
    }
}

// one instantiation per number of 16-population slots (LDS and chain lengths follow the model's populations: 100 populations take seven
// slots — five wavefronts per CU and seven chain rows where eight slots gave four and eight), each with and without recombination
#define QUADG_KERNELS(NSLOT, PMAX)                                                                                                                  \
    extern "C" __global__ void __launch_bounds__(64, VGX_QUADG_WAVES) vgx_quadg_kernel_p##PMAX(VgxQuadgKArgs) { quadg_body<NSLOT>(); } \
    extern "C" __global__ void __launch_bounds__(64, VGX_QUADG_WAVES) vgx_quadg_kernel_p##PMAX##_rec(VgxQuadgKArgs) { quadg_body<NSLOT, true>(); }
QUADG_KERNELS(1, 16)
QUADG_KERNELS(2, 32)
QUADG_KERNELS(3, 48)
QUADG_KERNELS(4, 64)
QUADG_KERNELS(5, 80)
QUADG_KERNELS(6, 96)
QUADG_KERNELS(7, 112)
QUADG_KERNELS(8, 128)

// ---- host-side launcher ----
extern "C" __attribute__((visibility("hidden"))) hipError_t vgxi_launch_quadg(const VgxDirectArgs *a, const VgxQuadgArgs *qa,
                                                                             hipStream_t stream) {
    const int P = a->p.P;
    const int NS = (P + 15) / 16;
    const VgxQuadgLayout L = vgx_quadg_layout(16 * NS, a->p.S, a->p.C, a->p.CB, qa->nseg, qa->W);
    const bool rec = a->p.recombination != 0.0;
    typedef void (*Kern)(VgxQuadgKArgs);
    static const Kern plain[8] = {vgx_quadg_kernel_p16, vgx_quadg_kernel_p32, vgx_quadg_kernel_p48, vgx_quadg_kernel_p64,
                                  vgx_quadg_kernel_p80, vgx_quadg_kernel_p96, vgx_quadg_kernel_p112, vgx_quadg_kernel_p128};
    static const Kern with_rec[8] = {vgx_quadg_kernel_p16_rec, vgx_quadg_kernel_p32_rec, vgx_quadg_kernel_p48_rec, vgx_quadg_kernel_p64_rec,
                                     vgx_quadg_kernel_p80_rec, vgx_quadg_kernel_p96_rec, vgx_quadg_kernel_p112_rec, vgx_quadg_kernel_p128_rec};
    if (NS < 1 || NS > 8) return hipErrorInvalidValue;
    const Kern k = rec ? with_rec[NS - 1] : plain[NS - 1];
    hipError_t err = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (err != hipSuccess) return err;
    const unsigned grid = (unsigned)((a->n_replicates + 3) / 4);
    if (getenv("VGX_TIMING")) {   // diagnostics: how many of these wavefronts a CU holds (LDS and registers: a few hundred bytes decide)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k, 64, (size_t)L.total) == hipSuccess)
            fprintf(stderr, "vgx_quadg: %d populations (%d slots), LDS %d B per wavefront, %d wavefronts per CU\n", P, NS, L.total, nb);
    }
    VgxQuadgKArgs ka;
    ka.a = *a; ka.qa = *qa;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), (size_t)L.total, stream, ka);
    return hipGetLastError();
}
