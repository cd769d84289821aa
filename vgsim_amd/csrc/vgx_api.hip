// vgx_api.hip — host side of libvgx.so: the C ABI of include/vgx.h.
//
// Owns device memory (hipMalloc), one HIP stream and a pair of HIP events per engine; converts between
// the reference's dense host arrays (BirthDeathModel fields, src/_BirthDeath.pyx:47-68) and the engine's
// HBM layout (vgx_dev.h); does the parameter-only parts of UpdateAllRates (pyx:284-297, 340-344) and the
// first-call part of PrepareParameters (pyx:435-448) on the host, in the reference's operation order
// (this file is compiled with -ffp-contract=off as well); launches the kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/vgx.h"
#include "vgx_dev.h"
#include "vgx_quadg.h"
#include "vgx_taus.h"
#include "vgx_solo.h"
#include "vgx_lone.h"
#include "vgx_rng.h"

// launchers defined next to their kernels (vgx_direct.hip)
extern "C" hipError_t vgxi_launch_direct(const VgxDirectArgs *a, size_t lds, hipStream_t stream);
extern "C" int vgxi_tau_inc_shards(int64_t H, int64_t P);
extern "C" int64_t vgxi_tau_queue_shards(int64_t H, int64_t P);
extern "C" int64_t vgxi_tau_queue_shard_max(int64_t H);
extern "C" int vgxi_tau_drift_blocks(const VgxTauArgs *a);
extern "C" hipError_t vgxi_tau_sieve(const VgxTauArgs *a, hipStream_t s);
extern "C" hipError_t vgxi_launch_lanes(const VgxDirectArgs *a, const VgxLaneWs *w, hipStream_t stream);
extern "C" hipError_t vgxi_launch_quad(const VgxDirectArgs *a, const double *cd, double *effMig, double *maxEBM, int32_t *has_mig, int long_lists,
                                       hipStream_t stream);
extern "C" size_t vgxi_direct_lds_bytes(int P, int S, int C, int CB);
extern "C" hipError_t vgxi_launch_counts64(const int32_t *c32, int64_t *c64, int64_t n, hipStream_t stream);
extern "C" hipError_t vgxi_launch_quad_prep(const VgxDevParams *p, const double *cd, double *effMig, double *maxEBM, int32_t *has_mig, hipStream_t stream);
extern "C" hipError_t vgxi_launch_quadf(const VgxDirectArgs *a, const double *cd, double *effMig, double *maxEBM, int32_t *has_mig, hipStream_t stream);
extern "C" hipError_t vgxi_launch_taus(const VgxTausArgs *a, hipStream_t s);
extern "C" hipError_t vgxi_launch_quadg(const VgxDirectArgs *a, const VgxQuadgArgs *qa, hipStream_t stream);
extern "C" hipError_t vgxi_launch_solo(const VgxDirectArgs *a, const VgxSoloArgs *sa, int clock, hipStream_t stream);
extern "C" hipError_t vgxi_launch_lone(const VgxDirectArgs *a, const VgxLoneArgs *la, int clock, hipStream_t stream);
extern "C" hipError_t vgxi_launch_counts32(const int64_t *c64, int32_t *c32, int64_t n, hipStream_t stream);
extern "C" hipError_t vgxi_launch_init_reps(const VgxDevRep *r, int P, int S, int64_t R, const int32_t *s_nocc,
                                            const int32_t *s_hap, const int32_t *s_cls, const int64_t *s_cnt,
                                            int64_t s_cap, const int64_t *s_sus, const double *s_cd,
                                            const int64_t *s_tot, hipStream_t stream);

#define TAU_DECL(name) extern "C" hipError_t vgxi_##name(const VgxTauArgs *a, hipStream_t s);
TAU_DECL(tau_eff) TAU_DECL(tau_scatter) TAU_DECL(tau_prep) TAU_DECL(tau_drift) TAU_DECL(tau_choose) TAU_DECL(tau_draw)
TAU_DECL(tau_conv8) TAU_DECL(tau_sync8) TAU_DECL(tau_arrivals) TAU_DECL(tau_verdict) TAU_DECL(tau_apply) TAU_DECL(tau_check) TAU_DECL(tau_decide) TAU_DECL(tau_commit) TAU_DECL(tau_finish) TAU_DECL(tau_draw_big) TAU_DECL(tau_suspect)

static std::string g_create_error;

struct HostState {
    std::vector<int64_t> susceptible, infectious, initial_susceptible, initial_infectious;
    std::vector<int64_t> totalSusceptible, totalInfectious, lockdownON;
    std::vector<double> contactDensity;
    int64_t first_simulation = 0, globalInfectious = 0;
    int64_t bCounter = 0, dCounter = 0, sCounter = 0, mCounter = 0, iCounter = 0, swapLockdown = 0, migPlus = 0,
            migNonPlus = 0, good_attempt = 0;
    double currentTime = 0, totalRate = 0, totalMigrationRate = 0, tau_l = 0.01;
    int64_t ev_ptr = 0, ev_size = 0;
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct vgx_engine {
    vgx_dims d{};
    int64_t R = 1;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    bool have_params = false, have_state = false, dev_state_valid = false;
    bool call_philox = false;      // the last direct call drew from the counter-based stream (the host clock must too)
    int64_t start_max_nocc = 0;    // longest occupancy list of the state last uploaded
    int64_t start_lone_rows = 0;   // heap rows of vgx_lone.hip that state (and the Restart snapshot) needs at least
    void *pin[2] = {nullptr, nullptr};   // pinned staging buffers of large uploads (VGX_PIN_BYTES each), allocated on first use
    void *pin_tau = nullptr;              // pinned mirror of what the tau step loop reads after every try and step (flags, the finish kernel's record)
    size_t pin_tau_bytes = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool counts32_valid = false;   // r_lcnt32 mirrors r_lcnt (vgx_quad.hip keeps it; other kernels do not)
    bool counts64_valid = true;    // r_lcnt is current (vgx_quadf.hip and the long-list kernel of vgx_quad.hip keep the 4-byte counts only)
    int C = 0, CB = 0;
    // host copies of what the host needs again
    std::vector<int32_t> cls;
    std::vector<int64_t> sizes, seeds;
    std::vector<double> suscepCumul, mig, actualSizes;
    std::vector<char> h_class_pos;       // [C] the class has a positive recovery, sampling, mutation or transmission rate
    HostState hs;
    // device
    std::vector<DevBuf *> all;
    DevBuf p_cls, p_suscType, p_mRate, p_hapMutType, p_bRate, p_susc, p_cd, p_cs, p_ctm, p_cbidx, p_cstype, p_cbb, p_cbsig,
        p_sizes, p_cdBefore, p_cdAfter, p_startLD, p_endLD, p_sampMult, p_actualSizes, p_mig, p_suscTrans,
        p_suscCumul, p_sitesPos;
    double recombination = 0.0;          // pyx:93, 1422-1426
    int64_t genome_length = 1000000, rec_cap = 0;
    DevBuf r_rec;
    DevBuf r_popD, r_popI, r_sus, r_immSrc, r_birthC, r_xC, r_effMig, r_nocc, r_lhap, r_lcls, r_lcnt, r_lcnt32, r_ltsum, r_lanews, r_sc, r_seeds,
        r_evrate, r_evcols, r_locrec, r_loctime, r_lociter, r_farate, r_fakey, r_traj, r_prof, r_qeff, r_qmebm, r_qflag;
    // tau-leaping (dense compartments)
    DevBuf t_I, t_S, t_dChk, t_dApp, t_dSi, t_dTot, t_totInf, t_gI, t_cd, t_lock, t_F, t_eff, t_Aeff, t_Gout, t_dS,
        t_taubits, t_tau, t_time, t_flags, t_counters, t_cnttry, t_cntpop, t_front, t_frontn, t_occ, t_occn, t_occpop, t_tIpt, t_d8spk, t_d8sbc, t_d8stile, t_d8sovf, t_d8smax, t_mev, t_mevn, t_mevbase, t_locn, t_mutcum, t_migcdf, t_migIn, t_mutHi, t_colT, t_colTW, t_inc, t_incn, t_sieve, t_sievepop, t_sieveskip, t_big, t_bign, t_res, t_susp, t_suspn, t_stkey, t_stval, t_dChkTot, t_q, t_qn, t_hist, t_I8, t_dSpart, t_tmax8, t_iI, t_iS, t_slog, t_sres;
    std::vector<int64_t> tau_sieve_skipped;   // [R] tries left out by the sieve in the last tau call
    bool last_was_tau = false;
    bool tau_staged = false;              // vgx_stage_tau put the current start state on the device in the tau kernels' layout
    int64_t tau_occupied = 0;             // ... and counted its occupied compartments
    bool direct_logs_valid = false;   // the rate / iteration logs of the last direct call are on the device (host_clock can run)
    int64_t tau_mev_cap = 0;
    struct TauStep { double time; int64_t m0, m1; int32_t tries; };   // tries: rejected tries of the step's halving loop (pyx:2316-2321)
    std::vector<std::vector<TauStep>> tau_log;      // [R] MULTITYPE records of the last tau call
    std::vector<std::vector<double>> tau_loc_time;  // [R] lockdown log of the last tau call
    std::vector<std::vector<int64_t>> tau_loc_state, tau_loc_pop;
    std::vector<int64_t> tau_ev_ptr0;
    std::vector<VgxRepScalars> tau_sc;
    std::vector<double> h_startLD, h_endLD, h_cdBefore, h_cdAfter;
    bool h_has_mig = false, h_mut_uniform = false, h_mig_uniform = false;
    double h_mig_b = 0.0, h_mig_d = 1.0;
    double h_mutp[16][3] = {}, h_mut_total = 0.0;
    DevBuf i_nocc, i_hap, i_cls, i_cnt, i_sus;          // initial state (Restart)
    DevBuf s_nocc, s_hap, s_cls, s_cnt, s_sus, s_cd, s_tot;  // state at the start of the call
    VgxDevParams dp{};
    VgxDevRep dr{};
    int64_t cap = 0, evcap = 0, ev_base = 0, ev_ptr0 = 0, traj_points = 0;
    std::vector<int64_t> call_ev0;        // [R] events.ptr of every replicate at the start of the last call (they differ once replicates of a
                                          // continued ensemble stopped at different places)
    // host clock of the last direct call (host_clock below)
    int64_t loc_cap = 1, fa_cap = 0;
    bool call_recorded = false, call_has_tlimit = false;
    double call_tlimit = 0.0;
    std::vector<double> call_t0;          // [R] currentTime at the start of the call
    struct HostClock {
        int64_t rep = -1, e0 = 0;         // replicate; first event index of the reconstructed range
        bool exact = false;               // false: no rate log (record_events = 0), device clock reported
        std::vector<double> times;        // [ev_ptr - e0] event times
        double final_time = 0.0;          // currentTime after the call
        std::vector<double> loc_times;    // lockdown records
        bool limit_mismatch = false;      // device and host clock disagreed on a time-limit stop (see host_clock)
    } hc;
    int64_t clock_mismatches = 0;
    bool last_used_lanes = false, last_used_quad = false, last_used_quadg = false, last_used_quadf = false;
    // BirthRate program of the general row kernel (vgx_quadg.h)
    std::vector<int32_t> h_seg_par, h_seg_sn, h_cb_seg;
    std::vector<double> h_seg_sig;
    DevBuf q_segpar, q_segsn, q_segsig, q_cbseg, r_cold;
    // BirthRate segments of the single-trajectory kernel (vgx_solo.h): distinct (group, non-zero susceptibility) pairs by group
    std::vector<int32_t> h_so_sn, h_so_hapcls, h_so_nnz, h_so_tsn;
    std::vector<double> h_so_sig, h_so_tsig, h_so_clssig;
    int h_so_ncls = 0, h_so_maxnnz = 0;
    DevBuf so_sn, so_sig, so_rcp, so_hapcls, so_nnz, so_tsn, so_tsig, so_clssig, so_pass;
    std::vector<int32_t> h_so_pass;
    int h_so_npass0 = 0, h_so_npass1 = 0;
    bool last_used_solo = false;
    bool last_used_lone = false;
    int64_t lone_fallbacks = 0;       // calls that ran again on the row kernel because the LDS heap of vgx_lone.hip was full
    bool dev_clock_stale = false;     // the last direct call ran without the device clock (vgx_solo.hip, CLOCK = false): r_sc[].currentTime is the
                                      // time at that call's START; a continued call must take the host clock's final time instead
    int64_t last_ev_size = 0;
    std::vector<VgxRepScalars> sc_host;
    bool sc_host_valid = false;
    float last_ms = 0.f;
    int64_t last_launches = 0;
    size_t dev_bytes = 0;
};

#define HIPCHECK(e_, call)                                                                             \
    do {                                                                                               \
        hipError_t err__ = (call);                                                                     \
        if (err__ != hipSuccess) {                                                                     \
            (e_)->err = std::string(#call) + ": " + hipGetErrorString(err__);                          \
            return VGX_ERR_HIP;                                                                        \
        }                                                                                              \
    } while (0)

static int fail(vgx_engine *e, int code, const std::string &msg) {
    e->err = msg;
    return code;
}

// Host-side loops over all compartments of a large state (2^28 at BASELINE config 4): f(first, last, part) on up to 16 threads
// (n items of `weight` elementary operations each; small jobs stay on the calling thread)
template <class F>
static void for_parts(int64_t n, F f, int64_t weight = 1) {
    unsigned nt = (unsigned)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u), std::max<int64_t>(n, 1));
    if (n * weight < ((int64_t)1 << 22)) nt = 1;
    if (nt == 1) { f((int64_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    const int64_t step = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const int64_t b = std::min<int64_t>(n, (int64_t)t * step), en = std::min<int64_t>(n, b + step);
        th.emplace_back([=]() { f(b, en, t); });
    }
    for (auto &x : th) x.join();
}

static int ensure(vgx_engine *e, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (b.bytes >= bytes) return VGX_OK;
    if (b.p) {
        HIPCHECK(e, hipFree(b.p));
        e->dev_bytes -= b.bytes;
        b.p = nullptr;
        b.bytes = 0;
    }
    HIPCHECK(e, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    e->dev_bytes += bytes;
    if (std::find(e->all.begin(), e->all.end(), &b) == e->all.end()) e->all.push_back(&b);
    return VGX_OK;
}

template <typename T>
static int upload(vgx_engine *e, DevBuf &b, const T *src, size_t n) {
    int rc = ensure(e, b, n * sizeof(T));
    if (rc) return rc;
    if (n) HIPCHECK(e, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, e->stream));
    return VGX_OK;
}

extern "C" int vgx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int vgx_create(const vgx_dims *dims, int64_t n_replicates, int device, vgx_engine **out) {
    if (!dims || !out || n_replicates < 1) { g_create_error = "vgx_create: bad argument"; return VGX_ERR_ARG; }
    int64_t h = 1;
    for (int64_t s = 0; s < dims->sites; s++) h *= 4;
    if (dims->sites < 0 || dims->sites > 15 || dims->hapNum != h || dims->popNum < 1 || dims->susNum < 1) {
        g_create_error = "vgx_create: hapNum must equal 4^sites (sites <= 15), popNum >= 1, susNum >= 1";
        return VGX_ERR_ARG;
    }
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess || n == 0) {
        g_create_error = std::string("vgx_create: no HIP device available (") + hipGetErrorString(he) +
                         "); this engine has no CPU fallback";
        return VGX_ERR_HIP;
    }
    if (device < 0 || device >= n) { g_create_error = "vgx_create: device index out of range"; return VGX_ERR_ARG; }
    vgx_engine *e = new vgx_engine();
    e->d = *dims;
    e->R = n_replicates;
    e->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&e->stream) != hipSuccess ||
        hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
        g_create_error = "vgx_create: could not create stream/events";
        delete e;
        return VGX_ERR_HIP;
    }
    e->seeds.assign((size_t)n_replicates, 0);
    *out = e;
    return VGX_OK;
}

extern "C" void vgx_destroy(vgx_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (DevBuf *b : e->all)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < 2; i++) {
        if (e->pin[i]) (void)hipHostFree(e->pin[i]);
        if (e->pin_ev[i]) (void)hipEventDestroy(e->pin_ev[i]);
    }
    if (e->pin_tau) (void)hipHostFree(e->pin_tau);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" const char *vgx_last_error(const vgx_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }
extern "C" double vgx_last_kernel_ms(const vgx_engine *e) { return e ? (double)e->last_ms : 0.0; }
extern "C" int64_t vgx_last_kernel_launches(const vgx_engine *e) { return e ? e->last_launches : 0; }
extern "C" int vgx_last_direct_kernel(const vgx_engine *e) {
    if (!e) return 0;
    return e->last_used_lone ? 6 : e->last_used_solo ? 5 : e->last_used_quadg ? 4 : (e->last_used_quad || e->last_used_quadf) ? 3 : e->last_used_lanes ? 2 : 1;
}
extern "C" int64_t vgx_device_bytes(const vgx_engine *e) { return e ? (int64_t)e->dev_bytes : 0; }

extern "C" int vgx_set_seeds(vgx_engine *e, const int64_t *seeds) {
    if (!e || !seeds) return VGX_ERR_ARG;
    for (int64_t r = 0; r < e->R; r++) {
        if (seeds[r] < 0) return fail(e, VGX_ERR_ARG, "vgx_set_seeds: seeds must be >= 0");
        e->seeds[(size_t)r] = seeds[r];
    }
    return VGX_OK;
}

extern "C" int vgx_set_params(vgx_engine *e, const vgx_params *p) {
    if (!e || !p) return VGX_ERR_ARG;
    HIPCHECK(e, hipSetDevice(e->device));
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum, sites = e->d.sites;
    if (!p->bRate || !p->dRate || !p->sRate || !p->susceptibility || !p->suscType || !p->suscepTransition ||
        !p->sizes || !p->contactDensityBeforeLockdown || !p->contactDensityAfterLockdown || !p->startLD ||
        !p->endLD || !p->samplingMultiplier || !p->migrationRates || (sites > 0 && (!p->mRate || !p->hapMutType)))
        return fail(e, VGX_ERR_ARG, "vgx_set_params: null parameter array");
    for (int64_t h = 0; h < H; h++)
        if (p->suscType[h] < 0 || p->suscType[h] >= S) return fail(e, VGX_ERR_ARG, "vgx_set_params: suscType out of range");

    // ---- classes of identical per-haplotype rate rows ----
    std::vector<double> tm((size_t)H);
    for (int64_t h = 0; h < H; h++) {  // tmRate, pyx:306-308
        double t = 0;
        for (int64_t s = 0; s < sites; s++) t += p->mRate[h * sites + s];
        tm[(size_t)h] = t;
    }
    std::unordered_map<std::string, int> fullmap, birthmap;
    std::vector<double> c_d, c_s, c_tm, cb_b, cb_sig;
    std::vector<int32_t> c_bidx, c_stype;
    e->cls.assign((size_t)H, 0);
    std::string key;
    for (int64_t h = 0; h < H; h++) {
        key.assign((const char *)&p->bRate[h], 8);
        key.append((const char *)&p->susceptibility[h * S], (size_t)S * 8);
        auto bi = birthmap.find(key);
        int cb;
        if (bi == birthmap.end()) {
            cb = (int)cb_b.size();
            birthmap.emplace(key, cb);
            cb_b.push_back(p->bRate[h]);
            for (int64_t s = 0; s < S; s++) cb_sig.push_back(p->susceptibility[h * S + s]);
        } else {
            cb = bi->second;
        }
        key.append((const char *)&p->dRate[h], 8);
        key.append((const char *)&p->sRate[h], 8);
        key.append((const char *)&tm[(size_t)h], 8);
        key.append((const char *)&p->suscType[h], 8);
        auto fi = fullmap.find(key);
        int c;
        if (fi == fullmap.end()) {
            c = (int)c_d.size();
            fullmap.emplace(key, c);
            c_d.push_back(p->dRate[h]);
            c_s.push_back(p->sRate[h]);
            c_tm.push_back(tm[(size_t)h]);
            c_bidx.push_back(cb);
            c_stype.push_back((int32_t)p->suscType[h]);
            if (c_d.size() > VGX_MAX_CLASSES)
                return fail(e, VGX_ERR_CLASSES,
                            "vgx_set_params: more than " + std::to_string(VGX_MAX_CLASSES) +
                                " distinct per-haplotype rate rows (bRate, susceptibility, dRate, sRate, sum of mRate, suscType)");
        } else {
            c = fi->second;
        }
        e->cls[(size_t)h] = c;
    }
    e->C = (int)c_d.size();
    e->CB = (int)cb_b.size();
    {   // BirthRate (pyx:382-392) per birth class as a program of chain segments: the groups with a non-zero susceptibility in
        // order, common prefixes (group, susceptibility) of different classes shared (vgx_quadg.hip)
        e->h_seg_par.clear(); e->h_seg_sn.clear(); e->h_seg_sig.clear();
        e->h_cb_seg.assign(cb_b.size(), -1);
        std::unordered_map<std::string, int> segmap;
        for (size_t cb = 0; cb < cb_b.size(); cb++) {
            int cur = -1;
            for (int64_t sn = 0; sn < S; sn++) {
                const double sg = cb_sig[cb * (size_t)S + (size_t)sn];
                if (sg == 0.0) continue;
                std::string k((const char *)&cur, sizeof(cur));
                k.append((const char *)&sn, sizeof(sn));
                k.append((const char *)&sg, sizeof(sg));
                auto it = segmap.find(k);
                if (it == segmap.end()) {
                    const int id = (int)e->h_seg_par.size();
                    segmap.emplace(k, id);
                    e->h_seg_par.push_back(cur); e->h_seg_sn.push_back((int32_t)sn); e->h_seg_sig.push_back(sg);
                    cur = id;
                } else {
                    cur = it->second;
                }
            }
            e->h_cb_seg[cb] = cur;
        }
    }
    {   // the single-trajectory kernel runs that program two chains at a time (vgx_solo.h): a segment is ready when its parent is done,
        // the sum of the migration rates (-2) from the start
        e->h_so_pass.assign(4 * VGX_SOLO_MAX_PASS, -1);
        const int ns = (int)e->h_seg_par.size();
        for (int with_mig = 0; with_mig < 2; with_mig++) {
            std::vector<char> done((size_t)ns, 0);
            bool mig_done = !with_mig;
            int left = ns, np = 0;
            while ((left > 0 || !mig_done) && np < VGX_SOLO_MAX_PASS) {
                int pick[2] = {-1, -1}, n = 0;
                for (int sg = 0; sg < ns && n < 2; sg++)
                    if (!done[(size_t)sg] && (e->h_seg_par[(size_t)sg] < 0 || done[(size_t)e->h_seg_par[(size_t)sg]])) pick[n++] = sg;
                if (n < 2 && !mig_done) { pick[n++] = -2; mig_done = true; }
                for (int k = 0; k < 2; k++)
                    if (pick[k] >= 0) { done[(size_t)pick[k]] = 1; left--; }
                e->h_so_pass[(size_t)((2 * with_mig + 0) * VGX_SOLO_MAX_PASS + np)] = pick[0];
                e->h_so_pass[(size_t)((2 * with_mig + 1) * VGX_SOLO_MAX_PASS + np)] = pick[1];
                np++;
            }
            (with_mig ? e->h_so_npass1 : e->h_so_npass0) = (left > 0 || !mig_done) ? -1 : np;   // -1: does not fit
        }
    }
    {   // segments of the single-trajectory kernel: for every group the distinct non-zero susceptibility values, in group order
        e->h_so_sn.clear(); e->h_so_sig.clear();
        for (int64_t sn = 0; sn < S; sn++) {
            const size_t first = e->h_so_sn.size();
            for (size_t cb = 0; cb < cb_b.size(); cb++) {
                const double sg = cb_sig[cb * (size_t)S + (size_t)sn];
                if (sg == 0.0) continue;
                bool seen = false;
                for (size_t k = first; k < e->h_so_sn.size() && !seen; k++) seen = memcmp(&e->h_so_sig[k], &sg, 8) == 0;
                if (!seen) { e->h_so_sn.push_back((int32_t)sn); e->h_so_sig.push_back(sg); }
            }
        }
    }
    {   // susceptibility classes of the single-trajectory kernel's compact layout: haplotypes with identical susceptibility rows
        std::unordered_map<std::string, int> cmap;
        e->h_so_hapcls.assign((size_t)H, 0);
        e->h_so_nnz.assign(VGX_SOLO_ROWS, 0);
        e->h_so_tsn.assign(VGX_SOLO_ROWS * VGX_SOLO_MAX_S, 0);
        e->h_so_tsig.assign(VGX_SOLO_ROWS * VGX_SOLO_MAX_S, 0.0);
        e->h_so_clssig.assign(VGX_SOLO_ROWS * VGX_SOLO_MAX_S, 0.0);
        e->h_so_ncls = 0; e->h_so_maxnnz = 0;
        for (int64_t h = 0; h < H; h++) {
            std::string k((const char *)&p->susceptibility[h * S], (size_t)S * 8);
            auto it = cmap.find(k);
            int c;
            if (it == cmap.end()) {
                c = (int)cmap.size();
                cmap.emplace(k, c);
                if (c < VGX_SOLO_ROWS && S <= VGX_SOLO_MAX_S) {
                    int n = 0;
                    for (int64_t sn = 0; sn < S; sn++) {
                        const double sg = p->susceptibility[h * S + sn];
                        e->h_so_clssig[(size_t)(c * VGX_SOLO_MAX_S + sn)] = sg;
                        if (sg == 0.0) continue;
                        e->h_so_tsn[(size_t)(c * VGX_SOLO_MAX_S + n)] = (int32_t)sn;
                        e->h_so_tsig[(size_t)(c * VGX_SOLO_MAX_S + n)] = sg;
                        n++;
                    }
                    e->h_so_nnz[(size_t)c] = n;
                    e->h_so_maxnnz = std::max(e->h_so_maxnnz, n);
                }
            } else {
                c = it->second;
            }
            e->h_so_hapcls[(size_t)h] = c;
        }
        e->h_so_ncls = (int)cmap.size();
    }
    e->h_class_pos.assign(c_d.size(), 0);
    for (size_t c = 0; c < c_d.size(); c++)
        e->h_class_pos[c] = (c_d[c] > 0.0 || c_s[c] > 0.0 || c_tm[c] > 0.0 || cb_b[(size_t)c_bidx[c]] > 0.0) ? 1 : 0;

    // ---- parameter-only parts of UpdateAllRates, in the reference's order ----
    e->suscepCumul.assign((size_t)S, 0.0);
    for (int64_t s1 = 0; s1 < S; s1++) {  // pyx:284-287
        double v = 0;
        for (int64_t s2 = 0; s2 < S; s2++) v += p->suscepTransition[s1 * S + s2];
        e->suscepCumul[(size_t)s1] = v;
    }
    e->mig.assign(p->migrationRates, p->migrationRates + P * P);
    e->actualSizes.assign((size_t)P, 0.0);
    e->sizes.assign(p->sizes, p->sizes + P);
    e->h_startLD.assign(p->startLD, p->startLD + P);
    e->h_endLD.assign(p->endLD, p->endLD + P);
    e->h_cdBefore.assign(p->contactDensityBeforeLockdown, p->contactDensityBeforeLockdown + P);
    e->h_cdAfter.assign(p->contactDensityAfterLockdown, p->contactDensityAfterLockdown + P);
    e->h_has_mig = false;
    for (int64_t i = 0; i < P; i++)
        for (int64_t j = 0; j < P; j++)
            if (i != j && p->migrationRates[i * P + j] != 0.0) e->h_has_mig = true;
    if (S > 64) return fail(e, VGX_ERR_ARG, "vgx_set_params: at most 64 susceptibility groups are supported");
    // uniform mutation model (every haplotype has the same mRate / hapMutType rows): table for the tau kernels
    e->h_mut_uniform = true;
    for (int64_t h = 1; h < H && e->h_mut_uniform; h++)
        if (memcmp(p->mRate + h * sites, p->mRate, (size_t)sites * 8) != 0 ||
            memcmp(p->hapMutType + h * sites * 3, p->hapMutType, (size_t)sites * 24) != 0)
            e->h_mut_uniform = false;
    e->h_mut_total = 0.0;
    for (int64_t s2 = 0; s2 < sites && s2 < 16; s2++) {
        const double *hm = p->hapMutType + s2 * 3;
        for (int i = 0; i < 3; i++) {
            e->h_mutp[s2][i] = sites > 0 ? p->mRate[s2] * hm[i] / (hm[0] + hm[1] + hm[2]) : 0.0;
            e->h_mut_total += e->h_mutp[s2][i];
        }
    }
    for (int64_t p1 = 0; p1 < P; p1++) {  // pyx:289-297
        e->mig[(size_t)(p1 * P + p1)] = 1.0;
        double a = 0.0;
        for (int64_t p2 = 0; p2 < P; p2++) {
            if (p1 == p2) continue;
            e->mig[(size_t)(p1 * P + p1)] -= e->mig[(size_t)(p1 * P + p2)];
            a += e->mig[(size_t)(p2 * P + p1)] * (double)p->sizes[p2];
        }
        // NB: the reference reads migrationRates[pn2, pn1] for pn2 != pn1 only, so the not-yet-rewritten
        // diagonals of later rows never enter (pyx:296)
        a += e->mig[(size_t)(p1 * P + p1)] * (double)p->sizes[p1];
        e->actualSizes[(size_t)p1] = a;
    }
    double maxEffectiveBirth = 0.0;  // pyx:340-344
    for (int64_t h = 0; h < H; h++)
        for (int64_t s = 0; s < S; s++) {
            double v = p->bRate[h] * p->susceptibility[h * S + s];
            if (v > maxEffectiveBirth) maxEffectiveBirth = v;
        }

    int rc = 0;
    rc |= upload(e, e->p_cls, e->cls.data(), (size_t)H);
    rc |= upload(e, e->p_suscType, p->suscType, (size_t)H);
    rc |= upload(e, e->p_mRate, p->mRate, (size_t)(H * sites));
    rc |= upload(e, e->p_hapMutType, p->hapMutType, (size_t)(H * sites * 3));
    rc |= upload(e, e->p_bRate, p->bRate, (size_t)H);
    rc |= upload(e, e->p_susc, p->susceptibility, (size_t)(H * S));
    rc |= upload(e, e->p_cd, c_d.data(), c_d.size());
    rc |= upload(e, e->p_cs, c_s.data(), c_s.size());
    rc |= upload(e, e->p_ctm, c_tm.data(), c_tm.size());
    rc |= upload(e, e->p_cbidx, c_bidx.data(), c_bidx.size());
    rc |= upload(e, e->p_cstype, c_stype.data(), c_stype.size());
    rc |= upload(e, e->p_cbb, cb_b.data(), cb_b.size());
    rc |= upload(e, e->p_cbsig, cb_sig.data(), cb_sig.size());
    {
        static const int32_t zero_i = 0;
        static const double zero_d = 0.0;
        const size_t ns = e->h_seg_par.size();
        rc |= upload(e, e->q_segpar, ns ? e->h_seg_par.data() : &zero_i, ns ? ns : 1);
        rc |= upload(e, e->q_segsn, ns ? e->h_seg_sn.data() : &zero_i, ns ? ns : 1);
        rc |= upload(e, e->q_segsig, ns ? e->h_seg_sig.data() : &zero_d, ns ? ns : 1);
        rc |= upload(e, e->q_cbseg, e->h_cb_seg.data(), e->h_cb_seg.size());
    }
    {
        static const int32_t zero_i = 0;
        static const double zero_d = 0.0;
        const size_t ns = e->h_so_sn.size();
        rc |= upload(e, e->so_sn, ns ? e->h_so_sn.data() : &zero_i, ns ? ns : 1);
        rc |= upload(e, e->so_sig, ns ? e->h_so_sig.data() : &zero_d, ns ? ns : 1);
        std::vector<double> rcp((size_t)P);
        for (int64_t pn = 0; pn < P; pn++) rcp[(size_t)pn] = 1.0 / e->actualSizes[(size_t)pn];
        rc |= upload(e, e->so_rcp, rcp.data(), rcp.size());
        rc |= upload(e, e->so_hapcls, e->h_so_hapcls.data(), e->h_so_hapcls.size());
        rc |= upload(e, e->so_nnz, e->h_so_nnz.data(), e->h_so_nnz.size());
        rc |= upload(e, e->so_tsn, e->h_so_tsn.data(), e->h_so_tsn.size());
        rc |= upload(e, e->so_tsig, e->h_so_tsig.data(), e->h_so_tsig.size());
        rc |= upload(e, e->so_clssig, e->h_so_clssig.data(), e->h_so_clssig.size());
        rc |= upload(e, e->so_pass, e->h_so_pass.data(), e->h_so_pass.size());
        if (rc == 0 && hipStreamSynchronize(e->stream) != hipSuccess) rc = VGX_ERR_HIP;   // rcp goes out of scope
    }
    rc |= upload(e, e->p_sizes, p->sizes, (size_t)P);
    rc |= upload(e, e->p_cdBefore, p->contactDensityBeforeLockdown, (size_t)P);
    rc |= upload(e, e->p_cdAfter, p->contactDensityAfterLockdown, (size_t)P);
    rc |= upload(e, e->p_startLD, p->startLD, (size_t)P);
    rc |= upload(e, e->p_endLD, p->endLD, (size_t)P);
    rc |= upload(e, e->p_sampMult, p->samplingMultiplier, (size_t)P);
    rc |= upload(e, e->p_actualSizes, e->actualSizes.data(), (size_t)P);
    rc |= upload(e, e->p_mig, e->mig.data(), (size_t)(P * P));
    // uniform migration: one common off-diagonal probability and (hence) one common recomputed diagonal
    e->h_mig_uniform = e->h_has_mig && P >= 2 && P <= 1024;
    for (int64_t i = 0; i < P && e->h_mig_uniform; i++)
        for (int64_t j = 0; j < P; j++) {
            const double v = e->mig[(size_t)(i * P + j)];
            if (v != (i == j ? e->mig[0] : e->mig[1])) { e->h_mig_uniform = false; break; }
        }
    if (e->h_mig_uniform) { e->h_mig_d = e->mig[0]; e->h_mig_b = e->mig[1]; }
    rc |= upload(e, e->p_suscTrans, p->suscepTransition, (size_t)(S * S));
    rc |= upload(e, e->p_suscCumul, e->suscepCumul.data(), (size_t)S);
    if (rc) return VGX_ERR_HIP;
    HIPCHECK(e, hipStreamSynchronize(e->stream));  // the sources are caller/stack memory

    VgxDevParams &d = e->dp;
    d.H = (int32_t)H; d.P = (int32_t)P; d.S = (int32_t)S; d.sites = (int32_t)sites; d.C = e->C; d.CB = e->CB;
    d.cls = (const int32_t *)e->p_cls.p; d.suscType = (const int64_t *)e->p_suscType.p;
    d.mRate = (const double *)e->p_mRate.p; d.hapMutType = (const double *)e->p_hapMutType.p;
    d.bRate = (const double *)e->p_bRate.p; d.susc = (const double *)e->p_susc.p;
    d.c_d = (const double *)e->p_cd.p; d.c_s = (const double *)e->p_cs.p; d.c_tm = (const double *)e->p_ctm.p;
    d.c_bidx = (const int32_t *)e->p_cbidx.p; d.c_stype = (const int32_t *)e->p_cstype.p;
    d.cb_b = (const double *)e->p_cbb.p;
    d.cb_sigma = (const double *)e->p_cbsig.p;
    d.sizes = (const int64_t *)e->p_sizes.p; d.cdBefore = (const double *)e->p_cdBefore.p;
    d.cdAfter = (const double *)e->p_cdAfter.p; d.startLD = (const double *)e->p_startLD.p;
    d.endLD = (const double *)e->p_endLD.p; d.sampMult = (const double *)e->p_sampMult.p;
    d.actualSizes = (const double *)e->p_actualSizes.p; d.mig = (const double *)e->p_mig.p;
    d.suscepTransition = (const double *)e->p_suscTrans.p; d.suscepCumul = (const double *)e->p_suscCumul.p;
    d.maxEffectiveBirth = maxEffectiveBirth;
    d.recombination = e->recombination; d.genome_length = e->genome_length;
    d.sitesPosition = (const int64_t *)e->p_sitesPos.p;
    e->have_params = true;
    e->tau_staged = false;
    e->dev_state_valid = false;  // class ids in the occupancy lists refer to the old parameter rows
    return VGX_OK;
}

extern "C" int vgx_set_recombination(vgx_engine *e, double recombination_probability, int64_t genome_length,
                                     const int64_t *sitesPosition) {
    if (!e) return VGX_ERR_ARG;
    const int64_t sites = e->d.sites;
    if (!(recombination_probability >= 0.0 && recombination_probability <= 1.0) || genome_length < 0)
        return fail(e, VGX_ERR_ARG, "vgx_set_recombination: probability outside [0, 1] or negative genome length");
    if (recombination_probability != 0.0 && (sites < 2 || !sitesPosition))
        return fail(e, VGX_ERR_ARG, "vgx_set_recombination: recombination needs at least two sites and their positions "
                                    "(the reference allocates its scratch vector only then, pyx:98-102)");
    HIPCHECK(e, hipSetDevice(e->device));
    e->recombination = recombination_probability;
    e->genome_length = genome_length;
    if (sites > 0 && sitesPosition) {
        int rc = upload(e, e->p_sitesPos, sitesPosition, (size_t)sites);
        if (rc) return rc;
        HIPCHECK(e, hipStreamSynchronize(e->stream));
    }
    e->dp.recombination = e->recombination; e->dp.genome_length = e->genome_length;
    e->dp.sitesPosition = (const int64_t *)e->p_sitesPos.p;
    return VGX_OK;
}

extern "C" int vgx_get_recombinations(vgx_engine *e, int64_t replicate, int64_t cap, int64_t *idevents, int64_t *his,
                                      int64_t *hi2s, int64_t *nhis, int64_t *posRecombs, int64_t *n) {
    if (!e || !n || replicate < 0 || replicate >= e->R) return VGX_ERR_ARG;
    if (!e->sc_host_valid) return fail(e, VGX_ERR_ARG, "vgx_get_recombinations: no simulate call yet");
    *n = 0;
    if (e->last_was_tau || e->rec_cap == 0) return VGX_OK;   // (rec_cap is set by calls with recombination only)
    HIPCHECK(e, hipSetDevice(e->device));
    int64_t cnt = std::min<int64_t>(e->sc_host[(size_t)replicate].rec_n, e->rec_cap);
    *n = cnt;
    cnt = std::min(cnt, cap);
    if (cnt <= 0) return VGX_OK;
    std::vector<int64_t> rec((size_t)cnt * 5);
    HIPCHECK(e, hipMemcpy(rec.data(), (int64_t *)e->r_rec.p + replicate * e->rec_cap * 5, (size_t)cnt * 40, hipMemcpyDeviceToHost));
    int64_t *dst[5] = {idevents, his, hi2s, nhis, posRecombs};
    for (int c = 0; c < 5; c++)
        if (dst[c])
            for (int64_t i = 0; i < cnt; i++) dst[c][i] = rec[(size_t)(i * 5 + c)];
    return VGX_OK;
}

extern "C" int vgx_set_state(vgx_engine *e, const vgx_state *s) {
    if (!e || !s) return VGX_ERR_ARG;
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum;
    if (!s->susceptible || !s->infectious || !s->lockdownON || !s->contactDensity)
        return fail(e, VGX_ERR_ARG, "vgx_set_state: susceptible, infectious, lockdownON and contactDensity are required");
    HostState &h = e->hs;
    h.susceptible.assign(s->susceptible, s->susceptible + P * S);
    h.infectious.assign(s->infectious, s->infectious + P * H);
    if (s->initial_susceptible) h.initial_susceptible.assign(s->initial_susceptible, s->initial_susceptible + P * S);
    else h.initial_susceptible.assign((size_t)(P * S), 0);
    if (s->initial_infectious) h.initial_infectious.assign(s->initial_infectious, s->initial_infectious + P * H);
    else h.initial_infectious.assign((size_t)(P * H), 0);
    h.lockdownON.assign(s->lockdownON, s->lockdownON + P);
    h.contactDensity.assign(s->contactDensity, s->contactDensity + P);
    h.totalSusceptible.assign((size_t)P, 0);
    h.totalInfectious.assign((size_t)P, 0);
    if (s->totalSusceptible) h.totalSusceptible.assign(s->totalSusceptible, s->totalSusceptible + P);
    if (s->totalInfectious) h.totalInfectious.assign(s->totalInfectious, s->totalInfectious + P);
    h.first_simulation = s->first_simulation;
    h.globalInfectious = s->globalInfectious;
    h.bCounter = s->bCounter; h.dCounter = s->dCounter; h.sCounter = s->sCounter; h.mCounter = s->mCounter;
    h.iCounter = s->iCounter; h.swapLockdown = s->swapLockdown; h.migPlus = s->migPlus; h.migNonPlus = s->migNonPlus;
    h.good_attempt = s->good_attempt;
    h.currentTime = s->currentTime; h.totalRate = s->totalRate; h.totalMigrationRate = s->totalMigrationRate;
    h.tau_l = s->tau_l;
    h.ev_ptr = s->ev_ptr; h.ev_size = s->ev_size;
    e->have_state = true;
    e->dev_state_valid = false;
    e->tau_staged = false;
    return VGX_OK;
}

// dense [P][H] -> ordered occupancy lists
static void build_lists(const vgx_engine *e, const std::vector<int64_t> &dense, std::vector<int32_t> &nocc,
                        std::vector<int32_t> &hap, std::vector<int32_t> &cl, std::vector<int64_t> &cnt, int64_t &cap) {
    const int64_t H = e->d.hapNum, P = e->d.popNum;
    nocc.assign((size_t)P, 0);
    int64_t mx = 0;
    for (int64_t pn = 0; pn < P; pn++) {
        int64_t n = 0;
        for (int64_t h = 0; h < H; h++) n += dense[(size_t)(pn * H + h)] != 0;
        nocc[(size_t)pn] = (int32_t)n;
        mx = std::max(mx, n);
    }
    cap = std::max<int64_t>(mx, 1);
    hap.assign((size_t)(P * cap), 0);
    cl.assign((size_t)(P * cap), 0);
    cnt.assign((size_t)(P * cap), 0);
    for (int64_t pn = 0; pn < P; pn++) {
        int64_t k = 0;
        for (int64_t h = 0; h < H; h++) {
            int64_t v = dense[(size_t)(pn * H + h)];
            if (v != 0) {
                hap[(size_t)(pn * cap + k)] = (int32_t)h;
                cl[(size_t)(pn * cap + k)] = e->cls[(size_t)h];
                cnt[(size_t)(pn * cap + k)] = v;
                k++;
            }
        }
    }
}

static size_t lds_bytes_for(const vgx_engine *e) {
    return vgxi_direct_lds_bytes((int)e->d.popNum, (int)e->d.susNum, e->C, e->CB);
}

// PrepareParameters' first-call part (pyx:435-448) + FirstInfection (pyx:234-242) on the host state
static void prepare_first(vgx_engine *e) {
    HostState &h = e->hs;
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum;
    if (!h.first_simulation) {
        if (h.globalInfectious == 0) {
            for (int64_t sn = 0; sn < S; sn++) {
                if (h.susceptible[(size_t)sn] == 0) continue;
                h.susceptible[(size_t)sn] -= 1;
                h.totalSusceptible[0] -= 1;
                h.infectious[0] += 1;
                h.totalInfectious[0] += 1;
                h.globalInfectious += 1;
                break;
            }
        }
        h.globalInfectious = 0;
        for_parts(P, [&](int64_t p0, int64_t p1, unsigned) {   // whole populations per thread
            for (int64_t pn = p0; pn < p1; pn++) {
                h.totalSusceptible[(size_t)pn] = 0;
                for (int64_t sn = 0; sn < S; sn++) {
                    h.initial_susceptible[(size_t)(pn * S + sn)] = h.susceptible[(size_t)(pn * S + sn)];
                    h.totalSusceptible[(size_t)pn] += h.susceptible[(size_t)(pn * S + sn)];
                }
                int64_t t = 0;
                for (int64_t hn = 0; hn < H; hn++) {
                    const int64_t v = h.infectious[(size_t)(pn * H + hn)];
                    h.initial_infectious[(size_t)(pn * H + hn)] = v;
                    t += v;
                }
                h.totalInfectious[(size_t)pn] = t;
            }
        }, H);
        for (int64_t pn = 0; pn < P; pn++) h.globalInfectious += h.totalInfectious[(size_t)pn];
        h.first_simulation = 1;
    }
}

static int init_device_state(vgx_engine *e, int64_t traj_points) {
    HostState &h = e->hs;
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum, R = e->R;
    std::vector<int32_t> nocc, hap, cl, i_nocc, i_hap, i_cl;
    std::vector<int64_t> cnt, i_cnt;
    int64_t s_cap = 1, i_cap = 1;
    build_lists(e, h.infectious, nocc, hap, cl, cnt, s_cap);
    build_lists(e, h.initial_infectious, i_nocc, i_hap, i_cl, i_cnt, i_cap);

    // list capacity per (replicate, population): worst case H when it fits the memory budget
    size_t free_b = 0, total_b = 0;
    HIPCHECK(e, hipMemGetInfo(&free_b, &total_b));
    int64_t need = std::max<int64_t>(std::max(s_cap, i_cap), 1);
    e->start_max_nocc = s_cap;
    {
        int64_t rows_s = 0, rows_i = 0;
        for (int64_t pn = 0; pn < P; pn++) { rows_s += vgx_lone_min_rows(nocc[(size_t)pn]); rows_i += vgx_lone_min_rows(i_nocc[(size_t)pn]); }
        e->start_lone_rows = std::max(rows_s, rows_i);
    }
    // capacity per list: what fits 45 % of the free memory (buffers of an earlier state count as free: they are reused), at most
    // 32 GiB for all lists together (an ensemble of 16 384 x 64 lists still gets 1600 entries each; the rest of the memory
    // belongs to the event logs and trajectories), never below the start state's longest list + one tile.  Rounded down to a
    // multiple of 64 entries: the row kernels read whole 64-entry tiles with 16-byte loads.
    const double reusable = (double)(e->r_lhap.bytes + e->r_lcls.bytes + e->r_lcnt.bytes + e->r_lcnt32.bytes);
    const double list_bytes = std::min(((double)free_b + reusable) * 0.45, 32.0 * 1073741824.0);
    int64_t budget = (int64_t)(list_bytes / (double)(R * P * 24));
    if (const char *lc = getenv("VGX_LIST_CAP")) {     // diagnostics / tests: a small list capacity (the overflow paths of the kernels)
        const long long v = atoll(lc);
        if (v > 0) budget = std::min<int64_t>(budget, (int64_t)v);
    }
    int64_t cap = std::min<int64_t>(H, std::max<int64_t>(budget, 64));
    cap = std::max(cap, std::min<int64_t>(H, need + 64));
    if (cap < H) cap = std::max<int64_t>((cap / 64) * 64, ((need + 63) / 64) * 64);
    if (cap < need) return fail(e, VGX_ERR_CAPACITY, "occupancy lists do not fit device memory");
    e->cap = cap;

    int rc = 0;
    rc |= ensure(e, e->r_popD, (size_t)(R * PD_COUNT * P) * 8);
    rc |= ensure(e, e->r_popI, (size_t)(R * PI_COUNT * P) * 8);
    rc |= ensure(e, e->r_sus, (size_t)(R * P * S) * 8);
    rc |= ensure(e, e->r_immSrc, (size_t)(R * P * S) * 8);
    rc |= ensure(e, e->r_birthC, (size_t)(R * P * e->CB) * 8);
    rc |= ensure(e, e->r_xC, (size_t)(R * P * e->CB * S) * 8);
    rc |= ensure(e, e->r_effMig, (size_t)(R * P * P) * 8);
    rc |= ensure(e, e->r_nocc, (size_t)(R * P) * 4);
    rc |= ensure(e, e->r_lhap, (size_t)(R * P * cap + 64) * 4);
    rc |= ensure(e, e->r_lcls, (size_t)(R * P * cap) * 4);
    rc |= ensure(e, e->r_lcnt, (size_t)(R * P * cap + 64) * 8);   // + one tile: vgx_quad.hip reads whole 64-entry tiles
    const bool want32 = P <= 64 && S == 1 && e->C == 1 && e->CB == 1;   // shapes the four-replicates-per-wavefront kernel takes
    if (want32) rc |= ensure(e, e->r_lcnt32, (size_t)(R * P * cap + 64) * 4 + (size_t)(R * P * cap) + 128);   // + the one-byte copy of vgx_quad_long_kernel
    const int64_t capT = cap / 64 + 1;
    rc |= ensure(e, e->r_ltsum, (size_t)(R * P * capT) * 16 + 64);   // tile sums, then the exact row kernel's cached running sums
    rc |= ensure(e, e->r_sc, (size_t)R * sizeof(VgxRepScalars));
    rc |= ensure(e, e->r_prof, (size_t)(R * VGX_PROF_SLOTS) * 8);
    if (rc) return rc;
    HIPCHECK(e, hipMemsetAsync(e->r_effMig.p, 0, (size_t)(R * P * P) * 8, e->stream));
    HIPCHECK(e, hipMemsetAsync(e->r_popD.p, 0, (size_t)(R * PD_COUNT * P) * 8, e->stream));
    HIPCHECK(e, hipMemsetAsync(e->r_immSrc.p, 0, (size_t)(R * P * S) * 8, e->stream));

    rc |= upload(e, e->s_nocc, nocc.data(), nocc.size());
    rc |= upload(e, e->s_hap, hap.data(), hap.size());
    rc |= upload(e, e->s_cls, cl.data(), cl.size());
    rc |= upload(e, e->s_cnt, cnt.data(), cnt.size());
    rc |= upload(e, e->s_sus, h.susceptible.data(), h.susceptible.size());
    rc |= upload(e, e->s_cd, h.contactDensity.data(), h.contactDensity.size());
    std::vector<int64_t> tot((size_t)(3 * P));
    for (int64_t pn = 0; pn < P; pn++) {
        tot[(size_t)pn] = h.totalSusceptible[(size_t)pn];
        tot[(size_t)(P + pn)] = h.totalInfectious[(size_t)pn];
        tot[(size_t)(2 * P + pn)] = h.lockdownON[(size_t)pn];
    }
    rc |= upload(e, e->s_tot, tot.data(), tot.size());
    rc |= upload(e, e->i_nocc, i_nocc.data(), i_nocc.size());
    rc |= upload(e, e->i_hap, i_hap.data(), i_hap.size());
    rc |= upload(e, e->i_cls, i_cl.data(), i_cl.size());
    rc |= upload(e, e->i_cnt, i_cnt.data(), i_cnt.size());
    rc |= upload(e, e->i_sus, h.initial_susceptible.data(), h.initial_susceptible.size());
    rc |= upload(e, e->r_seeds, e->seeds.data(), e->seeds.size());
    if (rc) return VGX_ERR_HIP;

    e->sc_host.assign((size_t)R, VgxRepScalars{});
    for (int64_t r = 0; r < R; r++) {
        VgxRepScalars &s = e->sc_host[(size_t)r];
        s.currentTime = h.currentTime; s.totalRate = h.totalRate; s.totalMig = h.totalMigrationRate; s.tau_l = h.tau_l;
        s.globalInfectious = h.globalInfectious;
        s.bCounter = h.bCounter; s.dCounter = h.dCounter; s.sCounter = h.sCounter; s.mCounter = h.mCounter;
        s.iCounter = h.iCounter; s.swapLockdown = h.swapLockdown; s.migPlus = h.migPlus; s.migNonPlus = h.migNonPlus;
        s.good_attempt = h.good_attempt;
        s.ev_ptr = h.ev_ptr;
    }
    HIPCHECK(e, hipMemcpyAsync(e->r_sc.p, e->sc_host.data(), (size_t)R * sizeof(VgxRepScalars), hipMemcpyHostToDevice, e->stream));

    VgxDevRep &d = e->dr;
    d.popD = (double *)e->r_popD.p; d.popI = (int64_t *)e->r_popI.p; d.sus = (int64_t *)e->r_sus.p;
    d.immSrc = (double *)e->r_immSrc.p; d.birthC = (double *)e->r_birthC.p; d.xC = (double *)e->r_xC.p;
    d.effMig = (double *)e->r_effMig.p; d.nocc = (int32_t *)e->r_nocc.p; d.lhap = (int32_t *)e->r_lhap.p;
    d.lcls = (int32_t *)e->r_lcls.p; d.lcnt = (int64_t *)e->r_lcnt.p; d.cap = cap;
    d.lcnt32 = want32 ? (int32_t *)e->r_lcnt32.p : nullptr;
    d.ltsum = (int64_t *)e->r_ltsum.p; d.capT = capT;
    HIPCHECK(e, hipMemsetAsync(e->r_ltsum.p, 0, (size_t)(R * P * capT) * 8, e->stream));
    d.i_nocc = (const int32_t *)e->i_nocc.p; d.i_hap = (const int32_t *)e->i_hap.p;
    d.i_cls = (const int32_t *)e->i_cls.p; d.i_cnt = (const int64_t *)e->i_cnt.p; d.i_cap = i_cap;
    d.i_sus = (const int64_t *)e->i_sus.p;
    d.sc = (VgxRepScalars *)e->r_sc.p; d.seeds = (const int64_t *)e->r_seeds.p;
    d.prof = (unsigned long long *)e->r_prof.p;
    HIPCHECK(e, hipMemsetAsync(e->r_prof.p, 0, (size_t)(R * VGX_PROF_SLOTS) * 8, e->stream));

    HIPCHECK(e, vgxi_launch_init_reps(&d, (int)P, (int)S, R, (const int32_t *)e->s_nocc.p, (const int32_t *)e->s_hap.p,
                                      (const int32_t *)e->s_cls.p, (const int64_t *)e->s_cnt.p, s_cap,
                                      (const int64_t *)e->s_sus.p, (const double *)e->s_cd.p,
                                      (const int64_t *)e->s_tot.p, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));  // host vectors above go out of scope
    e->dev_state_valid = true;
    e->counts32_valid = want32;   // (vgx_init_reps_kernel fills both)
    e->counts64_valid = true;
    (void)traj_points;
    return VGX_OK;
}

static int direct_core(vgx_engine *e, int64_t iterations, int64_t sample_size, float time, int64_t attempts,
                       const vgx_run_opts *opts);
static int host_clock(vgx_engine *e, int64_t rep);

extern "C" int vgx_simulate_direct(vgx_engine *e, int64_t iterations, int64_t sample_size, float time,
                                   int64_t attempts, const vgx_run_opts *opts) {
    if (!e) return VGX_ERR_ARG;
    e->last_was_tau = false;
    return direct_core(e, iterations, sample_size, time, attempts, opts);
}

static int direct_core(vgx_engine *e, int64_t iterations, int64_t sample_size, float time, int64_t attempts,
                       const vgx_run_opts *opts) {
    if (!e->have_params || !e->have_state) return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: set params and state first");
    if (iterations < 0 || attempts < 0) return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: negative iterations/attempts");
    HIPCHECK(e, hipSetDevice(e->device));
    const int64_t P = e->d.popNum, R = e->R;
    vgx_run_opts o{};
    o.record_events = 1;
    if (opts) o = *opts;
    if (o.max_loop_factor <= 0) o.max_loop_factor = 1024;
    if (o.mode < 0 || o.mode > 2) return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: mode must be 0 (exact), 1 (fast) or 2 (fast, Philox stream)");
    size_t lds = lds_bytes_for(e);
    if (lds > 160 * 1024)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the population/class tables need " + std::to_string(lds) +
                                        " bytes of LDS per wavefront (limit 163840): too many populations x rate classes");

    HostState &h = e->hs;
    const bool fresh_state = !e->dev_state_valid;
    if (!e->dev_state_valid) {
        prepare_first(e);
        int rc = init_device_state(e, o.traj_points);
        if (rc) return rc;
    }
    std::vector<double> cont_t0((size_t)R, h.currentTime);
    {
        std::vector<double> &t0 = cont_t0;
        if (!fresh_state && e->sc_host_valid && e->sc_host.size() == (size_t)R) {
            // A call that continues the device-resident state of the previous one: its clock starts where the HOST clock of the
            // previous call ended (the reference's own libm sums), rebuilt now from that call's logs while they still exist —
            // for up to 5e7 loop iterations over all replicates (about half a second of host time); beyond that, and after calls
            // without an event log, from the device clock (vgx_log: < 1 ulp per step from the host's).
            int64_t work = 0;
            for (int64_t r = 0; r < R; r++) work += e->sc_host[(size_t)r].loop_iterations;
            // After a call of the latency kernel without its device clock (event log, no time limit, no trajectories) the device's
            // currentTime is still that call's START time: the rebuild is then not optional, whatever it costs, and its result goes
            // back to the device before the continued call reads it (time limit, trajectory grid, calls without a log).
            const bool stale = e->dev_clock_stale;
            const bool rebuild = e->direct_logs_valid && (work <= 50000000 || stale);
            if (stale && !rebuild)
                return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the previous call ran without a device clock and left no event log to "
                                            "rebuild it from: set the state again (vgx_set_state) before continuing");
            const int64_t mism = e->clock_mismatches;
            for (int64_t r = 0; r < R; r++) {
                t0[(size_t)r] = e->sc_host[(size_t)r].currentTime;
                const bool ok = rebuild && host_clock(e, r) == VGX_OK && e->hc.exact;
                if (ok) t0[(size_t)r] = e->hc.final_time;
                else if (stale)
                    return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: cannot rebuild the clock of replicate " + std::to_string(r) +
                                                " after a call without a device clock");
            }
            e->clock_mismatches = mism;   // (counted when a caller fetches the replicate)
            e->hc.rep = -1;
            if (stale) {
                for (int64_t r = 0; r < R; r++) e->sc_host[(size_t)r].currentTime = t0[(size_t)r];
                HIPCHECK(e, hipMemcpy(e->r_sc.p, e->sc_host.data(), (size_t)R * sizeof(VgxRepScalars), hipMemcpyHostToDevice));
                e->dev_clock_stale = false;
            }
        }
    }
    // events.ptr / events.size as maintained by the caller's Events.CreateEvents (events.pxi:52-68)
    const int64_t ev_ptr0 = h.ev_ptr, ev_size = h.ev_size;
    if (ev_size < ev_ptr0) return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: ev_size < ev_ptr");
    // every replicate's own events.ptr (one value after vgx_set_state; where each one stopped when the call continues)
    e->call_ev0.assign((size_t)R, ev_ptr0);
    if (!fresh_state && e->sc_host_valid && e->sc_host.size() == (size_t)R)
        for (int64_t r = 0; r < R; r++) e->call_ev0[(size_t)r] = e->sc_host[(size_t)r].ev_ptr;
    const int64_t ev_min = *std::min_element(e->call_ev0.begin(), e->call_ev0.end());
    // a Restart (pyx:414-415) rewinds the log to 0, so the device log must then cover [0, ev_size)
    const bool may_restart = ev_min <= 100 && iterations > 100;
    e->ev_base = may_restart ? 0 : ev_min;
    e->ev_ptr0 = ev_ptr0;
    int64_t evcap = o.record_events ? std::max<int64_t>(ev_size - e->ev_base, 1) : 1;
    // lockdown log: a population can switch on only if its threshold lies below its size, off only if it is on
    bool ld_possible = false;
    for (int64_t pn = 0; pn < P; pn++)
        if (e->h_startLD[(size_t)pn] * (double)e->sizes[(size_t)pn] < (double)e->sizes[(size_t)pn] || h.lockdownON[(size_t)pn] != 0)
            ld_possible = true;
    e->loc_cap = ld_possible ? VGX_LOC_CAP : 1;
    e->fa_cap = (ld_possible && may_restart && o.record_events) ? VGX_FA_CAP : 0;
    int rc = 0;
    rc |= ensure(e, e->r_locrec, (size_t)(R * e->loc_cap * 2) * 4);
    rc |= ensure(e, e->r_loctime, (size_t)(R * e->loc_cap) * 8);
    rc |= ensure(e, e->r_lociter, (size_t)(R * e->loc_cap) * 8);
    rc |= ensure(e, e->r_farate, (size_t)(R * e->fa_cap) * 8);
    rc |= ensure(e, e->r_fakey, (size_t)(R * e->fa_cap) * 8);
    rc |= ensure(e, e->r_evrate, (size_t)(R * evcap) * 8);
    rc |= ensure(e, e->r_evcols, (size_t)(R * evcap * VGX_EV_COLS) * 4);
    if (o.traj_points > 0) rc |= ensure(e, e->r_traj, (size_t)(R * o.traj_points * P * 2) * 8);
    if (rc) return rc;
    e->evcap = evcap;
    e->traj_points = o.traj_points;
    e->last_ev_size = ev_size;

    VgxDirectArgs a{};
    a.p = e->dp;
    a.r = e->dr;
    a.r.ev_rate = (double *)e->r_evrate.p;
    a.r.ev_cols = (int32_t *)e->r_evcols.p;
    a.r.loc_rec = (int32_t *)e->r_locrec.p; a.r.loc_time = (double *)e->r_loctime.p; a.r.loc_iter = (int64_t *)e->r_lociter.p;
    a.r.loc_cap = e->loc_cap;
    a.r.fa_rate = (double *)e->r_farate.p; a.r.fa_key = (int64_t *)e->r_fakey.p; a.r.fa_cap = e->fa_cap;
    // the host clock starts where the caller's state stands (one value for all replicates after vgx_set_state; the
    // replicates' own device clocks when a call continues without a new state)
    e->call_t0 = cont_t0;
    e->call_recorded = o.record_events != 0;
    e->call_has_tlimit = !(time == -1.0f);
    e->call_tlimit = (double)time;
    e->hc.rep = -1;
    a.r.evcap = evcap;
    a.r.ev_base = e->ev_base;
    a.r.traj = o.traj_points > 0 ? (double *)e->r_traj.p : nullptr;
    a.r.traj_points = o.traj_points;
    a.r.traj_t0 = o.traj_t0;
    a.r.traj_dt = o.traj_points > 1 ? (o.traj_t1 - o.traj_t0) / (double)(o.traj_points - 1) : 0.0;
    a.n_replicates = R;
    a.iterations = iterations; a.sample_size = sample_size; a.attempts = attempts; a.time = time;
    a.ev_size = ev_size;
    a.max_loop = o.max_loop_factor * std::max<int64_t>(iterations, 1) + (1 << 20);
    a.record_events = o.record_events ? 1 : 0;
    a.lds_bytes = (int32_t)lds;

    // Kernel choice: small models run one replicate per LANE (vgx_lanes.hip: the reference's serial loops, dense state);
    // everything else one replicate per wavefront.  opts.kernel: 0 = automatic, 1 = wavefront, 2 = lane, 3 = four replicates per
    // wavefront (whichever of its forms takes the model), 4 = its general form, 5 = single trajectory (vgx_solo.hip), 6 = single
    // trajectory of a large haplotype space (vgx_lone.hip).
    const int64_t H = e->d.hapNum, S = e->d.susNum;
    // Recombination (pyx:575-596), exact mode only: the single-trajectory kernel, the general row kernel (its *_rec instantiations), the
    // wavefront kernel (any shape), and — when asked for — the lane kernel (serial, dense state) while the dense arrays fit.
    const bool recomb = e->recombination != 0.0;
    bool fast_remapped = false;   // a FAST request that the exact row / latency kernels serve better (undone below if the call lands on neither)
    bool philox_remapped = false; // a counter-based request on a general model: the general row kernel's exact arithmetic on that stream
    {
        // FAST mode promises the exact mode's integer rows on the same seed and times within 1e-9 — which the exact mode delivers.  Its
        // own row kernel (vgx_quadf.hip) takes one-class models; for every other model that the exact row kernels or the latency
        // kernel take, those ARE the fast path (tools/probe_fast_general.py, 16 384 replicates of the Table-3 model: 1.4e9 / 1.1e9 /
        // 4.2e8 events/s at 2 / 10 / 100 demes against 2.5e8 / 2.5e8 / 1.3e8 on the FAST form of the wavefront kernel).  The
        // counter-based stream (mode 2) is another trajectory and stays where it is.
        const bool one_class_shape = P <= 64 && S == 1 && e->C == 1 && e->CB == 1 && !ld_possible;
        const bool general_shape = P <= VGX_QG_MAX_P && S <= VGX_QG_MAX_S && e->C <= VGX_QG_MAX_C && e->CB <= VGX_QG_MAX_CB &&
                                   3 * S + e->CB <= VGX_QG_MAX_W && (int64_t)e->h_seg_par.size() <= VGX_QG_MAX_SEG;
        if (o.mode == 1 && o.kernel == 0 && !recomb && !one_class_shape && general_shape) { o.mode = 0; fast_remapped = true; }
        // The counter-based stream (mode 2) on such a model: the general row kernel's EXACT arithmetic on the Philox stream — the exact
        // mode's rows for those random numbers (the oracle on the same stream agrees bit for bit), at that kernel's rate instead of the
        // wavefront kernel's (16 384 replicates of the Table-3 model, K = 10: 1.3e9 against 2.5e8 events/s).  The latency kernels and the
        // one-class exact row kernel draw from the PCG64 stream only.
        // Likewise the latency kernels (one trajectory or a few: vgx_solo.hip, vgx_lone.hip), for every shape they take.
        if (o.mode == 2 && (o.kernel == 0 || o.kernel == 3 || o.kernel == 4) && !recomb && !one_class_shape && general_shape) { o.mode = 0; philox_remapped = true; }
        if (o.mode == 2 && (o.kernel == 0 || o.kernel == 5 || o.kernel == 6) && !recomb) { o.mode = 0; philox_remapped = true; }
    }
    const bool philox = o.mode == 2 || philox_remapped;
    a.fast = o.mode >= 1 ? 1 : 0;
    a.rng_philox = philox ? 1 : 0;
    e->call_philox = philox;
    const bool lane_ok = o.mode == 0 && !philox_remapped && (recomb ? P * H * std::max<int64_t>(S, 1) <= (1 << 24)
                                                : (P * H <= 1024 && P <= 16 && S <= 8 && H <= e->cap));
    if (recomb && o.mode != 0)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: recombination runs in exact mode only");
    if (o.kernel < 0 || o.kernel > 6) return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: kernel must be 0..6");
    if (o.kernel == 2 && !lane_ok)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the lane-per-replicate kernel needs exact mode, popNum <= 16, "
                                    "popNum * hapNum <= 1024 and susNum <= 8");
    // One trajectory (or a few) of a small model: the latency kernel (vgx_solo.hip), the whole model in LDS and registers.
    VgxSoloArgs soa{};
    bool solo_ok = o.mode == 0 && H <= VGX_SOLO_MAX_H && P <= VGX_SOLO_MAX_P && S <= VGX_SOLO_MAX_S &&
                   (int64_t)e->h_so_sn.size() <= VGX_SOLO_MAX_SEG && H <= e->cap &&
                   ((P <= 16 && H <= 16) || ((int64_t)e->h_seg_par.size() <= VGX_SOLO_MAX_TSEG && e->h_so_npass0 >= 0 && e->h_so_npass1 >= 0));
    for (int64_t pn = 0; pn < P && solo_ok; pn++)
        if (e->sizes[(size_t)pn] >= ((int64_t)1 << 52)) solo_ok = false;   // counts are kept as doubles
    if (solo_ok) {
        soa.mig_in_lds = vgx_solo_layout((int)P, (int)H, (int)S, (int)e->d.sites, 1).total <= VGX_SOLO_MAX_LDS ? 1 : 0;
        if (vgx_solo_layout((int)P, (int)H, (int)S, (int)e->d.sites, soa.mig_in_lds).total > VGX_SOLO_MAX_LDS) solo_ok = false;
    }
    if (o.kernel == 5 && !solo_ok)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the single-trajectory kernel needs exact mode, hapNum <= 64, "
                                    "popNum <= 128, susNum <= 16 and a model that fits 160 KB of LDS");
    // measured (tools/probe_lanes.py): the lane kernel only wins for minimal models in very large ensembles (config 2 at
    // 262 144 replicates: 2.4e9 vs 7.0e8 events/s); its state lives in HBM/L2, so every other shape is latency-bound
    // (recombination: the single-trajectory kernel for few replicates of a model it takes, else the general row kernel, else the wavefront kernel)
    // (recombination, measured in round 4 — tools/probe_recomb_ens.py, 16 384 replicates of the recomb_a / recomb_pos models: the general row
    // kernel 6.4-6.9e8 events/s, the wavefront kernel 3.1-3.4e8, the lane kernel 1.0-1.2e7: the lane kernel only when it is asked for)
    // (config 2, tools/probe_config2.py: 65 536 replicates 1.8e9 events/s here against 2.0e9 on the row kernel, 262 144: 4.0e9 against 2.1e9)
    const bool use_lanes = o.kernel == 2 || (o.kernel == 0 && !recomb && lane_ok && P * H * S <= 4 && R >= 131072);
    // Four replicates per wavefront, one per 16-lane DPP row (vgx_quad.hip): one rate class, one susceptibility group,
    // at most 64 populations, no population that can switch its lockdown state, exact mode.
    bool quad_shape = !recomb && P <= 64 && S == 1 && e->C == 1 && e->CB == 1 && !ld_possible &&
                      e->suscepCumul[0] == 0.0 && e->dr.lcnt32 != nullptr &&
                      e->cap <= ((int64_t)1 << 24);   // (the 64-ary lower bound of vgx_rowlist.h descends from stride 64^3: lists of up to 2^24 entries)
    for (int64_t pn = 0; pn < P && quad_shape; pn++)
        if (e->sizes[(size_t)pn] >= ((int64_t)1 << 31)) quad_shape = false;   // its streaming passes read 4-byte counts
    for (int64_t pn = 0; pn < P && quad_shape; pn++)
        if (h.totalSusceptible[(size_t)pn] != h.susceptible[(size_t)pn]) quad_shape = false;
    const bool quad_ok = o.mode == 0 && !philox_remapped && quad_shape;
    // One trajectory (or a few hundred) of such a model with a LARGE haplotype space: the latency kernel on occupancy lists
    // (vgx_lone.hip), every list resident in LDS.  One wavefront per CU with 160 KB each up to 256 replicates, two with 80 KB beyond
    // (512 at a time).  The row kernels' four replicates per wavefront win from about 2000 replicates on (tools/probe_lone_crossover.py,
    // config 3 / its general variant, events/s: 1536 replicates 2.6e8 / 1.6e8 here against 2.3e8 / 1.3e8 there, 2048: 2.6e8 / 1.6e8 against
    // 3.1e8 / 1.8e8).  Chosen by itself only for a state that came through
    // vgx_set_state (when the lists outgrow the heap the call runs again from that state on the row kernel) whose lists leave half
    // the heap free; opts.kernel = 6 forces it on any state (a full heap is then the call's error).
    // Its general form takes what the general row kernel takes at up to 64 populations: several rate classes (the class of a list entry
    // rides in the top six bits of its haplotype word: hapNum <= 2^26), several susceptibility groups, lockdown switches.
    VgxLoneArgs loa{};
    loa.lds_bytes = R <= 256 ? VGX_LONE_MAX_LDS : VGX_LONE_MAX_LDS / 2;
    bool lone_gen_shape = o.mode == 0 && !recomb && P <= 64 && S <= VGX_LONE_MAX_S && e->C <= VGX_LONE_MAX_C && e->CB <= VGX_LONE_MAX_CB &&
                          (int64_t)e->h_seg_par.size() <= VGX_LONE_MAX_SEG && H <= ((int64_t)1 << VGX_LONE_HAP_BITS);
    for (int64_t pn = 0; pn < P && lone_gen_shape; pn++)
        if (e->sizes[(size_t)pn] >= ((int64_t)1 << 31)) lone_gen_shape = false;   // 4-byte counts in the heap
    const bool lone_one_class = o.mode == 0 && quad_shape;      // (its one-class form: the exact row kernel's scope, whatever the stream)
    loa.general = lone_one_class ? 0 : 1;
    const int64_t lone_rows = vgx_lone_layout((int)P, loa.lds_bytes, loa.general ? (int)S : 0, loa.general ? e->CB : 0).nrows;
    const bool lone_ok = (lone_one_class || lone_gen_shape) && lone_rows >= 2 * P;
    if (o.kernel == 6 && !lone_ok)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the single-trajectory kernel for large haplotype spaces needs exact mode, popNum <= 64, "
                                    "susNum <= 16, at most 64 rate classes and 16 transmission/susceptibility classes, hapNum <= 2^26, no recombination");
    // FAST mode (order-free sums, PCG64 stream) on the same layout and scope: vgx_quadf.hip
    const bool quadf_ok = (o.mode == 1 || o.mode == 2 || philox_remapped) && quad_shape;     // FAST, with the PCG64 or the counter-based stream
    // The general form of that kernel (vgx_quadg.hip): several susceptibility groups and rate classes, lockdown switches,
    // up to 128 populations.
    const int64_t qg_W = 3 * S + e->CB;
    const bool quadg_ok = o.mode == 0 && P <= VGX_QG_MAX_P && S <= VGX_QG_MAX_S && e->C <= VGX_QG_MAX_C &&
                          e->CB <= VGX_QG_MAX_CB && qg_W <= VGX_QG_MAX_W && (int64_t)e->h_seg_par.size() <= VGX_QG_MAX_SEG;
    if (o.kernel == 3 && !quad_ok && !quadg_ok && !quadf_ok)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the four-replicates-per-wavefront kernels need "
                                    "popNum <= 128, susNum <= 8, at most 64 rate classes and 16 transmission/susceptibility classes");
    if (o.kernel == 4 && !quadg_ok)
        return fail(e, VGX_ERR_ARG, "vgx_simulate_direct: the general four-replicates-per-wavefront kernel needs exact mode, "
                                    "popNum <= 128, susNum <= 8, at most 64 rate classes and 16 transmission/susceptibility classes");
    // measured (tools/probe_single.py, round 3): the row kernels lead at every ensemble size, a single trajectory included —
    // config 2: 2.9e5 events/s against 2.0e5 on the one-replicate-per-wavefront kernel, config 3: 1.75e5 against 1.26e5; four
    // replicates in one wavefront: 1.1e6 / 5.9e5 against 7.9e5 / 4.9e5 in four wavefronts
    // measured (bench legs table3 / single_trajectory, round 4): a lone wavefront of the latency kernel does a Table-3 trajectory
    // several times faster than a row of the row kernels; those win back from a few thousand replicates on (four per wavefront)
    // ... except general models (what the general row kernel would take) in the compact layout with a small LDS footprint: there two
    // wavefronts per SIMD of this kernel beat it too (tools/probe_solo_ens.py, 16 384 replicates of the Table-3 model: K = 2 1.68e9
    // against 1.09e9 events/s, K = 10 1.37e9 against 1.09e9; K = 100, general layout, 150 KB of LDS: 4.7e7 against 3.6e8)
    bool solo_many = false;
    if (solo_ok && quad_ok) {
        // One-class models the latency kernel takes too: the row kernel needs four replicates per wavefront, so below 8192 replicates it
        // cannot give every SIMD its two wavefronts (tools/probe_oneclass_small.py, config 2 and the one-class goldens: 2048 replicates
        // 1.3-1.6e9 events/s here against 5e8 there, 4096: 1.4-1.7e9 against 1.0e9; from 8192 on the row kernel leads, 1.8-1.9e9 against 1.5-1.7e9)
        solo_many = R < 8192 && vgx_solo_layout((int)P, (int)H, (int)S, (int)e->d.sites, soa.mig_in_lds).total <= 20 * 1024;
    }
    if (solo_ok && !quad_ok) {
        const bool compact = e->h_so_ncls <= VGX_SOLO_ROWS && S <= VGX_SOLO_MAX_S && P <= 64 && (int64_t)e->h_so_maxnnz * P <= 32;
        solo_many = compact && vgx_solo_layout((int)P, (int)H, (int)S, (int)e->d.sites, soa.mig_in_lds).total <= 20 * 1024;
    }
    const bool use_solo = o.kernel == 5 || (o.kernel == 0 && solo_ok && !use_lanes && (R < 2048 || solo_many));
    const bool use_lone = !use_solo && (o.kernel == 6 || (o.kernel == 0 && lone_ok && !use_lanes && fresh_state && R <= 1536 &&
                                                          2 * e->start_lone_rows <= lone_rows && !getenv("VGX_NO_LONE")));
    const bool use_quad = !use_solo && !use_lone && ((o.kernel == 3 && quad_ok) || (o.kernel == 0 && quad_ok && !use_lanes));
    const bool use_quadf = !use_solo && !use_lone && ((o.kernel == 3 && quadf_ok) || (o.kernel == 0 && quadf_ok));
    // The general form also for FEW replicates of models with up to 16 populations (one register slot): a wavefront running alone
    // does a Table-3 trajectory at 1.7e5 events/s there against 1.0e5 on the one-replicate-per-wavefront kernel
    // (tools/probe_single.py; at 64 populations the wave kernel leads, 1.25e5 against 0.94e5).
    const bool use_quadg = !use_solo && !use_lone && !use_quad && !use_quadf && (o.kernel == 4 || (o.kernel == 3 && quadg_ok) ||
                                         (o.kernel == 0 && quadg_ok && !use_lanes && (R >= 2048 || (P <= 16 && (S > 1 || e->C > 1 || ld_possible)))));
    if (fast_remapped && !use_solo && !use_lone && !use_quadg && !use_quad && !use_lanes) {
        // (fewer than 2048 replicates of a model neither kernel takes there: the wavefront kernel's own FAST form is the faster one)
        a.fast = 1;
        o.mode = 1;
    }
    if (philox_remapped && !use_quadg && !use_solo && !use_lone) {   // (likewise: the FAST row kernel for one-class models, else the wavefront kernel's FAST form)
        a.fast = 1;
        o.mode = 2;
    }
    VgxLaneWs ws{};
    a.r.rec = nullptr; a.r.rec_cap = 0;
    e->rec_cap = 0;
    if (recomb) {
        // every recorded event can be a recombinant birth; failed attempts (<= 100 events each) keep their records
        const int64_t rec_cap = evcap + 101 * std::max<int64_t>(attempts, 1);
        int rcr = ensure(e, e->r_rec, (size_t)(R * rec_cap * 5) * 8);
        if (rcr) return rcr;
        a.r.rec = (int64_t *)e->r_rec.p; a.r.rec_cap = rec_cap;
        e->rec_cap = rec_cap;
    }
    if (use_lanes) {
        const int64_t PH = P * H;
        const int64_t n_i = PH + P * S + 3 * P, n_d = P + 3 * PH + PH * S + P * S + 5 * P + P * P;
        int rcw = ensure(e, e->r_lanews, (size_t)((n_i + n_d) * R) * 8);
        if (rcw) return rcw;
        int64_t *wi = (int64_t *)e->r_lanews.p;
        ws.inf = wi; wi += PH * R; ws.sus = wi; wi += P * S * R; ws.totS = wi; wi += P * R; ws.totI = wi; wi += P * R; ws.lock = wi; wi += P * R;
        double *wd = (double *)wi;
        ws.cd = wd; wd += P * R; ws.birth = wd; wd += PH * R; ws.tE = wd; wd += PH * R; ws.hpr = wd; wd += PH * R;
        ws.shpr = wd; wd += PH * S * R; ws.immSrc = wd; wd += P * S * R; ws.infP = wd; wd += P * R; ws.immP = wd; wd += P * R;
        ws.popR = wd; wd += P * R; ws.migR = wd; wd += P * R; ws.maxEBM = wd; wd += P * R; ws.effMig = wd; wd += P * P * R;
    }
    e->last_used_lanes = use_lanes;
    e->last_used_quad = use_quad;
    e->last_used_quadg = use_quadg;
    e->last_used_quadf = use_quadf;
    e->last_used_solo = use_solo;
    e->last_used_lone = use_lone;
    if (use_solo) {
        soa.seg_sn = (const int32_t *)e->so_sn.p; soa.seg_sig = (const double *)e->so_sig.p; soa.nseg = (int32_t)e->h_so_sn.size();
        soa.rcpAs = (const double *)e->so_rcp.p;
        soa.tseg_par = (const int32_t *)e->q_segpar.p; soa.tseg_sn = (const int32_t *)e->q_segsn.p; soa.tseg_sig = (const double *)e->q_segsig.p;
        soa.cb_seg = (const int32_t *)e->q_cbseg.p; soa.pass = (const int32_t *)e->so_pass.p;
        soa.tnseg = (int32_t)e->h_seg_par.size(); soa.npass0 = e->h_so_npass0; soa.npass1 = e->h_so_npass1;
        soa.exact_rcp_div = getenv("VGX_SOLO_PLAIN_DIV") ? 0 : 1;
        soa.hap_cls = (const int32_t *)e->so_hapcls.p; soa.cls_nnz = (const int32_t *)e->so_nnz.p; soa.cls_tsn = (const int32_t *)e->so_tsn.p;
        soa.cls_tsig = (const double *)e->so_tsig.p; soa.cls_sigma = (const double *)e->so_clssig.p;
        soa.n_cls = e->h_so_ncls;
        soa.maxterms = (int32_t)(e->h_so_maxnnz * P);
        soa.compact = 0;
        if (e->h_so_ncls <= VGX_SOLO_ROWS && S <= VGX_SOLO_MAX_S && P <= 64 && soa.maxterms <= 32 && !getenv("VGX_SOLO_GENERAL"))
            soa.compact = soa.maxterms <= 16 ? 1 : 2;
    }
    VgxQuadgArgs qga{};
    if (use_quadg) {
        int rcq = ensure(e, e->r_cold, (size_t)(R * P * qg_W) * 8);
        if (rcq) return rcq;
        qga.effMig0 = (const double *)e->r_qeff.p; qga.mebm0 = (const double *)e->r_qmebm.p; qga.has_mig0 = (const int32_t *)e->r_qflag.p;
        qga.cd0 = (const double *)e->s_cd.p;
        qga.seg_par = (const int32_t *)e->q_segpar.p; qga.seg_sn = (const int32_t *)e->q_segsn.p;
        qga.seg_sig = (const double *)e->q_segsig.p; qga.cb_seg = (const int32_t *)e->q_cbseg.p;
        qga.nseg = (int32_t)e->h_seg_par.size(); qga.W = (int32_t)qg_W;
        qga.cold = (int64_t *)e->r_cold.p;
    }
    if (use_quad || use_quadg || use_quadf || use_lone) {
        int rcq = 0;
        rcq |= ensure(e, e->r_qeff, (size_t)(P * P) * 8);
        rcq |= ensure(e, e->r_qmebm, (size_t)P * 8);
        rcq |= ensure(e, e->r_qflag, 8);
        if (rcq) return rcq;
        if (use_quadg) { qga.effMig0 = (const double *)e->r_qeff.p; qga.mebm0 = (const double *)e->r_qmebm.p; qga.has_mig0 = (const int32_t *)e->r_qflag.p; }
    }

    // the 4-byte copy of the counts is kept by the four-replicates-per-wavefront kernel alone
    // the row kernels with zero-count entries work on the 4-byte counts alone; everything else reads the 8-byte ones
    const bool leaves32 = use_quadf || (use_quad && e->start_max_nocc > 64);
    if (!e->counts64_valid && !(leaves32 && e->counts32_valid)) {
        HIPCHECK(e, vgxi_launch_counts64(e->dr.lcnt32, e->dr.lcnt, R * P * e->cap, e->stream));
        e->counts64_valid = true;
    }
    if ((use_quad || use_quadf) && !e->counts32_valid) {
        HIPCHECK(e, vgxi_launch_counts32(e->dr.lcnt, e->dr.lcnt32, R * P * e->cap, e->stream));
        e->counts32_valid = true;
    }
    // (which copy of the counts the kernel leaves current is recorded once it has been enqueued: a failure before that leaves the flags
    // describing what is on the device; whatever a failed launch may have touched is rebuilt from the host state, below)
    HIPCHECK(e, hipEventRecord(e->ev0, e->stream));
    const bool solo_clock = e->call_has_tlimit || o.traj_points > 0 || !o.record_events;
    if (use_solo) HIPCHECK(e, vgxi_launch_solo(&a, &soa, solo_clock ? 1 : 0, e->stream));
    else if (use_lone) {
        HIPCHECK(e, vgxi_launch_quad_prep(&a.p, (const double *)e->s_cd.p, (double *)e->r_qeff.p, (double *)e->r_qmebm.p,
                                          (int32_t *)e->r_qflag.p, e->stream));
        loa.effMig = (const double *)e->r_qeff.p; loa.maxEBM = (const double *)e->r_qmebm.p; loa.has_mig = (const int32_t *)e->r_qflag.p;
        loa.rcpAs = (const double *)e->so_rcp.p;
        loa.exact_rcp_div = getenv("VGX_SOLO_PLAIN_DIV") ? 0 : 1;
        loa.mut_uniform = (e->h_mut_uniform && !getenv("VGX_LONE_NO_MUTUNI")) ? 1 : 0;
        loa.seg_par = (const int32_t *)e->q_segpar.p; loa.seg_sn = (const int32_t *)e->q_segsn.p;
        loa.seg_sig = (const double *)e->q_segsig.p; loa.cb_seg = (const int32_t *)e->q_cbseg.p;
        loa.nseg = (int32_t)e->h_seg_par.size();
        HIPCHECK(e, vgxi_launch_lone(&a, &loa, solo_clock ? 1 : 0, e->stream));
    }
    else if (use_lanes) HIPCHECK(e, vgxi_launch_lanes(&a, &ws, e->stream));
    else if (use_quad) HIPCHECK(e, vgxi_launch_quad(&a, (const double *)e->s_cd.p, (double *)e->r_qeff.p, (double *)e->r_qmebm.p,
                                                    (int32_t *)e->r_qflag.p, e->start_max_nocc > 64 ? 1 : 0, e->stream));
    else if (use_quadf) HIPCHECK(e, vgxi_launch_quadf(&a, (const double *)e->s_cd.p, (double *)e->r_qeff.p, (double *)e->r_qmebm.p,
                                                      (int32_t *)e->r_qflag.p, e->stream));
    else if (use_quadg) {
        HIPCHECK(e, vgxi_launch_quad_prep(&a.p, (const double *)e->s_cd.p, (double *)e->r_qeff.p, (double *)e->r_qmebm.p,
                                          (int32_t *)e->r_qflag.p, e->stream));
        HIPCHECK(e, vgxi_launch_quadg(&a, &qga, e->stream));
    }
    else HIPCHECK(e, vgxi_launch_direct(&a, lds, e->stream));
    e->counts32_valid = use_quad || use_quadf;
    if (leaves32) e->counts64_valid = false;
    e->dev_clock_stale = (use_solo || use_lone) && !solo_clock;
    HIPCHECK(e, hipEventRecord(e->ev1, e->stream));
    if (hipStreamSynchronize(e->stream) != hipSuccess) {
        e->dev_state_valid = false;     // a kernel that did not finish leaves no state to continue from
        return fail(e, VGX_ERR_HIP, std::string("vgx_simulate_direct: kernel failed: ") + hipGetErrorString(hipGetLastError()));
    }
    HIPCHECK(e, hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1));
    e->last_launches = 1;

    e->sc_host.resize((size_t)R);
    HIPCHECK(e, hipMemcpy(e->sc_host.data(), e->r_sc.p, (size_t)R * sizeof(VgxRepScalars), hipMemcpyDeviceToHost));
    e->sc_host_valid = true;
    if (use_lone && o.kernel == 0) {
        // The lists of some replicate outgrew the LDS heap: the call is a function of the state and the seeds, so it runs again from the
        // state of vgx_set_state on the row kernel (the automatic choice takes this kernel only on such a state).
        bool full = false;
        for (int64_t r = 0; r < R; r++) full = full || e->sc_host[(size_t)r].error == (VGX_ERR_CAPACITY | (VGX_LONE_FULL_SITE << 8));
        if (full) {
            if (getenv("VGX_TIMING")) fprintf(stderr, "vgx_lone: LDS heap full, the call runs again on the row kernel\n");
            e->dev_state_valid = false;
            e->sc_host_valid = false;
            vgx_run_opts o2 = o;
            o2.kernel = (quad_ok || quadg_ok || quadf_ok) ? 3 : 1;
            if (philox_remapped) o2.mode = 2;       // (the request as it came)
            e->lone_fallbacks += 1;
            return direct_core(e, iterations, sample_size, time, attempts, &o2);
        }
    }
    e->direct_logs_valid = e->call_recorded;
    // the caller's next simulate continues from where replicate 0 stopped unless it sets a new state
    h.ev_ptr = e->sc_host[0].ev_ptr;
    for (int64_t r = 0; r < R; r++) {
        int64_t er = e->sc_host[(size_t)r].error;
        if (er) {
            const int64_t where = er >> 8;   // the quad kernel tags the site of a zero-weight alert (diagnostics)
            er &= 255;
            const char *what = er == VGX_ERR_ZERO_WEIGHT ? "zero weight sampled (fastChoose alert)"
                               : er == VGX_ERR_CAPACITY  ? "capacity exceeded (occupancy list / event or lockdown log)"
                               : er == VGX_ERR_LOOP_GUARD ? "loop guard tripped"
                                                          : "kernel error";
            return fail(e, (int)er, std::string("vgx_simulate_direct: replicate ") + std::to_string(r) + ": " + what +
                                        (where ? " [site " + std::to_string(where) + "]" : std::string()));
        }
    }
    return VGX_OK;
}

// ------------------------------------------------------------------------------------------------
#define VGX_PIN_BYTES ((int64_t)64 << 20)
static bool sites_ok16(const vgx_engine *e) { return e->d.sites <= 16; }

// SimulatePopulation_tau (pyx:2293-2346): the step loop runs on the host, the steps on the device.
template <typename T>
static int dl(vgx_engine *e, std::vector<T> &dst, const DevBuf &b, size_t n) {
    dst.resize(n);
    HIPCHECK(e, hipMemcpy(dst.data(), b.p, n * sizeof(T), hipMemcpyDeviceToHost));
    return VGX_OK;
}
template <typename T>
static int ul(vgx_engine *e, const DevBuf &b, const std::vector<T> &src) {
    HIPCHECK(e, hipMemcpy(b.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return VGX_OK;
}

// One replicate's compartments into the tau kernels' layout: 4 bytes per compartment (population sizes < 2^31, checked by the
// callers), susceptible counts and population totals.  Large states are converted chunk by chunk into two pinned staging buffers,
// the copy of one chunk overlapping the conversion of the next.
static int tau_upload_state(vgx_engine *e, int64_t r, const std::vector<int64_t> &inf, const std::vector<int64_t> &sus) {
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum;
    const int64_t n = P * H;
    std::vector<int64_t> tot((size_t)P, 0);
    for_parts(P, [&](int64_t p0, int64_t p1, unsigned) {   // whole populations per thread
        for (int64_t pn = p0; pn < p1; pn++) {
            int64_t t = 0;
            const int64_t *src = &inf[(size_t)(pn * H)];
            for (int64_t hn = 0; hn < H; hn++) t += src[hn];
            tot[(size_t)pn] = t;
        }
    }, H);
    int32_t *dst = (int32_t *)e->t_I.p + r * n;
    const int64_t chunk = VGX_PIN_BYTES / 4;
    if (n >= chunk) {
        for (int i = 0; i < 2; i++) {
            if (!e->pin[i]) HIPCHECK(e, hipHostMalloc(&e->pin[i], VGX_PIN_BYTES, hipHostMallocDefault));
            if (!e->pin_ev[i]) HIPCHECK(e, hipEventCreateWithFlags(&e->pin_ev[i], hipEventDisableTiming));
        }
        int k = 0;
        for (int64_t c0 = 0; c0 < n; c0 += chunk, k ^= 1) {
            const int64_t len = std::min<int64_t>(chunk, n - c0);
            if (c0 >= 2 * chunk) HIPCHECK(e, hipEventSynchronize(e->pin_ev[k]));   // the buffer's previous copy is through
            int32_t *buf = (int32_t *)e->pin[k];
            const int64_t *src = inf.data() + c0;
            for_parts(len, [&](int64_t b, int64_t en, unsigned) { for (int64_t i = b; i < en; i++) buf[i] = (int32_t)src[i]; });
            HIPCHECK(e, hipMemcpyAsync(dst + c0, buf, (size_t)len * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHECK(e, hipEventRecord(e->pin_ev[k], e->stream));
        }
        HIPCHECK(e, hipStreamSynchronize(e->stream));
    } else {
        std::vector<int32_t> inf32((size_t)n);
        for (int64_t i = 0; i < n; i++) inf32[(size_t)i] = (int32_t)inf[(size_t)i];
        HIPCHECK(e, hipMemcpy(dst, inf32.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    HIPCHECK(e, hipMemcpy((int64_t *)e->t_S.p + r * P * S, sus.data(), (size_t)(P * S) * 8, hipMemcpyHostToDevice));
    HIPCHECK(e, hipMemcpy((int64_t *)e->t_totInf.p + r * P, tot.data(), (size_t)P * 8, hipMemcpyHostToDevice));
    return VGX_OK;
}

static int64_t count_occupied(const vgx_engine *e) {
    const HostState &h = e->hs;
    int64_t part[16] = {0}, occupied = 0;
    for_parts(e->d.popNum * e->d.hapNum, [&](int64_t b, int64_t en, unsigned t) {
        int64_t n = 0;
        for (int64_t i = b; i < en; i++) n += h.infectious[(size_t)i] != 0;
        part[t] = n;
    });
    for (int t = 0; t < 16; t++) occupied += part[t];
    return occupied;
}

// Puts the state handed over by vgx_set_state on the device in the tau kernels' layout ahead of vgx_simulate_tau (which does it itself
// otherwise): the first-call snapshot of PrepareParameters (pyx:435-448), the count of occupied compartments, conversion and upload of
// the P x H counts of every replicate.  At BASELINE config 4 that is 2^28 compartments: about 0.1 s of host work and PCIe transfer
// that a caller who times the simulate call may want outside it.  Valid until the next vgx_set_state / vgx_set_params / simulate call.
extern "C" int vgx_stage_tau(vgx_engine *e) {
    if (!e) return VGX_ERR_ARG;
    if (!e->have_params || !e->have_state) return fail(e, VGX_ERR_ARG, "vgx_stage_tau: set params and state first");
    HIPCHECK(e, hipSetDevice(e->device));
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum, R = e->R;
    for (int64_t pn = 0; pn < P; pn++)
        if (e->sizes[(size_t)pn] >= ((int64_t)1 << 31)) return fail(e, VGX_ERR_ARG, "vgx_stage_tau: population sizes must be below 2^31");
    prepare_first(e);
    e->tau_occupied = count_occupied(e);
    int rc = 0;
    rc |= ensure(e, e->t_I, (size_t)(R * P * H) * 4);
    rc |= ensure(e, e->t_S, (size_t)(R * P * S) * 8);
    rc |= ensure(e, e->t_totInf, (size_t)(R * P) * 8);
    if (rc) return rc;
    for (int64_t r = 0; r < R; r++) {
        rc = tau_upload_state(e, r, e->hs.infectious, e->hs.susceptible);
        if (rc) return rc;
    }
    e->tau_staged = true;
    return VGX_OK;
}

extern "C" int vgx_simulate_tau(vgx_engine *e, int64_t iterations, int64_t sample_size, float time, int64_t attempts,
                                const vgx_run_opts *opts) {
    if (!e) return VGX_ERR_ARG;
    if (!e->have_params || !e->have_state) return fail(e, VGX_ERR_ARG, "vgx_simulate_tau: set params and state first");
    HIPCHECK(e, hipSetDevice(e->device));
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum, R = e->R;
    vgx_run_opts o{};
    o.record_events = 1;
    if (opts) o = *opts;
    HostState &h = e->hs;
    const int64_t ev_ptr_start = h.ev_ptr, ev_size = h.ev_size;

    // PrepareParameters (pyx:2298): first-call snapshot on the host, then CheckLockdown for every population and
    // UpdateAllRates.  The tau steps never read the direct path's rate caches; what the driver needs from
    // UpdateAllRates is only whether totalRate + totalMigrationRate is non-zero (pyx:2311).  For moderately
    // occupied states the direct kernel does that preparation exactly (run with zero attempts); for densely
    // occupied large states (its exact, lane-ordered row sums would take seconds) the lockdown switches and the
    // non-zero test are done on the host and totalRate is reported as NaN (the reference leaves a stale value).
    // VGX_TIMING=1: host-side phases of the call on stderr (diagnostics)
    const bool timing = getenv("VGX_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "vgx_simulate_tau: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    e->dev_state_valid = false;
    e->tau_loc_time.assign((size_t)R, {});
    e->tau_loc_state.assign((size_t)R, {});
    e->tau_loc_pop.assign((size_t)R, {});
    const bool staged = e->tau_staged;     // vgx_stage_tau already did the snapshot, the count and the upload of this start state
    e->tau_staged = false;                 // (the device copy stops being the start state as soon as a step is applied)
    if (!staged) prepare_first(e);
    lap("first-call snapshot");
    const int64_t occupied = staged ? e->tau_occupied : count_occupied(e);
    lap("count of occupied");
    bool rates_nonzero = false;
    int rc = 0;
    // (the device's UpdateAllRates for the start state — exact totalRate, lockdown switches — where building the direct kernels' occupancy
    // lists from the dense host arrays is cheap: at config 4's size that scan of 2.7e8 compartments is 0.3 s per call, and the host form
    // below — what a densely occupied state takes anyway — stands in)
    if (occupied <= ((int64_t)1 << 18) && P * H <= ((int64_t)1 << 24)) {
        vgx_run_opts po{};
        po.record_events = 0;
        rc = direct_core(e, 0, -1, -1.0f, 0, &po);
        if (rc) return rc;
        const VgxRepScalars prep = e->sc_host[0];
        std::vector<double> popD((size_t)(PD_COUNT * P));
        std::vector<int64_t> popI((size_t)(PI_COUNT * P));
        HIPCHECK(e, hipMemcpy(popD.data(), e->r_popD.p, popD.size() * 8, hipMemcpyDeviceToHost));
        HIPCHECK(e, hipMemcpy(popI.data(), e->r_popI.p, popI.size() * 8, hipMemcpyDeviceToHost));
        for (int64_t pn = 0; pn < P; pn++) {
            h.contactDensity[(size_t)pn] = popD[(size_t)(PD_CD * P + pn)];
            h.lockdownON[(size_t)pn] = popI[(size_t)(PI_LOCK * P + pn)];
        }
        h.swapLockdown = prep.swapLockdown;
        h.totalRate = prep.totalRate;
        h.totalMigrationRate = prep.totalMig;
        rates_nonzero = prep.totalRate + prep.totalMig != 0.0;
        int64_t n = std::min<int64_t>(prep.loc_n, e->loc_cap);
        std::vector<int32_t> rec((size_t)n * 2);
        std::vector<double> tt((size_t)n);
        if (n > 0) {
            HIPCHECK(e, hipMemcpy(rec.data(), e->r_locrec.p, (size_t)n * 8, hipMemcpyDeviceToHost));
            HIPCHECK(e, hipMemcpy(tt.data(), e->r_loctime.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        }
        for (int64_t r = 0; r < R; r++)
            for (int64_t i = 0; i < n; i++) {
                e->tau_loc_state[(size_t)r].push_back(rec[(size_t)(i * 2)]);
                e->tau_loc_pop[(size_t)r].push_back(rec[(size_t)(i * 2 + 1)]);
                e->tau_loc_time[(size_t)r].push_back(tt[(size_t)i]);
            }
    } else {
        for (int64_t pn = 0; pn < P; pn++) {  // CheckLockdown (pyx:698-710) on the host totals
            for (int pass = 0; pass < 2; pass++) {
                double ti = (double)h.totalInfectious[(size_t)pn], sz = (double)e->sizes[(size_t)pn];
                bool flip = pass == 0 ? (ti > e->h_startLD[(size_t)pn] * sz && h.lockdownON[(size_t)pn] == 0)
                                      : (ti < e->h_endLD[(size_t)pn] * sz && h.lockdownON[(size_t)pn] == 1);
                if (!flip) continue;
                h.contactDensity[(size_t)pn] = pass == 0 ? e->h_cdAfter[(size_t)pn] : e->h_cdBefore[(size_t)pn];
                h.lockdownON[(size_t)pn] = pass == 0 ? 1 : 0;
                h.swapLockdown += 1;
                for (int64_t r = 0; r < R; r++) {
                    e->tau_loc_state[(size_t)r].push_back(pass == 0 ? 1 : 0);
                    e->tau_loc_pop[(size_t)r].push_back(pn);
                    e->tau_loc_time[(size_t)r].push_back(h.currentTime);
                }
            }
        }
        rates_nonzero = h.globalInfectious != 0;   // an infected host always has a positive total event rate unless every rate is 0
        for (int64_t pn = 0; pn < P && !rates_nonzero; pn++)
            for (int64_t sn = 0; sn < S; sn++)
                if (e->suscepCumul[(size_t)sn] * (double)h.susceptible[(size_t)(pn * S + sn)] != 0.0) rates_nonzero = true;
        h.totalRate = std::nan("");
        h.totalMigrationRate = std::nan("");
    }
    e->dev_state_valid = false;  // the occupancy lists are not maintained by the tau path
    for (int64_t pn = 0; pn < P; pn++)
        if (e->sizes[(size_t)pn] >= ((int64_t)1 << 31)) return fail(e, VGX_ERR_ARG, "vgx_simulate_tau: population sizes must be below 2^31");
    if (e->C > 256 && (e->CB > 16 || S > 64)) return fail(e, VGX_ERR_CLASSES, "vgx_simulate_tau: more than 16 transmission classes together with more than 256 rate classes is not supported");
    const bool start_ok = rates_nonzero && h.globalInfectious != 0;
    // the same guard for the state a Restart restores: does an infected host of the initial state have any event rate?
    bool rates_nonzero_initial = false;
    for (int64_t pn = 0; pn < P && !rates_nonzero_initial; pn++) {
        for (int64_t hn = 0; hn < H && !rates_nonzero_initial; hn++) {
            if (h.initial_infectious[(size_t)(pn * H + hn)] == 0) continue;
            // any positive recovery / sampling / mutation / transmission rate of the haplotype's class makes tEventHapPopRate,
            // hence totalRate, non-zero (transmission additionally needs a susceptible host; a model without the other
            // three rates and without susceptibles has nothing left to simulate either way)
            if (e->h_class_pos[(size_t)e->cls[(size_t)hn]]) rates_nonzero_initial = true;
        }
        for (int64_t sn = 0; sn < S; sn++)
            if (e->suscepCumul[(size_t)sn] * (double)h.initial_susceptible[(size_t)(pn * S + sn)] != 0.0) rates_nonzero_initial = true;
    }

    lap("PrepareParameters");
    // ---- device arrays ----
    // multievent rows (num > 0 only): at most a few per occupied compartment and step; sized from the start state with
    // room for the epidemic to grow, within 2^27 rows (6 GiB) per replicate; a run that still outgrows it fails loudly
    const int64_t rows_per_step = 16 * occupied + 4 * P * S * S + 4096;
    const int64_t mev_max = ((int64_t)1 << 28) / std::max<int64_t>(R, 1);   // 12 GiB of rows over all replicates
    int64_t mev_cap = o.record_events
        ? std::max<int64_t>(1, std::min<int64_t>(mev_max / 2,
                                                 std::max<int64_t>((int64_t)1 << 22, std::min<int64_t>(iterations, 1 << 20) * rows_per_step)))
        : 0;   // doubled on demand (a try whose rows do not fit is run again), up to mev_max
    const size_t nF = 11;  // int32 flag arrays
    const int64_t Ppad = (P + 31) / 32 * 32;
    rc = 0;
    rc |= ensure(e, e->r_locrec, (size_t)(R * VGX_LOC_CAP * 2) * 4);
    rc |= ensure(e, e->r_loctime, (size_t)(R * VGX_LOC_CAP) * 8);
    rc |= ensure(e, e->t_I, (size_t)(R * P * H) * 4);
    rc |= ensure(e, e->t_S, (size_t)(R * P * S) * 8);
    rc |= ensure(e, e->t_I8, (size_t)(R * P * H) + 64);
    // mode of the tries: sparse (no dense delta arrays; the default), or dense with the fused checks (reserved[1] = 2), or
    // dense with the bounds check as a pass of its own (reserved[1] = 1); the dense arrays are allocated when first needed
    const bool sparse_default = !(o.reserved[1] == 1 || o.reserved[1] == 2);
    bool dense_ready = false;
    auto ensure_dense = [&]() -> int {
        if (dense_ready) return 0;
        int r2 = ensure(e, e->t_dChk, (size_t)(R * P * H) * 4) | ensure(e, e->t_dApp, (size_t)(R * P * H) * 4);
        dense_ready = r2 == 0;
        return r2;
    };
    if (!sparse_default) rc |= ensure_dense();
    rc |= ensure(e, e->t_dChkTot, (size_t)(R * P) * 8);
    // queue of the compartments that may draw events in a try: an eighth of the compartments to begin with, grown on demand
    const int64_t q_shards = vgxi_tau_queue_shards(H, P), q_shard_max = vgxi_tau_queue_shard_max(H);
    int64_t q_scap = std::min<int64_t>(q_shard_max, std::max<int64_t>(256, q_shard_max / 8));
    rc |= ensure(e, e->t_q, (size_t)(R * q_shards * q_scap) * 8);
    rc |= ensure(e, e->t_qn, (size_t)(R * q_shards) * 8);
    rc |= ensure(e, e->t_dSi, (size_t)(R * P * S) * 8);
    rc |= ensure(e, e->t_dTot, (size_t)(R * P) * 8);
    rc |= ensure(e, e->t_totInf, (size_t)(R * P) * 8);
    rc |= ensure(e, e->t_gI, (size_t)R * 8);
    rc |= ensure(e, e->t_cd, (size_t)(R * P) * 8);
    rc |= ensure(e, e->t_lock, (size_t)(R * P) * 4);
    rc |= ensure(e, e->t_F, (size_t)(R * P) * 8);
    rc |= ensure(e, e->t_eff, (size_t)(R * P * P) * 8);
    rc |= ensure(e, e->t_Aeff, (size_t)(R * P * Ppad) * 8);
    rc |= ensure(e, e->t_Gout, (size_t)(R * P * e->CB) * 8);
    rc |= ensure(e, e->t_dS, (size_t)(R * P * S) * 8);
    rc |= ensure(e, e->t_taubits, (size_t)R * 8);
    rc |= ensure(e, e->t_tau, (size_t)R * 8);
    rc |= ensure(e, e->t_time, (size_t)R * 8);
    rc |= ensure(e, e->t_flags, (size_t)R * nF * 4);
    rc |= ensure(e, e->t_counters, (size_t)R * 8 * 8);
    int64_t big_cap = std::min<int64_t>(P * H, (int64_t)1 << 20);   // grown on demand
    rc |= ensure(e, e->t_big, (size_t)(R * big_cap) * 8);
    rc |= ensure(e, e->t_bign, (size_t)R * 8);
    rc |= ensure(e, e->t_res, (size_t)R * 16 * 8);
    const int64_t suspect_cap = std::min<int64_t>(P * H, (int64_t)1 << 18);
    rc |= ensure(e, e->t_susp, (size_t)(R * suspect_cap * 2) * 8);
    int64_t st_size = 64;
    while (st_size < 2 * suspect_cap) st_size *= 2;
    rc |= ensure(e, e->t_stkey, (size_t)(R * st_size) * 8);
    rc |= ensure(e, e->t_stval, (size_t)(R * st_size) * 8);
    rc |= ensure(e, e->t_suspn, (size_t)R * 8);
    rc |= ensure(e, e->t_sieve, (size_t)R * VGX_SIEVE_K * 8);
    rc |= ensure(e, e->t_sievepop, (size_t)(R * P) * VGX_SIEVE_K * 8);
    rc |= ensure(e, e->t_sieveskip, (size_t)R * 8);
    rc |= ensure(e, e->t_cnttry, (size_t)R * 8 * 8);
    rc |= ensure(e, e->t_cntpop, (size_t)(R * P) * 8 * 8);
    rc |= ensure(e, e->t_mev, (size_t)(R * std::max<int64_t>(mev_cap, 1) * 6) * 8);
    rc |= ensure(e, e->t_mevn, (size_t)R * 8);
    rc |= ensure(e, e->t_mevbase, (size_t)R * 8);
    rc |= ensure(e, e->t_locn, (size_t)R * 8);
    if (e->h_has_mig && !e->h_mig_uniform) rc |= ensure(e, e->t_migIn, (size_t)(R * P * H) * 8);
    if (e->h_mig_uniform) { rc |= ensure(e, e->t_colT, (size_t)(R * H) * 8); rc |= ensure(e, e->t_colTW, (size_t)(R * H) * 8); }
    rc |= ensure(e, e->t_mutHi, (size_t)(e->d.sites > 6 ? R * P * H : 1) * 8);   // tiled drift, first pass (vgx_tau_muthigh_kernel)
    int64_t inc_cap = std::max<int64_t>((int64_t)1 << 22, P * H / 8) / VGX_INC_SHARDS * VGX_INC_SHARDS;   // grown on demand
    rc |= ensure(e, e->t_inc, (size_t)(R * inc_cap) * 8);
    rc |= ensure(e, e->t_incn, (size_t)R * VGX_INC_SHARDS * 8);
    rc |= ensure(e, e->t_migcdf, (size_t)(R * P * e->CB * P * S) * 8);
    {
        std::vector<double> cum;
        double acc = 0.0;
        for (int64_t s2 = 0; s2 < e->d.sites && s2 < 16; s2++)
            for (int i = 0; i < 3; i++) { acc += e->h_mutp[s2][i]; cum.push_back(acc); }
        if (cum.empty()) cum.push_back(0.0);
        rc |= upload(e, e->t_mutcum, cum.data(), cum.size());
    }
    rc |= upload(e, e->r_seeds, e->seeds.data(), e->seeds.size());
    if (rc) return VGX_ERR_HIP;
    lap("device allocations");
    HIPCHECK(e, hipMemset(e->t_incn.p, 0, (size_t)R * VGX_INC_SHARDS * 8));
    HIPCHECK(e, hipMemset(e->t_stkey.p, 0, (size_t)(R * st_size) * 8));   // try counter 0: every slot reads as empty
    HIPCHECK(e, hipMemset(e->t_dChkTot.p, 0, (size_t)(R * P) * 8));
    HIPCHECK(e, hipMemset(e->t_qn.p, 0, (size_t)(R * q_shards) * 8));
    HIPCHECK(e, hipMemset(e->t_dSi.p, 0, (size_t)(R * P * S) * 8));
    HIPCHECK(e, hipMemset(e->t_dTot.p, 0, (size_t)(R * P) * 8));
    HIPCHECK(e, hipMemset(e->t_counters.p, 0, (size_t)R * 64));
    HIPCHECK(e, hipMemset(e->t_bign.p, 0, (size_t)R * 8));
    HIPCHECK(e, hipMemset(e->t_suspn.p, 0, (size_t)R * 8));
    HIPCHECK(e, hipMemset(e->t_sieve.p, 0, (size_t)R * VGX_SIEVE_K * 8));
    HIPCHECK(e, hipMemset(e->t_sievepop.p, 0, (size_t)(R * P) * VGX_SIEVE_K * 8));
    HIPCHECK(e, hipMemset(e->t_sieveskip.p, 0, (size_t)R * 8));
    HIPCHECK(e, hipMemset(e->t_cnttry.p, 0, (size_t)R * 64));
    HIPCHECK(e, hipMemset(e->t_cntpop.p, 0, (size_t)(R * P) * 64));
    HIPCHECK(e, hipMemset(e->t_mevn.p, 0, (size_t)R * 8));
    HIPCHECK(e, hipMemset(e->t_mevbase.p, 0, (size_t)R * 8));
    HIPCHECK(e, hipMemset(e->t_locn.p, 0, (size_t)R * 8));
    HIPCHECK(e, hipMemset(e->t_flags.p, 0, (size_t)R * nF * 4));
    std::vector<int32_t> lock32((size_t)P);
    for (int64_t pn = 0; pn < P; pn++) lock32[(size_t)pn] = (int32_t)h.lockdownON[(size_t)pn];
    for (int64_t r = 0; r < R; r++) {
        if (!staged) {
            rc = tau_upload_state(e, r, h.infectious, h.susceptible);
            if (rc) return rc;
        }
        HIPCHECK(e, hipMemcpy((double *)e->t_cd.p + r * P, h.contactDensity.data(), (size_t)P * 8, hipMemcpyHostToDevice));
        HIPCHECK(e, hipMemcpy((int32_t *)e->t_lock.p + r * P, lock32.data(), (size_t)P * 4, hipMemcpyHostToDevice));
    }

    lap("memsets + state upload");
    VgxTauArgs a{};
    a.p = e->dp;
    a.R = R;
    a.I = (int32_t *)e->t_I.p; a.I8 = (uint8_t *)e->t_I8.p; a.S = (int64_t *)e->t_S.p; a.dChk = (int32_t *)e->t_dChk.p; a.dApp = (int32_t *)e->t_dApp.p;
    a.dSi = (int64_t *)e->t_dSi.p; a.dTot = (int64_t *)e->t_dTot.p; a.totInf = (int64_t *)e->t_totInf.p;
    a.gI = (int64_t *)e->t_gI.p; a.cd = (double *)e->t_cd.p; a.lockON = (int32_t *)e->t_lock.p; a.F = (double *)e->t_F.p;
    a.effMig = (double *)e->t_eff.p; a.Aeff = (double *)e->t_Aeff.p; a.Gout = (double *)e->t_Gout.p;
    a.dS = (double *)e->t_dS.p; a.tau_bits = (unsigned long long *)e->t_taubits.p; a.tau = (double *)e->t_tau.p;
    a.time_now = (double *)e->t_time.p;
    int32_t *fl = (int32_t *)e->t_flags.p;
    a.active = fl; a.ok = fl + R; a.accepted = fl + 2 * R; a.grow = fl + 3 * R; a.retry = fl + 4 * R;   // accepted, grow: one copy per try
    {   // the host's pinned mirror of accepted / grow (written by the decide kernel) and of the finish kernel's record: read after a
        // stream synchronisation, no copy in between
        const size_t need = (size_t)R * 3 * 4 + 64 + (size_t)R * 16 * 8;
        if (e->pin_tau_bytes < need) {
            if (e->pin_tau) (void)hipHostFree(e->pin_tau);
            e->pin_tau = nullptr; e->pin_tau_bytes = 0;
            HIPCHECK(e, hipHostMalloc(&e->pin_tau, need, hipHostMallocDefault));
            e->pin_tau_bytes = need;
        }
        memset(e->pin_tau, 0, need);
        void *dp = nullptr;
        HIPCHECK(e, hipHostGetDevicePointer(&dp, e->pin_tau, 0));
        a.host_flags = (int32_t *)dp;
        a.host_res = (int64_t *)((char *)dp + (((size_t)R * 3 * 4 + 63) & ~(size_t)63));
    }
    const int32_t *pin_flags = (const int32_t *)e->pin_tau;
    const int64_t *pin_res = (const int64_t *)((const char *)e->pin_tau + (((size_t)R * 3 * 4 + 63) & ~(size_t)63));
    a.step = fl + 5 * R; a.error = fl + 6 * R; a.attempt = fl + 7 * R; a.eff_dirty = fl + 8 * R; a.deciding = fl + 9 * R;
    a.spec = fl + 10 * R; a.gate = 0;
    a.Ppad = (int32_t)Ppad;
    {
        std::vector<int32_t> ones((size_t)R, 1);
        HIPCHECK(e, hipMemcpy(a.eff_dirty, ones.data(), (size_t)R * 4, hipMemcpyHostToDevice));
    }
    a.seeds = (const int64_t *)e->r_seeds.p;
    a.has_mig = e->h_has_mig ? 1 : 0;
    a.mut_uniform = (e->h_mut_uniform && sites_ok16(e)) ? 1 : 0;
    memcpy(a.mutp, e->h_mutp, sizeof(a.mutp));
    a.mut_total = e->h_mut_total;
    a.mutcum = (const double *)e->t_mutcum.p;
    a.migcdf = (double *)e->t_migcdf.p;
    a.migIn = (double *)e->t_migIn.p;
    a.mutHi = (double *)e->t_mutHi.p;
    {   // high sites (the first sites - 6) all with one rate and equally likely derived states?
        const int nh = (int)e->d.sites - 6;
        bool same = a.mut_uniform && nh > 0 && nh <= 4;
        for (int s2 = 0; s2 < nh && same; s2++)
            for (int i = 0; i < 3; i++)
                if (e->h_mutp[s2][i] != e->h_mutp[0][0]) same = false;
        a.mutHi_int = same ? 1 : 0;
        a.mutHi_rate = same ? e->h_mutp[0][0] : 0.0;
    }
    a.mig_uniform = e->h_mig_uniform ? 1 : 0; a.mig_b = e->h_mig_b; a.mig_d = e->h_mig_d;
    a.colT = (double *)e->t_colT.p; a.colTW = (double *)e->t_colTW.p;
    a.inc = (int64_t *)e->t_inc.p; a.inc_cap = inc_cap; a.inc_shards = vgxi_tau_inc_shards(H, P); a.inc_n = (unsigned long long *)e->t_incn.p;
    a.counters = (int64_t *)e->t_counters.p; a.cnt_try = (int64_t *)e->t_cnttry.p; a.cnt_pop = (unsigned long long *)e->t_cntpop.p;
    {   // the front pass of a try (vgx_tau_front_kernel): the tabulated scan's shapes, sparse mode
        const char *nf = getenv("VGX_TAU_NO_FRONT");
        a.front_cap = 512;
        a.front_on = (sparse_default && e->C <= 16 && e->CB <= 16 && (H & 15) == 0 && !(nf && nf[0] == '1')) ? 1 : 0;
        int rcf = ensure(e, e->t_front, (size_t)(R * P) * (size_t)a.front_cap * 8);
        if (!rcf) rcf = ensure(e, e->t_frontn, (size_t)(R * P) * 4 + 64);
        if (rcf) return rcf;
        HIPCHECK(e, hipMemset(e->t_frontn.p, 0, (size_t)(R * P) * 4));
        a.front = (int64_t *)e->t_front.p; a.front_n = (unsigned int *)e->t_frontn.p;
    }
    a.big = (int64_t *)e->t_big.p; a.big_cap = big_cap; a.big_n = (unsigned long long *)e->t_bign.p;
    a.res = (int64_t *)e->t_res.p;
    a.suspect = (int64_t *)e->t_susp.p; a.suspect_cap = suspect_cap; a.suspect_n = (unsigned long long *)e->t_suspn.p;
    a.dense_check = o.reserved[1] == 1 ? 1 : 0;   // validation: the bounds check as one dense pass over all compartments
    a.sparse = sparse_default ? 1 : 0;
    a.gen = 0;
    a.st_key = (unsigned long long *)e->t_stkey.p; a.st_val = (long long *)e->t_stval.p; a.st_size = st_size;
    a.dChkTot = (int64_t *)e->t_dChkTot.p;
    a.q = (int64_t *)e->t_q.p; a.q_cap = q_shards * q_scap; a.q_shards = q_shards; a.q_n = (unsigned long long *)e->t_qn.p;
    // A compartment's events are drawn by ONE lane of the events kernel (a draw of their number, then one by one) up to this
    // mean, by a group of lanes of vgx_tau_draw_big_kernel (one Poisson draw per channel) from it on.  The lane's way is far
    // cheaper per compartment but its time grows with the mean, and the slowest lane holds its wavefront: with few
    // compartments (nothing else to overlap with) the switch comes earlier.  Same joint law either way.
    a.big_lam = P * H * R <= ((int64_t)1 << 18) ? VGX_TAU_BIG_SMALL : VGX_TAU_BIG;
    {   // tests: the thresholds of large models on a small one (so that its draws go through the one-draw-per-kind form)
        const char *th = getenv("VGX_TAU_LARGE_MODEL_THRESHOLDS");
        if (th && th[0] == '1') a.big_lam = VGX_TAU_BIG;
    }
    // enough blocks of the events kernel to fill the chip whatever the number of shards (mid-size models have few)
    a.ev_split = (int32_t)std::max<int64_t>(1, std::min<int64_t>(q_shard_max / 64, 4096 / std::max<int64_t>(1, q_shards * R)));
    a.sieve = (double *)e->t_sieve.p; a.sieve_pop = (double *)e->t_sievepop.p; a.sieve_skipped = (int64_t *)e->t_sieveskip.p;
    // vgx_run_opts.reserved[0] = 1: run every try of the halving loop; with few compartments no try is ever a certain rejection
    a.sieve_on = (o.reserved[0] == 1 || P * H < 32768) ? 0 : 1;
    {   // low sites (the last min(sites, 6)): equally likely derived states at each of them?  one common rate?
        const int ns = (int)e->d.sites, low = ns < 6 ? ns : 6, nh = ns - low;
        bool flat = a.mut_uniform && low >= 2 && ns <= 10 && e->t_mutHi.p != nullptr && e->C <= 256 && e->CB <= 16, same = true;
        for (int s2 = nh; s2 < ns && flat; s2++) {
            if (e->h_mutp[s2][0] != e->h_mutp[s2][1] || e->h_mutp[s2][1] != e->h_mutp[s2][2]) flat = false;
            if (e->h_mutp[s2][0] != e->h_mutp[nh][0]) same = false;
        }
        // the fast drift kernel's inner loop loads its inputs unconditionally an iteration ahead: high-site sums as integers,
        // migration (if any) through the two column sums; the other forms take the general tiled kernel
        flat = flat && (nh == 0 || a.mutHi_int) && (!a.has_mig || a.mig_uniform);
        a.mutlow_fast = flat ? 1 : 0;
        a.mutlow_same = (flat && same) ? 1 : 0;
        // the drift pass on the one-byte counts (vgx_tau_drift8_kernel): the fast form's models from seven sites on with one
        // rate class and one rate for the low sites (VGX_TAU_NO_BYTE_DRIFT=1: the two-pass form, for comparisons)
        const char *nb8 = getenv("VGX_TAU_NO_BYTE_DRIFT");
        a.use8 = (flat && same && nh >= 1 && e->C == 1 && S <= 64 && !(nb8 && nb8[0] == '1')) ? 1 : 0;
        a.nt8 = ns > 8 ? 1 << (2 * (ns - 8)) : 1;
        {
            int rc8 = ensure(e, e->t_tmax8, (size_t)(R * P * a.nt8) * 4 + 64);
            if (rc8) return rc8;
            a.tmax8 = (unsigned int *)e->t_tmax8.p;
        }
        if (a.use8) {   // lists of the occupied compartments for sparse states (vgx_tau_listscan_kernel), filled by the drift pass
            a.occ_nreg = a.nt8 * VGX_D8_WAVES;
            int rco = ensure(e, e->t_occ, (size_t)(R * P) * (size_t)a.occ_nreg * VGX_OCC_CAP * 4);
            if (!rco) rco = ensure(e, e->t_occn, (size_t)(R * P) * (size_t)a.occ_nreg * 4 + 64);
            if (!rco) rco = ensure(e, e->t_occpop, (size_t)(R * P) * 8 + 64);
            if (rco) return rco;
            HIPCHECK(e, hipMemset(e->t_occpop.p, 0, (size_t)(R * P) * 8));
            a.occ = (int32_t *)e->t_occ.p; a.occ_n = (unsigned int *)e->t_occn.p; a.occ_pop = (unsigned long long *)e->t_occpop.p;
            // the drift pass over those lists (vgx_tau_drift8s_*)
            rco = ensure(e, e->t_tIpt, (size_t)(R * P * a.nt8) * 8 + 64);
            if (!rco) rco = ensure(e, e->t_d8spk, (size_t)(R * P) * 64 + 64);
            if (!rco) rco = ensure(e, e->t_d8sbc, (size_t)R * 64 + 64);
            if (!rco) rco = ensure(e, e->t_d8sovf, (size_t)(R * P) * (size_t)a.occ_nreg * 4 + 64);
            if (!rco) rco = ensure(e, e->t_d8smax, (size_t)(R * P) * (size_t)a.occ_nreg * 4 + 64);
            if (!rco) rco = ensure(e, e->t_d8stile, (size_t)(R * 2 * a.nt8) * 8 + 64);
            if (rco) return rco;
            a.tI_pt = (unsigned long long *)e->t_tIpt.p; a.d8s_pk = (double *)e->t_d8spk.p; a.d8s_bc = (unsigned long long *)e->t_d8sbc.p;
            a.d8s_tile = (double *)e->t_d8stile.p; a.d8s_ovf = (int32_t *)e->t_d8sovf.p; a.d8s_regmax = (int32_t *)e->t_d8smax.p;
        }
        a.hist = nullptr;
        if (flat && a.sieve_on && e->C <= 8) {   // (VGX_HIST_CMAX classes x 64 sizes per population)
            int rch = ensure(e, e->t_hist, (size_t)(R * P * e->C * 64) * 4);
            if (rch) return rch;
            HIPCHECK(e, hipMemset(e->t_hist.p, 0, (size_t)(R * P * e->C * 64) * 4));
            a.hist = (unsigned int *)e->t_hist.p;
        }
    }
    {   // the drift kernel's blocks write their parts of the susceptible drift into their own slots
        a.ds_nb = vgxi_tau_drift_blocks(&a);
        int rcd = ensure(e, e->t_dSpart, (size_t)(R * P * a.ds_nb * S) * 8);
        if (rcd) return rcd;
        a.dS_part = (double *)e->t_dSpart.p;
    }
    a.mev = (int64_t *)e->t_mev.p; a.mev_cap = mev_cap;
    e->tau_mev_cap = mev_cap;
    a.mev_n = (unsigned long long *)e->t_mevn.p; a.mev_base = (unsigned long long *)e->t_mevbase.p;
    a.loc_n = (unsigned long long *)e->t_locn.p; a.loc_rec = (int32_t *)e->r_locrec.p; a.loc_time = (double *)e->r_loctime.p;

    // ---- per-replicate host bookkeeping ----
    std::vector<double> tnow((size_t)R, h.currentTime), tau_h;
    std::vector<int64_t> ev_ptr((size_t)R, ev_ptr_start), att((size_t)R, 0), good((size_t)R, h.good_attempt), gI((size_t)R, h.globalInfectious);
    std::vector<int64_t> base_cnt = {h.bCounter, h.dCounter, h.sCounter, h.mCounter, h.iCounter, h.migPlus, h.swapLockdown, 0};
    std::vector<std::vector<int64_t>> cnt0((size_t)R, base_cnt);  // counters before this call / after a restart
    std::vector<int64_t> cnt((size_t)R * 8, 0);
    std::vector<int32_t> running((size_t)R, 0), finished((size_t)R, 0), step_h((size_t)R, 0), att32((size_t)R, 0), acc_h, err_h;
    std::vector<int64_t> restarts((size_t)R, 0), steps_done((size_t)R, 0), swaps_kept((size_t)R, 0);
    e->tau_log.assign((size_t)R, {});
    e->tau_ev_ptr0.assign((size_t)R, ev_ptr_start);
    std::vector<unsigned long long> mevn((size_t)R, 0);
    for (int64_t r = 0; r < R; r++) running[(size_t)r] = (attempts > 0 && start_ok) ? 1 : 0;
    std::vector<int32_t> fresh((size_t)R, 1);  // attempt just opened: the pyx:2311 guard applies
    std::vector<unsigned long long> susp_h;
    float ms_total = 0.f;
    int64_t launches = 0;
    std::vector<int32_t> dev_active, dev_step, dev_att;   // what the device holds (empty: nothing uploaded yet)
    std::vector<double> dev_time;
    const bool has_tl = !(time == -1.0f);
    auto sC_of = [&](int64_t r) { return cnt0[(size_t)r][2] + cnt[(size_t)r * 8 + 2]; };
    int64_t guard = 0;
    // occupied-compartment lists (sparse states): possible with the byte drift pass and the front pass; the estimate is the count of the
    // uploaded state, then what the drift pass of the last step counted (the largest replicate)
    const char *nol = getenv("VGX_TAU_NO_OCCLIST");
    const bool occ_lists_ok = a.use8 && a.front_on && a.occ != nullptr && !(nol && nol[0] == '1');
    const char *ddr = getenv("VGX_TAU_DENSE_DRIFT");     // comparisons: vgx_tau_drift8_kernel also on sparse states
    const bool dense_drift = ddr && ddr[0] == '1';
    int sparse_ban = 0, sparse_ban_len = 32;      // steps for which the drift pass stays dense; the next such span
    const bool occ_lists_banned = false;
    int64_t occ_est = occupied;
    int64_t tries_total = 0, tries_lists = 0;
    const char *nfo = getenv("VGX_TAU_NO_FRONT_ALONE");
    const bool front_split = R == 1 && a.front_on && !(nfo && nfo[0] == '1');   // (several replicates: their tries end at different places)
    // ... and whole rounds of a step without the host in between (VgxTauArgs.spec / gate); VGX_TAU_SPEC=0: one try per synchronisation as
    // before, VGX_TAU_SPEC=k: k front passes per round
    int spec_k = 6;
    bool spec_adapt = true;      // the number of front passes per round follows the last step's (rejected tries + the one that ran + one to spare)
    if (const char *sk = getenv("VGX_TAU_SPEC")) { spec_k = atoi(sk); spec_adapt = false; }
    const bool spec_rounds = front_split && spec_k > 0;
    int64_t host_syncs = 0;
    bool front_done = false;
    bool i8_dirty = true;    // I8 does not mirror I (start of the call, after a Restart's upload, after a dense try)
    // Small models: the whole step loop on the device, one workgroup per replicate (vgx_taus.hip).  VGX_TAU_STEP_KERNELS=1 and the
    // test switches of the step kernels (dense validation modes, the large-model draw thresholds) keep the step kernels.
    const int64_t slog_cap = std::max<int64_t>(ev_size - ((ev_ptr_start <= 100 && iterations > 100) ? 0 : ev_ptr_start), 1);
    // One workgroup (one CU) runs a replicate's whole loop: that wins where a step is launch-bound (up to ~2000 compartments at any
    // ensemble size) or where there are replicates to fill the chip with; few replicates of a larger model are faster spread over the
    // chip by the step kernels.  Measured in round 4 (tools/probe_tau_single.py, steps/s of ONE trajectory, step kernels / loop):
    // 256 compartments 8.9e3 / 5.3e4, 1280: 8.6e3 / 1.46e4 (1.2e6 infected: 7.1e3 / 6.5e3), 2048: 8.7e3 / 9.3e3, 4096: 8.3e3 / 3.9e3,
    // 8192: 8.2e3 / 3.1e3; at 32 replicates 4096 compartments are level (1.4e5 / 1.2e5), from 128 on the loop leads everywhere.
    const int64_t n_channels = P * H * (2 + 3 * e->d.sites + S + (P - 1) * S) + P * S * S;
    bool use_small = P * H <= VGX_TAUS_MAX_CELLS && P <= VGX_TAUS_MAX_P && S <= VGX_TAUS_MAX_S && e->d.sites <= 15 && sparse_default &&
                     e->C <= VGX_TAUS_MAX_C && e->CB <= VGX_TAUS_MAX_CB && (n_channels <= 4096 || P * H <= 2048 || R >= 32) &&
                     vgx_taus_lds_bytes(P, H, S, e->C, e->CB) <= 150 * 1024 && (double)R * (double)slog_cap * 24.0 <= 8e9;
    {
        const char *fs = getenv("VGX_TAU_STEP_KERNELS"), *th = getenv("VGX_TAU_LARGE_MODEL_THRESHOLDS");
        if ((fs && fs[0] == '1') || (th && th[0] == '1')) use_small = false;
        if (fs && fs[0] == '0' && P * H <= VGX_TAUS_MAX_CELLS && P <= VGX_TAUS_MAX_P && S <= VGX_TAUS_MAX_S && e->d.sites <= 15 && sparse_default &&
            e->C <= VGX_TAUS_MAX_C && e->CB <= VGX_TAUS_MAX_CB && vgx_taus_lds_bytes(P, H, S, e->C, e->CB) <= 150 * 1024)
            use_small = true;      // (VGX_TAU_STEP_KERNELS=0: the on-device loop wherever it can run, for tests and comparisons)
    }
    if (use_small) {
        std::vector<int32_t> i32((size_t)(P * H));
        for (int64_t i = 0; i < P * H; i++) i32[(size_t)i] = (int32_t)h.initial_infectious[(size_t)i];
        rc = upload(e, e->t_iI, i32.data(), i32.size());
        rc |= upload(e, e->t_iS, h.initial_susceptible.data(), h.initial_susceptible.size());
        rc |= ensure(e, e->t_slog, (size_t)(R * slog_cap * 3) * 8);
        rc |= ensure(e, e->t_sres, (size_t)(R * 24) * 8);
        if (rc) return rc;
        VgxTausArgs ta{};
        ta.p = e->dp; ta.R = R;
        ta.I = a.I; ta.S = a.S; ta.totInf = a.totInf; ta.cd = a.cd; ta.lock = a.lockON;
        ta.i_I = (const int32_t *)e->t_iI.p; ta.i_S = (const int64_t *)e->t_iS.p;
        ta.seeds = a.seeds;
        ta.iterations = iterations; ta.sample_size = sample_size; ta.attempts = attempts; ta.time = time;
        ta.start_ok = start_ok ? 1 : 0; ta.rates_nonzero_initial = rates_nonzero_initial ? 1 : 0;
        ta.ev_ptr0 = ev_ptr_start; ta.ev_size = ev_size;
        ta.t0 = h.currentTime; ta.gI0 = h.globalInfectious; ta.good0 = h.good_attempt;
        for (int i = 0; i < 8; i++) ta.base_cnt[i] = base_cnt[(size_t)i];
        ta.mut_uniform = a.mut_uniform;
        memcpy(ta.mutp, a.mutp, sizeof(ta.mutp));
        ta.mev = a.mev; ta.mev_cap = mev_cap;
        ta.slog = (int64_t *)e->t_slog.p; ta.slog_cap = slog_cap;
        ta.loc_rec = a.loc_rec; ta.loc_time = a.loc_time; ta.loc_n = a.loc_n;
        ta.res = (int64_t *)e->t_sres.p;
        HIPCHECK(e, hipEventRecord(e->ev0, e->stream));
        HIPCHECK(e, vgxi_launch_taus(&ta, e->stream));
        HIPCHECK(e, hipEventRecord(e->ev1, e->stream));
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        HIPCHECK(e, hipEventElapsedTime(&ms_total, e->ev0, e->ev1));
        launches = 1;
        std::vector<int64_t> res((size_t)R * 24);
        HIPCHECK(e, hipMemcpy(res.data(), e->t_sres.p, res.size() * 8, hipMemcpyDeviceToHost));
        tau_h.assign((size_t)R, 0.0);
        std::vector<int64_t> sl;
        for (int64_t r = 0; r < R; r++) {
            const int64_t *o2 = &res[(size_t)r * 24];
            const int64_t er = o2[TS_ERROR];
            if (er == VGX_ERR_CAPACITY)
                return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: replicate " + std::to_string(r) + ": multievent buffer full (" + std::to_string(mev_cap) +
                                                     " rows per replicate; pass record_events=0 for large runs)");
            if (er == 7) return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: replicate " + std::to_string(r) + ": lockdown log full (" + std::to_string(VGX_LOC_CAP) + " switches per call)");
            if (er) {
                double tl_, tn_;
                memcpy(&tl_, &o2[TS_TAU], 8); memcpy(&tn_, &o2[TS_TIME], 8);
                return fail(e, VGX_ERR_LOOP_GUARD, "vgx_simulate_tau: replicate " + std::to_string(r) + (er == 6 ? ": step loop guard" : ": tau underflow in the halving loop") +
                                                       " (step " + std::to_string(o2[TS_STEPS]) + ", tries " + std::to_string(o2[TS_TRIES]) + ", tau " + std::to_string(tl_) +
                                                       ", time " + std::to_string(tn_) + ", infected " + std::to_string(o2[TS_GI]) + ")");
            }
            memcpy(&tau_h[(size_t)r], &o2[TS_TAU], 8);
            if (o2[TS_STEPS] == 0 && o2[TS_RESTARTS] == 0) tau_h[(size_t)r] = h.tau_l;
            if (timing && r == 0) fprintf(stderr, "vgx_simulate_tau: on-device loop: %lld steps, %lld tries, %.3f ms\n", (long long)o2[TS_STEPS], (long long)o2[TS_TRIES], (double)ms_total);
            gI[(size_t)r] = o2[TS_GI];
            for (int i = 0; i < 8; i++) { cnt[(size_t)r * 8 + i] = o2[TS_CNT0 + i]; cnt0[(size_t)r][(size_t)i] = 0; }
            ev_ptr[(size_t)r] = o2[TS_EVPTR]; att[(size_t)r] = o2[TS_ATT]; good[(size_t)r] = o2[TS_GOOD];
            restarts[(size_t)r] = o2[TS_RESTARTS]; steps_done[(size_t)r] = o2[TS_STEPS];
            mevn[(size_t)r] = (unsigned long long)o2[TS_MEVROWS];
            memcpy(&tnow[(size_t)r], &o2[TS_TIME], 8);
            e->tau_ev_ptr0[(size_t)r] = o2[TS_EVPTR0];
            const int64_t n = o2[TS_EVPTR] - o2[TS_EVPTR0];
            sl.resize((size_t)std::max<int64_t>(n, 0) * 3);
            if (n > 0) HIPCHECK(e, hipMemcpy(sl.data(), (int64_t *)e->t_slog.p + r * slog_cap * 3, (size_t)n * 24, hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < n; k++) {
                double t;
                memcpy(&t, &sl[(size_t)k * 3], 8);
                e->tau_log[(size_t)r].push_back({t, sl[(size_t)k * 3 + 1] & (((int64_t)1 << 56) - 1), sl[(size_t)k * 3 + 2], (int32_t)((uint64_t)sl[(size_t)k * 3 + 1] >> 56)});
            }
        }
    } else
    while (true) {
        // loop condition (pyx:2312) / end of attempt (pyx:2331-2335)
        bool any = false;
        for (int64_t r = 0; r < R; r++) {
            if (finished[(size_t)r]) continue;
            if (attempts <= 0) { finished[(size_t)r] = 1; continue; }
            bool go = running[(size_t)r] && ev_ptr[(size_t)r] < ev_size && (sample_size == -1 || sC_of(r) < sample_size) &&
                      (!has_tl || tnow[(size_t)r] < (double)time) && (fresh[(size_t)r] || gI[(size_t)r] != 0);
            fresh[(size_t)r] = 0;
            if (go) { any = true; continue; }
            running[(size_t)r] = 0;
            if (ev_ptr[(size_t)r] <= 100 && iterations > 100) {  // Restart (pyx:714-738)
                restarts[(size_t)r] += 1;
                rc = tau_upload_state(e, r, h.initial_infectious, h.initial_susceptible);
                if (rc) return rc;
                i8_dirty = true;
                occ_est = occupied;   // (the start state again)
                {   // the lockdown records of the failed attempt stay (Restart does not clear `loc`); then CheckLockdown for
                    // every population on the restored totals at time 0 (pyx:736-737), whose switches change the contact
                    // densities the next attempt starts with
                    unsigned long long ln_r = 0;
                    HIPCHECK(e, hipMemcpy(&ln_r, a.loc_n + r, 8, hipMemcpyDeviceToHost));
                    const int64_t nrec = std::min<int64_t>((int64_t)ln_r, VGX_LOC_CAP);
                    if (nrec > 0) {
                        std::vector<int32_t> rec((size_t)nrec * 2);
                        std::vector<double> tt((size_t)nrec);
                        HIPCHECK(e, hipMemcpy(rec.data(), (int32_t *)e->r_locrec.p + r * VGX_LOC_CAP * 2, (size_t)nrec * 8, hipMemcpyDeviceToHost));
                        HIPCHECK(e, hipMemcpy(tt.data(), (double *)e->r_loctime.p + r * VGX_LOC_CAP, (size_t)nrec * 8, hipMemcpyDeviceToHost));
                        for (int64_t i = 0; i < nrec; i++) {
                            e->tau_loc_state[(size_t)r].push_back(rec[(size_t)(i * 2)]);
                            e->tau_loc_pop[(size_t)r].push_back(rec[(size_t)(i * 2 + 1)]);
                            e->tau_loc_time[(size_t)r].push_back(tt[(size_t)i]);
                        }
                        HIPCHECK(e, hipMemset(a.loc_n + r, 0, 8));
                    }
                    std::vector<double> cd_r((size_t)P);
                    std::vector<int32_t> lk_r((size_t)P);
                    HIPCHECK(e, hipMemcpy(cd_r.data(), (double *)e->t_cd.p + r * P, (size_t)P * 8, hipMemcpyDeviceToHost));
                    HIPCHECK(e, hipMemcpy(lk_r.data(), (int32_t *)e->t_lock.p + r * P, (size_t)P * 4, hipMemcpyDeviceToHost));
                    int64_t flips = 0;
                    for (int64_t pn = 0; pn < P; pn++) {
                        int64_t ti = 0;
                        for (int64_t hn = 0; hn < H; hn++) ti += h.initial_infectious[(size_t)(pn * H + hn)];
                        for (int pass = 0; pass < 2; pass++) {
                            const double sz = (double)e->sizes[(size_t)pn];
                            const bool flip = pass == 0 ? ((double)ti > e->h_startLD[(size_t)pn] * sz && lk_r[(size_t)pn] == 0)
                                                        : ((double)ti < e->h_endLD[(size_t)pn] * sz && lk_r[(size_t)pn] == 1);
                            if (!flip) continue;
                            cd_r[(size_t)pn] = pass == 0 ? e->h_cdAfter[(size_t)pn] : e->h_cdBefore[(size_t)pn];
                            lk_r[(size_t)pn] = pass == 0 ? 1 : 0;
                            e->tau_loc_state[(size_t)r].push_back(pass == 0 ? 1 : 0);
                            e->tau_loc_pop[(size_t)r].push_back(pn);
                            e->tau_loc_time[(size_t)r].push_back(0.0);
                            flips += 1;
                        }
                    }
                    if (flips > 0) {
                        HIPCHECK(e, hipMemcpy((double *)e->t_cd.p + r * P, cd_r.data(), (size_t)P * 8, hipMemcpyHostToDevice));
                        HIPCHECK(e, hipMemcpy((int32_t *)e->t_lock.p + r * P, lk_r.data(), (size_t)P * 4, hipMemcpyHostToDevice));
                        const int32_t one = 1;
                        HIPCHECK(e, hipMemcpy(a.eff_dirty + r, &one, 4, hipMemcpyHostToDevice));
                    }
                    swaps_kept[(size_t)r] += cnt[(size_t)r * 8 + 6] + flips;   // swapLockdown survives a Restart
                }
                HIPCHECK(e, hipMemset((int64_t *)e->t_counters.p + r * 8, 0, 64));
                HIPCHECK(e, hipMemset((unsigned long long *)e->t_mevn.p + r, 0, 8));
                HIPCHECK(e, hipMemset((unsigned long long *)e->t_mevbase.p + r, 0, 8));
                for (int i = 0; i < 8; i++) cnt[(size_t)r * 8 + i] = 0;
                cnt0[(size_t)r] = {0, 0, 0, 0, 0, 0, base_cnt[6] + swaps_kept[(size_t)r], 0};
                tnow[(size_t)r] = 0.0;
                ev_ptr[(size_t)r] = 0;
                e->tau_ev_ptr0[(size_t)r] = 0;
                e->tau_log[(size_t)r].clear();
                int64_t g0 = 0;
                for (int64_t i = 0; i < P * H; i++) g0 += h.initial_infectious[(size_t)i];
                gI[(size_t)r] = g0;
                att[(size_t)r] += 1;
                if (att[(size_t)r] < attempts) {
                    running[(size_t)r] = (g0 != 0 && rates_nonzero_initial) ? 1 : 0;   // pyx:2311 on the restored state
                    fresh[(size_t)r] = 1;
                    if (running[(size_t)r]) any = true;
                    else r -= 1;  // re-evaluate: the attempt ends at once
                } else {
                    finished[(size_t)r] = 1;
                }
            } else {
                good[(size_t)r] = att[(size_t)r] + 1;
                finished[(size_t)r] = 1;
            }
        }
        if (!any) break;
        if (++guard > (int64_t)4 * (iterations + 16) * std::max<int64_t>(attempts, 1)) return fail(e, VGX_ERR_LOOP_GUARD, "vgx_simulate_tau: step loop guard");
        // the device keeps step and time itself (vgx_tau_finish_kernel advances them exactly as the host does below): only
        // what an attempt's end or a Restart changed is uploaded
        for (int64_t r = 0; r < R; r++) att32[(size_t)r] = (int32_t)att[(size_t)r];
        if (running != dev_active) { HIPCHECK(e, hipMemcpy(a.active, running.data(), (size_t)R * 4, hipMemcpyHostToDevice)); dev_active = running; }
        if (step_h != dev_step) { HIPCHECK(e, hipMemcpy(a.step, step_h.data(), (size_t)R * 4, hipMemcpyHostToDevice)); dev_step = step_h; }
        if (att32 != dev_att) { HIPCHECK(e, hipMemcpy(a.attempt, att32.data(), (size_t)R * 4, hipMemcpyHostToDevice)); dev_att = att32; }
        if (tnow != dev_time) { HIPCHECK(e, hipMemcpy(a.time_now, tnow.data(), (size_t)R * 8, hipMemcpyHostToDevice)); dev_time = tnow; }
        HIPCHECK(e, hipEventRecord(e->ev0, e->stream));
        if (a.use8 && i8_dirty) {     // the one-byte counts after an upload / a dense try: one pass over the 4-byte counts
            HIPCHECK(e, hipMemsetAsync(a.tmax8, 0, (size_t)(R * P * a.nt8) * 4, e->stream));
            HIPCHECK(e, vgxi_tau_conv8(&a, e->stream));
            i8_dirty = false;
            launches += 1;
        }
        // a sparse state (at most 1/32 of the compartments occupied when the last step began): the drift pass lists the occupied
        // compartments and the tries' scan and front pass go over the lists
        a.build_occ = a.use_list = (occ_lists_ok && !occ_lists_banned && occ_est >= 0 && occ_est * 32 <= P * H) ? 1 : 0;
        // ... and with uniform migration (the column sums' pass is there to write the lists) the drift pass itself goes over them
        // (unless the last such pass had to form the empty neighbours of too many compartments — a high mutation rate, or a smallest
        // candidate far above what the lineages' mutants bring: the dense pass for a while, then another look)
        if (sparse_ban > 0) sparse_ban -= 1;
        a.drift_sparse = (a.build_occ && a.has_mig && a.mig_uniform && !dense_drift && sparse_ban == 0 && e->d.sites <= 12) ? 1 : 0;   // (12: VGX_D8S_MAX_SITES)
        HIPCHECK(e, vgxi_tau_eff(&a, e->stream));
        HIPCHECK(e, vgxi_tau_prep(&a, e->stream));
        HIPCHECK(e, vgxi_tau_drift(&a, e->stream));
        HIPCHECK(e, vgxi_tau_choose(&a, e->stream));
        launches += 4;
        if (a.sieve_on) { HIPCHECK(e, vgxi_tau_sieve(&a, e->stream)); launches += 2; }
        bool dense_once = false;   // the last try asked for dense delta arrays
        // what a discarded try asks of the host (VgxTauArgs.grow): a larger list / buffer, or the dense delta arrays for the same try
        auto handle_again = [&](int again) -> int {
            if (again & 1) {
                if (inc_cap > ((int64_t)1 << 33) / std::max<int64_t>(R, 1))
                    return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: more than 2^33 individuals change compartment in one leap");
                inc_cap *= 2;
                int rcg = ensure(e, e->t_inc, (size_t)(R * inc_cap) * 8);
                if (rcg) return rcg;
                a.inc = (int64_t *)e->t_inc.p;
                a.inc_cap = inc_cap;
            }
            if (again & 4) {
                if (big_cap >= P * H) return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: list of large compartments full");
                big_cap = std::min<int64_t>(P * H, big_cap * 2);
                int rcg = ensure(e, e->t_big, (size_t)(R * big_cap) * 8);
                if (rcg) return rcg;
                a.big = (int64_t *)e->t_big.p;
                a.big_cap = big_cap;
            }
            if (again & 8) {
                if (q_scap >= q_shard_max) return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: queue of drawing compartments full");
                q_scap = std::min<int64_t>(q_shard_max, q_scap * 2);
                int rcg = ensure(e, e->t_q, (size_t)(R * q_shards * q_scap) * 8);
                if (rcg) return rcg;
                a.q = (int64_t *)e->t_q.p;
                a.q_cap = q_shards * q_scap;
            }
            if (again & 16) {   // multievent rows: a larger buffer, the rows of the accepted steps move over
                if (mev_cap >= mev_max)
                    return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: multievent buffer full (" + std::to_string(mev_cap) +
                                                     " rows per replicate; pass record_events=0 for large runs)");
                const int64_t new_cap = std::min<int64_t>(mev_max, mev_cap * 2);
                DevBuf nb;
                int rcg = ensure(e, nb, (size_t)(R * new_cap * 6) * 8);
                if (rcg) return rcg;
                std::vector<unsigned long long> base_h((size_t)R);
                HIPCHECK(e, hipMemcpy(base_h.data(), a.mev_base, (size_t)R * 8, hipMemcpyDeviceToHost));
                for (int64_t r = 0; r < R; r++)
                    if (base_h[(size_t)r] > 0)
                        HIPCHECK(e, hipMemcpy((int64_t *)nb.p + r * new_cap * 6, (int64_t *)e->t_mev.p + r * mev_cap * 6,
                                              (size_t)std::min<int64_t>((int64_t)base_h[(size_t)r], mev_cap) * 48, hipMemcpyDeviceToDevice));
                // the new buffer takes the old one's place in the engine's bookkeeping
                HIPCHECK(e, hipFree(e->t_mev.p));
                e->dev_bytes -= e->t_mev.bytes;
                e->all.erase(std::remove(e->all.begin(), e->all.end(), &nb), e->all.end());
                e->t_mev.p = nb.p; e->t_mev.bytes = nb.bytes;
                mev_cap = new_cap;
                a.mev = (int64_t *)e->t_mev.p; a.mev_cap = mev_cap;
                e->tau_mev_cap = mev_cap;
            }
            if (again & 2) dense_once = true;
            if (again) HIPCHECK(e, hipMemset(a.grow, 0, (size_t)R * 4));
            return VGX_OK;
        };
        bool finished_on_device = false;
        for (int tries = 0;; tries++) {
            if (spec_rounds && sparse_default && !dense_once) {
                // One replicate, ONE synchronisation per round: the front passes of `spec_k` tries back to back (each returns at once when
                // an earlier one has found nothing: VgxTauArgs.spec), the try proper of that one, and the end of the step, all enqueued
                // without a look from the host.  What the host reads afterwards: accepted / grow as the last decide kernel that ran left
                // them.  (Tries that find a failure cost what they cost before; what goes is the host's turn between them.)
                a.sparse = 1;
                HIPCHECK(e, hipMemsetAsync(a.spec, 0, (size_t)R * 4, e->stream));
                auto next_gen = [&]() -> int {
                    if (++a.gen >= (1u << 25)) {   // the table's try counter wraps: start over with an empty table
                        if (hipMemsetAsync(e->t_stkey.p, 0, (size_t)(R * st_size) * 8, e->stream) != hipSuccess) return VGX_ERR_HIP;
                        a.gen = 1;
                    }
                    return VGX_OK;
                };
                for (int j = 0; j < spec_k; j++) {
                    if (next_gen()) return fail(e, VGX_ERR_HIP, "vgx_simulate_tau: hipMemsetAsync failed");
                    a.phase = 1; a.gate = 1;
                    HIPCHECK(e, vgxi_tau_draw(&a, e->stream));
                    HIPCHECK(e, vgxi_tau_decide(&a, e->stream));
                    launches += 3;
                }
                if (next_gen()) return fail(e, VGX_ERR_HIP, "vgx_simulate_tau: hipMemsetAsync failed");
                a.phase = 2; a.gate = 2;
                HIPCHECK(e, vgxi_tau_draw(&a, e->stream));
                HIPCHECK(e, vgxi_tau_draw_big(&a, e->stream));
                HIPCHECK(e, vgxi_tau_arrivals(&a, e->stream));
                HIPCHECK(e, vgxi_tau_verdict(&a, e->stream));
                HIPCHECK(e, vgxi_tau_decide(&a, e->stream));
                HIPCHECK(e, vgxi_tau_apply(&a, e->stream));
                if (a.use8) { HIPCHECK(e, vgxi_tau_sync8(&a, e->stream)); launches += 1; }
                a.gate = 3;
                HIPCHECK(e, vgxi_tau_finish(&a, e->stream));
                a.gate = 0; a.phase = 0;
                launches += 8;
                HIPCHECK(e, hipEventRecord(e->ev1, e->stream));
                HIPCHECK(e, hipStreamSynchronize(e->stream));
                host_syncs += 1;
                tries += spec_k;
                acc_h.assign(pin_flags, pin_flags + (size_t)R * 2);
                if (acc_h[0]) { finished_on_device = true; break; }
                const int again = acc_h[(size_t)R];
                if (again) {   // (the rare cases: a list to enlarge, or the dense delta arrays — then the loop below runs this try)
                    const int rca = handle_again(again);
                    if (rca) return rca;
                }
                if (tries > 600) return fail(e, VGX_ERR_LOOP_GUARD, "vgx_simulate_tau: tau halving did not converge");
                continue;
            }
            a.sparse = (sparse_default && !dense_once) ? 1 : 0;
            if (!a.sparse && !dense_ready) {
                int rcd = ensure_dense();
                if (rcd) return rcd;
                a.dChk = (int32_t *)e->t_dChk.p; a.dApp = (int32_t *)e->t_dApp.p;
            }
            dense_once = false;
            if (++a.gen >= (1u << 25)) {   // the table's try counter wraps: start over with an empty table
                HIPCHECK(e, hipMemsetAsync(e->t_stkey.p, 0, (size_t)(R * st_size) * 8, e->stream));
                a.gen = 1;
            }
            if (front_split && a.sparse && !front_done) {
                // One replicate: the front pass of the try first, alone (most tries end there: three kernels and the host's turn instead
                // of ten); if it finds nothing the try proper follows (phase 2), with the queue the list pass has already built.
                a.phase = 1;
                tries_total += 1;
                if (a.use_list) tries_lists += 1;
                HIPCHECK(e, vgxi_tau_draw(&a, e->stream));
                HIPCHECK(e, vgxi_tau_decide(&a, e->stream));
                launches += 3;
                HIPCHECK(e, hipStreamSynchronize(e->stream));
                host_syncs += 1;
                a.phase = 0;
                if (pin_flags[2 * R] == 1) { front_done = true; continue; }     // nothing found: the same try, for real
                if (pin_flags[0]) break;                                        // (the loop guard of the halving: handled below like an accepted step)
                if (tries > 600) return fail(e, VGX_ERR_LOOP_GUARD, "vgx_simulate_tau: tau halving did not converge");
                continue;                                                       // rejected: tau halved, the next try
            }
            a.phase = front_done ? 2 : 0;
            front_done = false;
            if (a.phase == 0) {
                tries_total += 1;
                if (a.use_list && a.front_on && a.sparse) tries_lists += 1;
            }
            HIPCHECK(e, vgxi_tau_draw(&a, e->stream));
            HIPCHECK(e, vgxi_tau_draw_big(&a, e->stream));   // (+ the immunity transitions: extra blocks of the same launch)
            if (a.sparse) {
                HIPCHECK(e, vgxi_tau_arrivals(&a, e->stream));
                HIPCHECK(e, vgxi_tau_verdict(&a, e->stream));
                HIPCHECK(e, vgxi_tau_decide(&a, e->stream));
                HIPCHECK(e, vgxi_tau_apply(&a, e->stream));
                if (a.use8) { HIPCHECK(e, vgxi_tau_sync8(&a, e->stream)); launches += 1; }
            } else {
                i8_dirty = true;     // (the dense commit pass changes the counts without the one-byte copy)
                HIPCHECK(e, vgxi_tau_scatter(&a, e->stream));
                HIPCHECK(e, vgxi_tau_suspect(&a, e->stream));
                if (a.dense_check) HIPCHECK(e, vgxi_tau_check(&a, e->stream));
                else if (suspect_cap < P * H) {
                    // more compartments below zero on their own than the list holds (never at tries the sieve lets through; tiny
                    // models list every compartment): the dense pass decides
                    susp_h.resize((size_t)R);
                    HIPCHECK(e, hipMemcpyAsync(susp_h.data(), a.suspect_n, (size_t)R * 8, hipMemcpyDeviceToHost, e->stream));
                    HIPCHECK(e, hipStreamSynchronize(e->stream));
                    bool over = false;
                    for (int64_t r = 0; r < R; r++) over = over || (int64_t)susp_h[(size_t)r] > suspect_cap;
                    if (over) { HIPCHECK(e, vgxi_tau_check(&a, e->stream)); launches += 1; }
                }
                HIPCHECK(e, vgxi_tau_decide(&a, e->stream));
                HIPCHECK(e, vgxi_tau_commit(&a, e->stream));
            }
            launches += 8;
            HIPCHECK(e, hipStreamSynchronize(e->stream));
            host_syncs += 1;
            acc_h.assign(pin_flags, pin_flags + (size_t)R * 2);   // accepted[R], grow[R]: the decide kernel's copy in pinned host memory
            bool all = true;
            for (int64_t r = 0; r < R; r++)
                if (running[(size_t)r] && !acc_h[(size_t)r]) all = false;
            if (all) break;
            // a try that lost data (a full list) or that the sparse check could not decide was discarded by the decide kernel
            // without touching tau or the try index: enlarge the list (it is empty now) / switch to the dense delta arrays
            // and run the same try again
            int again = 0;
            for (int64_t r = 0; r < R; r++) again |= acc_h[(size_t)(R + r)];
            {
                const int rca = handle_again(again);
                if (rca) return rca;
            }
            if (tries > 600) return fail(e, VGX_ERR_LOOP_GUARD, "vgx_simulate_tau: tau halving did not converge");
        }
        if (!finished_on_device) {
            HIPCHECK(e, vgxi_tau_finish(&a, e->stream));
            launches += 1;
            HIPCHECK(e, hipEventRecord(e->ev1, e->stream));
            HIPCHECK(e, hipStreamSynchronize(e->stream));
            host_syncs += 1;
        }
        float ms = 0.f;
        HIPCHECK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        ms_total += ms;
        tau_h.resize((size_t)R);
        err_h.resize((size_t)R);
        std::vector<int64_t> gI_d((size_t)R), res_h((size_t)R * 16);
        int64_t occ_step = -1;
        std::vector<unsigned long long> mevb((size_t)R);
        memcpy(res_h.data(), pin_res, (size_t)R * 16 * 8);   // packed by the finish kernel, its copy in pinned host memory
        for (int64_t r = 0; r < R; r++) {
            if (!running[(size_t)r]) continue;
            const int64_t *o = &res_h[(size_t)r * 16];
            memcpy(&tau_h[(size_t)r], &o[0], 8);
            gI_d[(size_t)r] = o[1];
            for (int i = 0; i < 8; i++) cnt[(size_t)r * 8 + i] = o[2 + i];
            mevb[(size_t)r] = (unsigned long long)o[10];
            mevn[(size_t)r] = (unsigned long long)o[11];
            err_h[(size_t)r] = (int32_t)o[12];
            if (o[13] >= 0) occ_step = std::max<int64_t>(occ_step, o[13]);
            if (a.drift_sparse && o[15] >= 0) {
                // (another look after 32 steps, then 64, ... 4096 while the answer stays the same: such a pass can be many times the dense one)
                if ((o[13] + 30 * o[15]) * 26 > P * H) { sparse_ban = sparse_ban_len; sparse_ban_len = std::min(2 * sparse_ban_len, 4096); }
                else sparse_ban_len = 32;
            }
            if (spec_rounds && r == 0) {
                tries_total += o[14] + 1;
                if (a.use_list) tries_lists += o[14] + 1;
                if (spec_adapt) spec_k = (int)std::min<int64_t>(std::max<int64_t>(o[14] + 2, 2), 8);
            }
        }
        if (occ_step >= 0) occ_est = occ_step;
        for (int64_t r = 0; r < R; r++) {
            if (!running[(size_t)r]) continue;
            if (err_h[(size_t)r] == VGX_ERR_CAPACITY) return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: replicate " + std::to_string(r) + ": list of cross-compartment events full");
            if (err_h[(size_t)r] == 7) return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: replicate " + std::to_string(r) + ": lockdown log full (" + std::to_string(VGX_LOC_CAP) + " switches per call)");
            if (err_h[(size_t)r]) return fail(e, VGX_ERR_LOOP_GUARD, "vgx_simulate_tau: replicate " + std::to_string(r) + ": tau underflow in the halving loop");
            if (mev_cap > 0 && (int64_t)mevn[(size_t)r] > mev_cap)
                return fail(e, VGX_ERR_CAPACITY, "vgx_simulate_tau: multievent buffer full (" + std::to_string(mevn[(size_t)r]) + " rows after " +
                                                     std::to_string(steps_done[(size_t)r] + 1) + " steps, room for " + std::to_string(mev_cap) +
                                                     "; pass record_events=0 for large runs)");
            tnow[(size_t)r] += tau_h[(size_t)r];                       // pyx:2322
            e->tau_log[(size_t)r].push_back({tnow[(size_t)r], (int64_t)mevb[(size_t)r], (int64_t)mevn[(size_t)r], (int32_t)res_h[(size_t)r * 16 + 14]});  // pyx:2325
            ev_ptr[(size_t)r] += 1;
            step_h[(size_t)r] += 1;
            steps_done[(size_t)r] += 1;
            gI[(size_t)r] = gI_d[(size_t)r];
            dev_time[(size_t)r] = tnow[(size_t)r];   // the finish kernel made the same two updates on the device
            dev_step[(size_t)r] = step_h[(size_t)r];
        }
    }

    lap("step loop");
    if (timing) {
        int64_t st_all = 0;
        for (int64_t r = 0; r < R; r++) st_all += steps_done[(size_t)r];
        fprintf(stderr, "vgx_simulate_tau: %lld tries, %lld of them over the lists of occupied compartments; %lld host synchronisations in the step loop "
                        "(%.2f per step)\n", (long long)tries_total, (long long)tries_lists, (long long)host_syncs,
                (double)host_syncs / (double)std::max<int64_t>(st_all, 1));
    }
    // ---- results ----
    std::vector<unsigned long long> locn((size_t)R);
    HIPCHECK(e, hipMemcpy(locn.data(), a.loc_n, (size_t)R * 8, hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < R; r++) {
        int64_t n = std::min<int64_t>((int64_t)locn[(size_t)r], VGX_LOC_CAP);
        if (n <= 0) continue;
        std::vector<int32_t> rec((size_t)n * 2);
        std::vector<double> tt((size_t)n);
        HIPCHECK(e, hipMemcpy(rec.data(), (int32_t *)e->r_locrec.p + r * VGX_LOC_CAP * 2, (size_t)n * 8, hipMemcpyDeviceToHost));
        HIPCHECK(e, hipMemcpy(tt.data(), (double *)e->r_loctime.p + r * VGX_LOC_CAP, (size_t)n * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            e->tau_loc_state[(size_t)r].push_back(rec[(size_t)(i * 2)]);
            e->tau_loc_pop[(size_t)r].push_back(rec[(size_t)(i * 2 + 1)]);
            e->tau_loc_time[(size_t)r].push_back(tt[(size_t)i]);
        }
    }
    e->tau_sc.assign((size_t)R, VgxRepScalars{});
    for (int64_t r = 0; r < R; r++) {
        VgxRepScalars &s = e->tau_sc[(size_t)r];
        const std::vector<int64_t> &c0 = cnt0[(size_t)r];
        const int64_t *c = &cnt[(size_t)r * 8];
        s.currentTime = tnow[(size_t)r]; s.totalRate = h.totalRate; s.totalMig = h.totalMigrationRate;
        s.tau_l = tau_h.empty() ? h.tau_l : tau_h[(size_t)r];
        s.globalInfectious = gI[(size_t)r];
        s.bCounter = c0[0] + c[0]; s.dCounter = c0[1] + c[1]; s.sCounter = c0[2] + c[2]; s.mCounter = c0[3] + c[3];
        s.iCounter = c0[4] + c[4]; s.migPlus = c0[5] + c[5]; s.migNonPlus = h.migNonPlus;
        s.swapLockdown = c0[6] + c[6];
        s.good_attempt = good[(size_t)r];
        s.ev_ptr = ev_ptr[(size_t)r];
        s.loop_iterations = steps_done[(size_t)r];
        s.restarts = restarts[(size_t)r];
        s.loc_n = (int64_t)e->tau_loc_time[(size_t)r].size();
        s.mev_rows = (int64_t)mevn[(size_t)r];
        s.traj_next = c[7];  // events drawn (sum of multiplicities), reported through vgx_counters.reserved[0]
        if (restarts[(size_t)r] > 0) s.migNonPlus = 0;
    }
    e->tau_sieve_skipped.assign((size_t)R, 0);
    HIPCHECK(e, hipMemcpy(e->tau_sieve_skipped.data(), a.sieve_skipped, (size_t)R * 8, hipMemcpyDeviceToHost));
    e->sc_host = e->tau_sc;
    e->sc_host_valid = true;
    e->direct_logs_valid = false;
    e->last_was_tau = true;
    e->last_ms = ms_total;
    e->last_launches = launches;
    e->last_ev_size = ev_size;
    e->ev_ptr0 = ev_ptr_start;
    h.ev_ptr = ev_ptr[0];
    return VGX_OK;
}

// ------------------------------------------------------------------------------------------------
// The host clock.  The reference advances currentTime with the C library's log() (SampleTime, pyx:476-478:
// currentTime += -log(u) / (totalRate + totalMigrationRate)), which is neither correctly rounded nor the same on every
// CPU; the device cannot reproduce it, and time never feeds back into the dynamics.  So the kernels log, per recorded
// event, the denominator of its time step and the index of its loop iteration, and the times are accumulated HERE, with
// this host's libm, from the same PCG64 stream (uniform 2j of the attempt's stream is iteration j's time draw,
// pyx:477,488): event times, the final currentTime and the lockdown timestamps come out as the reference computes them,
// bit for bit on a host whose libm matches.  Rejected migrations (pyx:691-692) leave the state and therefore the
// denominator unchanged, so every iteration between two records uses the later record's denominator.
struct ClockRun {      // accumulates one attempt's clock
    VgxPcg64 g;
    bool philox = false;   // the call drew from the counter-based stream: iteration i took outputs 2 i (time) and 2 i + 1 (event)
    uint64_t seed_ = 0;
    uint32_t att_ = 0;
    int64_t iter = 0;
    double t = 0.0;
    void open(int64_t seed, int64_t attempt, double t0) {
        vgx_pcg64_seed(g, (uint64_t)seed, (uint32_t)attempt);
        seed_ = (uint64_t)seed; att_ = (uint32_t)attempt;
        iter = 0; t = t0;
    }
    void advance(int64_t to_iter, double rate) {
        while (iter < to_iter) {
            double u;
            if (philox) {
                u = vgx_philox_stream_double(seed_, att_, 2 * (uint64_t)iter);
            } else {
                u = vgx_pcg64_double(g);
                (void)vgx_pcg64_next(g);             // the event's own uniform (pyx:488)
            }
            t += -std::log(u) / rate;
            iter++;
        }
    }
};

static int host_clock(vgx_engine *e, int64_t rep) {
    vgx_engine::HostClock &hc = e->hc;
    if (hc.rep == rep) return VGX_OK;
    const VgxRepScalars &s = e->sc_host[(size_t)rep];
    hc.rep = -1;
    hc.times.clear();
    hc.loc_times.clear();
    hc.limit_mismatch = false;
    const bool rewound = s.restarts > 0;
    hc.e0 = rewound ? 0 : ((size_t)rep < e->call_ev0.size() ? e->call_ev0[(size_t)rep] : e->ev_ptr0);
    const int64_t n = std::max<int64_t>(s.ev_ptr - hc.e0, 0);
    const int64_t nloc = std::min<int64_t>(s.loc_n, e->loc_cap);
    std::vector<double> loc_dev((size_t)nloc);
    std::vector<int64_t> loc_key((size_t)nloc);
    if (nloc > 0) {
        HIPCHECK(e, hipMemcpy(loc_dev.data(), (double *)e->r_loctime.p + rep * e->loc_cap, (size_t)nloc * 8, hipMemcpyDeviceToHost));
        HIPCHECK(e, hipMemcpy(loc_key.data(), (int64_t *)e->r_lociter.p + rep * e->loc_cap, (size_t)nloc * 8, hipMemcpyDeviceToHost));
    }
    hc.loc_times = loc_dev;
    hc.final_time = s.currentTime;
    hc.exact = e->call_recorded;
    if (!e->call_recorded) {       // no rate log: the device clock (vgx_log) is all there is
        hc.rep = rep;
        return VGX_OK;
    }
    const double t_call = e->call_t0[(size_t)rep];
    const int64_t seed = e->seeds[(size_t)rep];
    const int64_t IT = ((int64_t)1 << 40) - 1;
    // lockdown records written outside the event loop (PrepareParameters / Restart): the attempt's start time
    for (int64_t i = 0; i < nloc; i++)
        if ((loc_key[(size_t)i] & IT) == 0) hc.loc_times[(size_t)i] = (loc_key[(size_t)i] >> 40) == 0 ? t_call : 0.0;
    // failed attempts that switched a lockdown inside their loop: their own (rate, iteration) pairs
    const int64_t nfa = std::min<int64_t>(s.fa_n, e->fa_cap);
    if (nfa > 0) {
        std::vector<double> fr((size_t)nfa);
        std::vector<int64_t> fk((size_t)nfa);
        HIPCHECK(e, hipMemcpy(fr.data(), (double *)e->r_farate.p + rep * e->fa_cap, (size_t)nfa * 8, hipMemcpyDeviceToHost));
        HIPCHECK(e, hipMemcpy(fk.data(), (int64_t *)e->r_fakey.p + rep * e->fa_cap, (size_t)nfa * 8, hipMemcpyDeviceToHost));
        int64_t k = 0;
        while (k < nfa) {
            const int64_t att = fk[(size_t)k] >> 40;
            ClockRun c;
            c.philox = e->call_philox;
            c.open(seed, att, att == 0 ? t_call : 0.0);
            for (; k < nfa && (fk[(size_t)k] >> 40) == att; k++) {
                c.advance(fk[(size_t)k] & IT, fr[(size_t)k]);
                for (int64_t i = 0; i < nloc; i++)
                    if (loc_key[(size_t)i] == fk[(size_t)k]) hc.loc_times[(size_t)i] = c.t;
            }
        }
    }
    // the attempt whose events are in the log
    if (s.last_attempt < 0) {                 // no attempt drew a number
        hc.final_time = t_call;
    } else if (s.restarts > s.last_attempt) { // the last attempt failed too: Restart left currentTime = 0 (pyx:717)
        hc.final_time = 0.0;
    } else {
        const int64_t slot0 = hc.e0 - e->ev_base;
        if (slot0 < 0 || slot0 + n > e->evcap) return fail(e, VGX_ERR_ARG, "host_clock: event range outside the device log");
        std::vector<double> rate((size_t)n);
        std::vector<int32_t> cols((size_t)n * VGX_EV_COLS);
        if (n > 0) {
            HIPCHECK(e, hipMemcpy(rate.data(), (double *)e->r_evrate.p + rep * e->evcap + slot0, (size_t)n * 8, hipMemcpyDeviceToHost));
            HIPCHECK(e, hipMemcpy(cols.data(), (int32_t *)e->r_evcols.p + (rep * e->evcap + slot0) * VGX_EV_COLS,
                                  (size_t)n * VGX_EV_COLS * 4, hipMemcpyDeviceToHost));
        }
        ClockRun c;
        c.philox = e->call_philox;
        c.open(seed, s.last_attempt, rewound ? 0.0 : t_call);
        hc.times.resize((size_t)n);
        int64_t li = 0;
        while (li < nloc && ((loc_key[(size_t)li] >> 40) != s.last_attempt || (loc_key[(size_t)li] & IT) == 0)) li++;
        bool limit_ok = true;
        int64_t it = 0;
        for (int64_t k = 0; k < n; k++) {
            // iteration index: 32 bits logged, strictly increasing
            const uint32_t lo = (uint32_t)cols[(size_t)(k * VGX_EV_COLS + 5)];
            it += (int64_t)(uint32_t)(lo - (uint32_t)it);
            if (e->call_has_tlimit && it > c.iter + 1) {   // loop condition of the iterations without a record (pyx:407)
                ClockRun probe = c;
                probe.advance(it - 1, rate[(size_t)k]);
                if (!(probe.t < e->call_tlimit)) limit_ok = false;
            }
            c.advance(it, rate[(size_t)k]);
            hc.times[(size_t)k] = c.t;
            if (e->call_has_tlimit && k + 1 < n && !(c.t < e->call_tlimit)) limit_ok = false;
            while (li < nloc && (loc_key[(size_t)li] >> 40) == s.last_attempt && (loc_key[(size_t)li] & IT) == it) {
                hc.loc_times[(size_t)li] = c.t;
                li++;
            }
        }
        c.advance(s.last_attempt_loops, s.totalRate + s.totalMig);   // trailing iterations without a record
        hc.final_time = c.t;
        // The kernel took its `currentTime < time` decisions (pyx:407) on the device clock (vgx_log, < 1 ulp from the host's
        // log).  If an event time lands within rounding of the limit the two clocks can disagree on one stop decision; the
        // run is then the device clock's run (a getter must not fail after the work is done): times stay the host clock's,
        // the mismatch is kept for vgx_clock_mismatches().
        if (e->call_has_tlimit && (!limit_ok || ((s.currentTime < e->call_tlimit) != (hc.final_time < e->call_tlimit)))) {
            hc.limit_mismatch = true;
            e->clock_mismatches += 1;
        }
    }
    hc.rep = rep;
    return VGX_OK;
}

extern "C" int vgx_get_counters(vgx_engine *e, int64_t replicate, vgx_counters *out) {
    if (!e || !out || replicate < 0 || replicate >= e->R) return VGX_ERR_ARG;
    if (!e->sc_host_valid) return fail(e, VGX_ERR_ARG, "vgx_get_counters: no simulate call yet");
    const VgxRepScalars &s = e->sc_host[(size_t)replicate];
    memset(out, 0, sizeof(*out));
    out->ev_ptr = s.ev_ptr;
    out->ev_first_new = s.restarts > 0 ? 0 : ((!e->last_was_tau && (size_t)replicate < e->call_ev0.size()) ? e->call_ev0[(size_t)replicate] : e->ev_ptr0);
    if (e->last_was_tau) out->reserved[0] = s.traj_next;  // tau: events drawn (sum of channel multiplicities)
    if (e->last_was_tau && (size_t)replicate < e->tau_sieve_skipped.size()) out->reserved[3] = e->tau_sieve_skipped[(size_t)replicate];
    out->loop_iterations = s.loop_iterations;
    out->restarts = s.restarts;
    out->lockdown_records = s.loc_n;
    out->error = s.error;
    out->multievent_rows = s.mev_rows;
    if (!e->last_was_tau) { out->reserved[1] = s.last_attempt; out->reserved[2] = s.last_attempt_loops; }
    else out->reserved[1] = -1;
    return VGX_OK;
}

extern "C" int vgx_get_counters_all(vgx_engine *e, int64_t *out /* [R][4]: ev_ptr, loop_iterations, restarts, tau events drawn */) {
    if (!e || !out) return VGX_ERR_ARG;
    if (!e->sc_host_valid) return fail(e, VGX_ERR_ARG, "vgx_get_counters_all: no simulate call yet");
    for (int64_t r = 0; r < e->R; r++) {
        const VgxRepScalars &s = e->sc_host[(size_t)r];
        out[r * 4 + 0] = s.ev_ptr;
        out[r * 4 + 1] = s.loop_iterations;
        out[r * 4 + 2] = s.restarts;
        out[r * 4 + 3] = e->last_was_tau ? s.traj_next : 0;
    }
    return VGX_OK;
}

extern "C" int vgx_get_events(vgx_engine *e, int64_t replicate, int64_t first, int64_t count, double *times,
                              int64_t *types, int64_t *haplotypes, int64_t *populations, int64_t *newHaplotypes,
                              int64_t *newPopulations) {
    if (!e || replicate < 0 || replicate >= e->R || first < 0 || count < 0) return VGX_ERR_ARG;
    if (count == 0) return VGX_OK;
    HIPCHECK(e, hipSetDevice(e->device));
    if (e->last_was_tau) {  // MULTITYPE records (pyx:2325): [start, end) of the step's multievent rows
        const auto &lg = e->tau_log[(size_t)replicate];
        int64_t i0 = first - e->tau_ev_ptr0[(size_t)replicate];
        if (i0 < 0 || i0 + count > (int64_t)lg.size()) return fail(e, VGX_ERR_ARG, "vgx_get_events: range outside the last tau call");
        for (int64_t i = 0; i < count; i++) {
            const auto &st = lg[(size_t)(i0 + i)];
            if (times) times[i] = st.time;
            if (types) types[i] = VGX_MULTITYPE;
            if (haplotypes) haplotypes[i] = st.m0;
            if (populations) populations[i] = st.m1;
            if (newHaplotypes) newHaplotypes[i] = 0;
            if (newPopulations) newPopulations[i] = 0;
        }
        return VGX_OK;
    }
    int64_t slot0 = first - e->ev_base;
    if (slot0 < 0 || slot0 + count > e->evcap) return fail(e, VGX_ERR_ARG, "vgx_get_events: range outside the device log of the last call");
    if (!e->call_recorded) return fail(e, VGX_ERR_ARG, "vgx_get_events: the last call did not record events");
    std::vector<int32_t> cols((size_t)count * VGX_EV_COLS);
    HIPCHECK(e, hipMemcpy(cols.data(), (int32_t *)e->r_evcols.p + (replicate * e->evcap + slot0) * VGX_EV_COLS,
                          (size_t)count * VGX_EV_COLS * 4, hipMemcpyDeviceToHost));
    if (times) {   // accumulated on the host with libm, as the reference does (host_clock)
        int rc = host_clock(e, replicate);
        if (rc) return rc;
        const int64_t i0 = first - e->hc.e0;
        if (i0 < 0 || i0 + count > (int64_t)e->hc.times.size())
            return fail(e, VGX_ERR_ARG, "vgx_get_events: times exist for the events of the last call only");
        memcpy(times, e->hc.times.data() + i0, (size_t)count * 8);
    }
    int64_t *dst[5] = {types, haplotypes, populations, newHaplotypes, newPopulations};
    for (int c = 0; c < 5; c++)
        if (dst[c])
            for (int64_t i = 0; i < count; i++) dst[c][i] = cols[(size_t)(i * VGX_EV_COLS + c)];
    return VGX_OK;
}

extern "C" int vgx_get_tau_tries(vgx_engine *e, int64_t replicate, int64_t first, int64_t count, int32_t *out) {
    if (!e || !out || replicate < 0 || replicate >= e->R || first < 0 || count < 0) return VGX_ERR_ARG;
    if (!e->last_was_tau || (size_t)replicate >= e->tau_log.size()) return fail(e, VGX_ERR_ARG, "vgx_get_tau_tries: the last call was not vgx_simulate_tau");
    const auto &lg = e->tau_log[(size_t)replicate];
    if (first + count > (int64_t)lg.size()) return fail(e, VGX_ERR_ARG, "vgx_get_tau_tries: the last call made fewer steps");
    for (int64_t i = 0; i < count; i++) out[i] = lg[(size_t)(first + i)].tries;
    return VGX_OK;
}

extern "C" int vgx_get_lockdowns(vgx_engine *e, int64_t replicate, int64_t cap, int64_t *states, int64_t *populations,
                                 double *times, int64_t *n) {
    if (!e || !n || replicate < 0 || replicate >= e->R) return VGX_ERR_ARG;
    if (!e->sc_host_valid) return fail(e, VGX_ERR_ARG, "vgx_get_lockdowns: no simulate call yet");
    HIPCHECK(e, hipSetDevice(e->device));
    if (e->last_was_tau) {
        const auto &tt = e->tau_loc_time[(size_t)replicate];
        *n = (int64_t)tt.size();
        for (int64_t i = 0; i < std::min<int64_t>(*n, cap); i++) {
            if (states) states[i] = e->tau_loc_state[(size_t)replicate][(size_t)i];
            if (populations) populations[i] = e->tau_loc_pop[(size_t)replicate][(size_t)i];
            if (times) times[i] = tt[(size_t)i];
        }
        return VGX_OK;
    }
    int64_t cnt = std::min<int64_t>(e->sc_host[(size_t)replicate].loc_n, e->loc_cap);
    *n = cnt;
    cnt = std::min(cnt, cap);
    if (cnt <= 0) return VGX_OK;
    std::vector<int32_t> rec((size_t)cnt * 2);
    HIPCHECK(e, hipMemcpy(rec.data(), (int32_t *)e->r_locrec.p + replicate * e->loc_cap * 2, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    int rch = host_clock(e, replicate);
    if (rch) return rch;
    const std::vector<double> &tt = e->hc.loc_times;
    for (int64_t i = 0; i < cnt; i++) {
        if (states) states[i] = rec[(size_t)(i * 2)];
        if (populations) populations[i] = rec[(size_t)(i * 2 + 1)];
        if (times) times[i] = tt[(size_t)i];
    }
    return VGX_OK;
}

extern "C" int vgx_get_multievents(vgx_engine *e, int64_t replicate, int64_t cap, int64_t *num, double *times,
                                   int64_t *types, int64_t *haplotypes, int64_t *populations, int64_t *newHaplotypes,
                                   int64_t *newPopulations, int64_t *n) {
    if (!e || !n || replicate < 0 || replicate >= e->R) return VGX_ERR_ARG;
    *n = 0;
    if (!e->last_was_tau) return VGX_OK;
    HIPCHECK(e, hipSetDevice(e->device));
    const auto &lg = e->tau_log[(size_t)replicate];
    int64_t rows = (lg.empty() || e->tau_mev_cap <= 0) ? 0 : lg.back().m1;   // no rows when the call did not record them
    *n = rows;
    rows = std::min(rows, cap);
    if (rows <= 0) return VGX_OK;
    const int64_t mev_cap = e->tau_mev_cap;
    if (mev_cap <= 0) return VGX_OK;  // the call did not record multievents
    std::vector<int64_t> buf((size_t)rows * 6);
    HIPCHECK(e, hipMemcpy(buf.data(), (int64_t *)e->t_mev.p + replicate * mev_cap * 6, (size_t)rows * 48, hipMemcpyDeviceToHost));
    size_t st = 0;
    for (int64_t i = 0; i < rows; i++) {
        while (st + 1 < lg.size() && i >= lg[st].m1) st++;
        if (num) num[i] = buf[(size_t)(i * 6)];
        if (times) times[i] = lg[st].time;
        if (types) types[i] = buf[(size_t)(i * 6 + 1)];
        if (haplotypes) haplotypes[i] = buf[(size_t)(i * 6 + 2)];
        if (populations) populations[i] = buf[(size_t)(i * 6 + 3)];
        if (newHaplotypes) newHaplotypes[i] = buf[(size_t)(i * 6 + 4)];
        if (newPopulations) newPopulations[i] = buf[(size_t)(i * 6 + 5)];
    }
    return VGX_OK;
}

extern "C" int vgx_get_trajectories(vgx_engine *e, double *out, int out_is_device) {
    if (!e || !out) return VGX_ERR_ARG;
    if (e->traj_points <= 0) return fail(e, VGX_ERR_ARG, "vgx_get_trajectories: the last call recorded none");
    HIPCHECK(e, hipSetDevice(e->device));
    size_t bytes = (size_t)(e->R * e->traj_points * e->d.popNum * 2) * 8;
    HIPCHECK(e, hipMemcpy(out, e->r_traj.p, bytes, out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return VGX_OK;
}

extern "C" hipError_t vgxi_launch_traj_i32(const double *src, int32_t *dst, int64_t n, hipStream_t stream);
extern "C" int vgx_get_trajectories_int(vgx_engine *e, int32_t *out_device) {
    if (!e || !out_device) return VGX_ERR_ARG;
    if (e->traj_points <= 0) return fail(e, VGX_ERR_ARG, "vgx_get_trajectories_int: the last call recorded none");
    for (int64_t pn = 0; pn < e->d.popNum; pn++)
        if (e->sizes[(size_t)pn] >= ((int64_t)1 << 31)) return fail(e, VGX_ERR_ARG, "vgx_get_trajectories_int: population sizes of 2^31 or more");
    HIPCHECK(e, hipSetDevice(e->device));
    HIPCHECK(e, vgxi_launch_traj_i32((const double *)e->r_traj.p, out_device, e->R * e->traj_points * e->d.popNum * 2, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return VGX_OK;
}

extern "C" int64_t vgx_clock_mismatches(const vgx_engine *e) { return e ? e->clock_mismatches : 0; }

extern "C" int vgx_get_state(vgx_engine *e, int64_t replicate, vgx_state *out) {
    if (!e || !out || replicate < 0 || replicate >= e->R) return VGX_ERR_ARG;
    if (e->last_was_tau && e->sc_host_valid) {
        HIPCHECK(e, hipSetDevice(e->device));
        const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum;
        const VgxRepScalars &s = e->sc_host[(size_t)replicate];
        HostState &h = e->hs;
        if (out->susceptible) HIPCHECK(e, hipMemcpy(out->susceptible, (int64_t *)e->t_S.p + replicate * P * S, (size_t)(P * S) * 8, hipMemcpyDeviceToHost));
        if (out->infectious) {
            std::unique_ptr<int32_t[]> inf32(new int32_t[(size_t)(P * H)]);   // (not zero-filled: it is overwritten at once)
            HIPCHECK(e, hipMemcpy(inf32.get(), (int32_t *)e->t_I.p + replicate * P * H, (size_t)(P * H) * 4, hipMemcpyDeviceToHost));
            int64_t *dst = out->infectious;
            const int32_t *src = inf32.get();
            for_parts(P * H, [&](int64_t b0, int64_t en, unsigned) { for (int64_t i = b0; i < en; i++) dst[i] = src[i]; });
        }
        if (out->initial_susceptible) memcpy(out->initial_susceptible, h.initial_susceptible.data(), (size_t)(P * S) * 8);
        if (out->initial_infectious) memcpy(out->initial_infectious, h.initial_infectious.data(), (size_t)(P * H) * 8);
        std::vector<int64_t> tot((size_t)P);
        std::vector<double> cd((size_t)P);
        std::vector<int32_t> lk((size_t)P);
        HIPCHECK(e, hipMemcpy(tot.data(), (int64_t *)e->t_totInf.p + replicate * P, (size_t)P * 8, hipMemcpyDeviceToHost));
        HIPCHECK(e, hipMemcpy(cd.data(), (double *)e->t_cd.p + replicate * P, (size_t)P * 8, hipMemcpyDeviceToHost));
        HIPCHECK(e, hipMemcpy(lk.data(), (int32_t *)e->t_lock.p + replicate * P, (size_t)P * 4, hipMemcpyDeviceToHost));
        std::vector<int64_t> sus((size_t)(P * S));
        HIPCHECK(e, hipMemcpy(sus.data(), (int64_t *)e->t_S.p + replicate * P * S, (size_t)(P * S) * 8, hipMemcpyDeviceToHost));
        for (int64_t pn = 0; pn < P; pn++) {
            if (out->totalInfectious) out->totalInfectious[pn] = tot[(size_t)pn];
            if (out->totalSusceptible) {
                int64_t t = 0;
                for (int64_t sn = 0; sn < S; sn++) t += sus[(size_t)(pn * S + sn)];
                out->totalSusceptible[pn] = t;
            }
            if (out->lockdownON) out->lockdownON[pn] = lk[(size_t)pn];
            if (out->contactDensity) out->contactDensity[pn] = cd[(size_t)pn];
        }
        out->first_simulation = h.first_simulation;
        out->globalInfectious = s.globalInfectious;
        out->bCounter = s.bCounter; out->dCounter = s.dCounter; out->sCounter = s.sCounter; out->mCounter = s.mCounter;
        out->iCounter = s.iCounter; out->swapLockdown = s.swapLockdown; out->migPlus = s.migPlus;
        out->migNonPlus = s.migNonPlus; out->good_attempt = s.good_attempt;
        out->currentTime = s.currentTime; out->totalRate = s.totalRate; out->totalMigrationRate = s.totalMig;
        out->tau_l = s.tau_l;
        out->ev_ptr = s.ev_ptr; out->ev_size = e->last_ev_size;
        return VGX_OK;
    }
    if (!e->dev_state_valid || !e->sc_host_valid) return fail(e, VGX_ERR_ARG, "vgx_get_state: no device state");
    HIPCHECK(e, hipSetDevice(e->device));
    const int64_t H = e->d.hapNum, P = e->d.popNum, S = e->d.susNum, cap = e->cap;
    const VgxRepScalars &s = e->sc_host[(size_t)replicate];
    std::vector<int64_t> popI((size_t)(PI_COUNT * P));
    std::vector<double> popD((size_t)(PD_COUNT * P));
    HIPCHECK(e, hipMemcpy(popI.data(), (int64_t *)e->r_popI.p + replicate * PI_COUNT * P, popI.size() * 8, hipMemcpyDeviceToHost));
    HIPCHECK(e, hipMemcpy(popD.data(), (double *)e->r_popD.p + replicate * PD_COUNT * P, popD.size() * 8, hipMemcpyDeviceToHost));
    if (out->susceptible)
        HIPCHECK(e, hipMemcpy(out->susceptible, (int64_t *)e->r_sus.p + replicate * P * S, (size_t)(P * S) * 8, hipMemcpyDeviceToHost));
    if (out->infectious) {
        if (!e->counts64_valid) {
            HIPCHECK(e, vgxi_launch_counts64(e->dr.lcnt32, e->dr.lcnt, e->R * P * cap, e->stream));
            HIPCHECK(e, hipStreamSynchronize(e->stream));
            e->counts64_valid = true;
        }
        std::vector<int32_t> nocc((size_t)P);
        HIPCHECK(e, hipMemcpy(nocc.data(), (int32_t *)e->r_nocc.p + replicate * P, (size_t)P * 4, hipMemcpyDeviceToHost));
        memset(out->infectious, 0, (size_t)(P * H) * 8);
        std::vector<int32_t> hap;
        std::vector<int64_t> cnt;
        for (int64_t pn = 0; pn < P; pn++) {
            int64_t n = nocc[(size_t)pn];
            if (n <= 0) continue;
            hap.resize((size_t)n);
            cnt.resize((size_t)n);
            HIPCHECK(e, hipMemcpy(hap.data(), (int32_t *)e->r_lhap.p + (replicate * P + pn) * cap, (size_t)n * 4, hipMemcpyDeviceToHost));
            HIPCHECK(e, hipMemcpy(cnt.data(), (int64_t *)e->r_lcnt.p + (replicate * P + pn) * cap, (size_t)n * 8, hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < n; k++) out->infectious[pn * H + hap[(size_t)k]] = cnt[(size_t)k];
        }
    }
    HostState &h = e->hs;
    if (out->initial_susceptible) memcpy(out->initial_susceptible, h.initial_susceptible.data(), (size_t)(P * S) * 8);
    if (out->initial_infectious) memcpy(out->initial_infectious, h.initial_infectious.data(), (size_t)(P * H) * 8);
    for (int64_t pn = 0; pn < P; pn++) {
        if (out->totalSusceptible) out->totalSusceptible[pn] = popI[(size_t)(PI_TOTSUS * P + pn)];
        if (out->totalInfectious) out->totalInfectious[pn] = popI[(size_t)(PI_TOTINF * P + pn)];
        if (out->lockdownON) out->lockdownON[pn] = popI[(size_t)(PI_LOCK * P + pn)];
        if (out->contactDensity) out->contactDensity[pn] = popD[(size_t)(PD_CD * P + pn)];
    }
    out->first_simulation = h.first_simulation;
    out->globalInfectious = s.globalInfectious;
    out->bCounter = s.bCounter; out->dCounter = s.dCounter; out->sCounter = s.sCounter; out->mCounter = s.mCounter;
    out->iCounter = s.iCounter; out->swapLockdown = s.swapLockdown; out->migPlus = s.migPlus;
    out->migNonPlus = s.migNonPlus; out->good_attempt = s.good_attempt;
    {
        int rch = host_clock(e, replicate);
        if (rch) return rch;
    }
    out->currentTime = e->hc.final_time; out->totalRate = s.totalRate; out->totalMigrationRate = s.totalMig;
    out->tau_l = s.tau_l;
    out->ev_ptr = s.ev_ptr; out->ev_size = e->last_ev_size;
    return VGX_OK;
}

extern "C" int vgx_get_profile(vgx_engine *e, int64_t replicate, int64_t *out16) {
    if (!e || !out16 || replicate < 0 || replicate >= e->R || !e->r_prof.p) return VGX_ERR_ARG;
    HIPCHECK(e, hipSetDevice(e->device));
    HIPCHECK(e, hipMemcpy(out16, (unsigned long long *)e->r_prof.p + replicate * VGX_PROF_SLOTS, VGX_PROF_SLOTS * 8, hipMemcpyDeviceToHost));
    return VGX_OK;
}
