// vgx_dev.h — structures shared by the host side of libvgx (vgx_api.hip) and the gfx950 kernels.
//
// HBM layout.  Parameters are one read-only copy shared by all replicates.  Each
// replicate (= one seeded trajectory) owns:
//   * a population block: per-population rate caches and counters, [field][P], loaded into LDS by the
//     replicate's wavefront for the whole run;
//   * per population an ORDERED OCCUPANCY LIST of the haplotypes present there: struct-of-arrays
//     (hap int32, class int32, count int64), sorted by haplotype index, contiguous -> the per-event
//     propensity pass of population pi is one coalesced stream of nocc[pi]*16 bytes instead of the
//     reference's H*(84+16S)-byte dense row (pyx:516-528), and visits haplotypes in the reference's
//     summation order, so sums and scans stay bit-identical;
//   * the event log (f64 time + 5 x int32 per event) and a small lockdown log.
#pragma once
#include <stdint.h>

#define VGX_WAVE 64
#define VGX_MAX_CLASSES 1024   // distinct per-haplotype rate rows (full classes) supported by the direct kernel
#define VGX_LOC_CAP 4096       // lockdown switches recorded per replicate and call
#define VGX_INC_SHARDS 4096     // shards of the tau kernels' cross-compartment event list
#define VGX_SIEVE_K 16          // halvings the tau sieve looks ahead
#define VGX_EV_COLS 6           // int32 columns of a device log record
#define VGX_FA_CAP 2048         // (rate, iteration) pairs kept per replicate for failed attempts that switched a lockdown
#define VGX_TAU_BIG 64.0         // tau: expected events of a compartment per leap from which every channel is drawn on its own
#define VGX_TAU_BIG_SMALL 16.0   // ... on models with few compartments (the wavefront's slowest lane sets the step time there)
#define VGX_D8_WAVES 8           // wavefronts of a block of vgx_tau_drift8_kernel (= regions of the occupied-compartment lists per tile)
#define VGX_OCC_CAP 1024         // entries of a region of the occupied-compartment lists (a region covers 8192 compartments from eight sites on)
#define VGX_TAU_KINDS 16.0       // ... and from which a lane draws one Poisson number per KIND of event instead of walking through the events
#define VGX_PROF_SLOTS 16       // in-kernel phase stamps of the diagnostic (-DVGX_PROFILE) build

// fields of the per-replicate f64 population block
enum { PD_POPRATE = 0, PD_INFECT, PD_IMMUNE, PD_MIG, PD_MAXEBM, PD_CD, PD_COUNT };
// fields of the per-replicate i64 population block
enum { PI_TOTSUS = 0, PI_TOTINF, PI_LOCK, PI_COUNT };

struct VgxDevParams {
    int32_t H, P, S, sites, C, CB;
    const int32_t *cls;         // [H] full rate class of a haplotype
    const int64_t *suscType;    // [H]
    const double *mRate;        // [H][sites]
    const double *hapMutType;   // [H][sites][3]
    const double *bRate;        // [H]
    const double *susc;         // [H][S]
    const double *c_d, *c_s, *c_tm;  // [C] recovery, sampling, total mutation rate of a class
    const int32_t *c_bidx;      // [C] -> birth class
    const int32_t *c_stype;     // [C] suscType of the class
    const double *cb_b;         // [CB] transmission rate of a birth class
    const double *cb_sigma;     // [CB][S] susceptibility row of a birth class
    const int64_t *sizes;       // [P]
    const double *cdBefore, *cdAfter, *startLD, *endLD, *sampMult, *actualSizes;  // [P]
    const double *mig;          // [P][P] migrationRates with the recomputed diagonal (pyx:290-295)
    const double *suscepTransition;  // [S][S]
    const double *suscepCumul;  // [S]
    double maxEffectiveBirth;   // pyx:340-344
    // recombination branch of Birth (pyx:575-596); recombination == 0: never taken
    double recombination;
    int64_t genome_length;
    const int64_t *sitesPosition;  // [sites]
};

struct VgxRepScalars {
    double currentTime, totalRate, totalMig, tau_l;
    int64_t globalInfectious;
    int64_t bCounter, dCounter, sCounter, mCounter, iCounter, swapLockdown, migPlus, migNonPlus;
    int64_t good_attempt;
    int64_t ev_ptr;          // absolute events.ptr
    int64_t loop_iterations;
    int64_t restarts;
    int64_t loc_n;
    int64_t error;
    int64_t mev_rows;
    int64_t traj_next;       // next trajectory grid point to emit
    int64_t last_attempt;        // index of the last attempt that drew random numbers (-1: none)
    int64_t last_attempt_loops;  // loop iterations of that attempt (2 uniforms each)
    int64_t rec_n;               // recombination records written so far (kept across Restarts like upstream's `rec`)
    int64_t fa_n;                // (rate, iteration) pairs of failed attempts kept for the host clock (may exceed fa_cap)
};

struct VgxDevRep {
    double *popD;        // [R][PD_COUNT][P]
    int64_t *popI;       // [R][PI_COUNT][P]
    int64_t *sus;        // [R][P][S]
    double *immSrc;      // [R][P][S]   immuneSourcePopRate
    double *birthC;      // [R][P][CB]  eventHapPopRate[.,.,0] per birth class, as of the last infect-update of the population
    double *xC;          // [R][P][CB][S] susceptHapPopRate per birth class, same staleness
    double *effMig;      // [R][P][P]
    int32_t *nocc;       // [R][P]
    int32_t *lhap;       // [R][P][cap]
    int32_t *lcls;       // [R][P][cap]
    int64_t *lcnt;       // [R][P][cap]
    int32_t *lcnt32;     // [R][P][cap] the same counts in 4 bytes (population sizes < 2^31), kept by vgx_quad.hip for its streaming passes; or null
    int64_t cap;
    int64_t *ltsum;      // [R][P][capT] sum of the counts of every 64-entry tile of the list (0 beyond the list); behind it, [R][P][capT] doubles:
                         // exact row kernel, long lists: running sum of hapPopRate at the end of every tile as of the last refresh
    int64_t capT;        // cap / 64 + 1
    // initial state for Restart (pyx:714-738), one copy shared by the replicates
    const int32_t *i_nocc;   // [P]
    const int32_t *i_hap;    // [P][i_cap]
    const int32_t *i_cls;
    const int64_t *i_cnt;
    int64_t i_cap;
    const int64_t *i_sus;    // [P][S]
    VgxRepScalars *sc;       // [R]
    const int64_t *seeds;    // [R]
    // event log.  Event TIMES are not logged: the reference accumulates them with the host's libm (pyx:476-478), so the
    // kernel logs what the host needs to do the same bit for bit (vgx_api.hip: host_clock) — the denominator
    // totalRate + totalMigrationRate of the event's time step and the index of its loop iteration (rejected migrations
    // consume an iteration, two uniforms and a time step without a record, pyx:691-692).
    double *ev_rate;         // [R][evcap]
    int32_t *ev_cols;        // [R][evcap][VGX_EV_COLS]  type, haplotype, population, newHaplotype, newPopulation,
                             //                          loop iteration of the attempt (1-based, low 32 bits)
    int64_t evcap;
    int64_t ev_base;         // absolute index of log slot 0
    // lockdown log
    int32_t *loc_rec;        // [R][loc_cap][2]  state, population
    double *loc_time;        // [R][loc_cap]     device clock (used only where the host clock cannot be rebuilt)
    int64_t *loc_iter;       // [R][loc_cap]     (attempt << 40) | loop iteration of the attempt (0: before its first one)
    int64_t loc_cap;
    // failed attempts (<= 100 events, pyx:414) that switched a lockdown: their (rate, iteration) pairs, so that the
    // lockdown records they leave behind (Restart does not clear `loc`, pyx:714-738) get the host clock too
    double *fa_rate;         // [R][fa_cap]
    int64_t *fa_key;         // [R][fa_cap]      (attempt << 40) | loop iteration
    int64_t fa_cap;
    // trajectories
    double *traj;            // [R][T][P][2] or null
    int64_t traj_points;
    double traj_t0, traj_dt;
    // forward recombination records (models.pxi:69-89): idevent, hi, hi2, nhi, posRecomb
    int64_t *rec;            // [R][rec_cap][5] or null
    int64_t rec_cap;
    unsigned long long *prof; // [R][VGX_PROF_SLOTS] cycle sums of the diagnostic build (else untouched)
};

struct VgxDirectArgs {
    VgxDevParams p;
    VgxDevRep r;
    int64_t n_replicates;
    int64_t iterations, sample_size, attempts;
    float time;
    int64_t ev_size;         // absolute events.size
    int64_t max_loop;
    int32_t record_events;
    int32_t lds_bytes;
    int32_t fast;            // 0: reference summation order (bit-exact); 1: order-free sums (vgx_run_opts.mode)
    int32_t rng_philox;      // FAST mode only: 1 = the counter-based Philox stream of vgx_rng.h instead of PCG64 (vgx_run_opts.mode = 2)
    int32_t pad_;
};

// Dense per-replicate state of the lane-per-replicate kernel (vgx_lanes.hip): element i of replicate r at [i * R + r].
struct VgxLaneWs {
    int64_t *inf, *sus, *totS, *totI, *lock;    // [P*H], [P*S], [P], [P], [P]
    double *cd, *birth, *tE, *hpr, *shpr, *immSrc, *infP, *immP, *popR, *migR, *maxEBM, *effMig;
    // [P], [P*H], [P*H], [P*H], [P*H*S], [P*S], [P] x 5, [P*P]
};

// Tau-leaping (vgx_tau.hip): dense compartment arrays per replicate, [R][P][H] / [R][P][S].
struct VgxTauArgs {
    VgxDevParams p;
    int64_t R;
    int32_t *I;          // [R][P][H] infectious (tau population sizes are < 2^31: 4 bytes per compartment)
    uint8_t *I8;         // [R][P][H] min(infectious, 255): streamed by the scan kernel of a step's tries.  Rewritten by every step's
                         // drift pass, or (use8) kept in step with I by vgx_tau_sync8_kernel / vgx_tau_conv8_kernel and READ by the drift pass
    unsigned int *tmax8; // [R][P][nt8] upper bound of the bytes of a tile of 4^8 haplotypes (the whole row up to 8 sites) (use8)
    int32_t nt8;         // tiles of 4^8 haplotypes per population row: 1, 4 or 16
    int32_t use8;        // the drift pass is vgx_tau_colsum8_kernel + vgx_tau_drift8_kernel on the one-byte counts
    int64_t *S;          // [R][P][S] susceptible
    int32_t *dChk;       // [R][P][H] infectious deltas as the reference's bounds check books them (pyx:2473)
    int32_t *dApp;       // [R][P][H] infectious deltas as UpdateCompartmentCounts_tau applies them (pyx:2548)
    int64_t *inc;        // [R][inc_cap] list of individual moves: compartment | magnitude << 38 | sign << 61 | applied-only << 62
                         // (mutants and migrants entering a compartment; in sparse mode also every compartment's own net change)
    int64_t inc_cap;
    int32_t inc_shards;  // shards in use: a power of two <= VGX_INC_SHARDS matched to the draw kernel's grid; inc_cap / inc_shards entries each
    unsigned long long *inc_n;  // [R][VGX_INC_SHARDS]
    int64_t *dSi;        // [R][P][S] susceptible deltas
    int64_t *dTot;       // [R][P]    delta of totalInfectious
    int64_t *totInf;     // [R][P]
    int64_t *gI;         // [R]
    double *cd;          // [R][P] contact density
    int32_t *lockON;     // [R][P]
    double *F;           // [R][P]     sum_spn m[pn,spn]^2 cd[spn]/as[spn]
    double *effMig;      // [R][P][P]
    double *Aeff;        // [R][P][Ppad] TRANSPOSED: [spn][tpn] = effMig[tpn][spn] * m[spn][spn], rows zero-padded
    int32_t Ppad;        // P rounded up to a multiple of 32
    int32_t *eff_dirty;  // [R] contact densities changed since effMig/Aeff/F were computed
    double *migIn;       // [R][P][H]  sum_spn Aeff[tpn][spn] * I[spn][hn] (drift of incoming migration)
    double *mutHi;       // [R][P][H]  incoming mutation drift through the high sites (tiled drift, sites > 6), or null
    int32_t mutHi_int;   // all high sites share one rate and equally likely derived states: mutHi holds int32 neighbour sums
    double mutHi_rate;   // ... to be scaled by this rate
    int32_t mutlow_fast; // uniform mutation model, at least two low sites, each with equally likely derived states: vgx_tau_drift_fast_kernel
    int32_t mutlow_same; // ... and all low sites share one rate
    unsigned int *hist;  // [R][P][C][64] compartments of size 1..64 per class (filled by the fast drift kernel for the sieve), or null
    // uniform migration (every off-diagonal migration probability equal to mig_b, hence every diagonal equal to mig_d):
    // effMig[t][s] = b^2 W + (d b - b^2)(w_t + w_s), w = cd / actualSizes, W = sum w, so the incoming migration pressure
    // needs only the two column sums below instead of the [P x P] x [P x H] product
    int32_t mig_uniform;
    double mig_b, mig_d;
    double *colT;        // [R][H] sum_spn I[spn][h]
    double *colTW;       // [R][H] sum_spn w[spn] I[spn][h]
    double *Gout;        // [R][P][CB] out-migration weight of a source population per birth class
    double *dS;          // [R][P][S]  drift of the susceptible compartments
    double *dS_part;     // [R][P][ds_nb][S] the drift kernel's blocks' parts of it (summed in block order: reproducible)
    int32_t ds_nb;       // blocks of the drift kernel per (population, replicate)
    unsigned long long *tau_bits;  // [R] running minimum of the tau candidates (bit pattern)
    double *tau;         // [R]
    double *time_now;    // [R]
    int32_t *active, *ok, *accepted, *deciding, *retry, *step, *error;  // [R]
    // compartments expecting many events in the current try (vgx_tau_draw_big_kernel)
    int64_t *big;        // [R][big_cap] pn * H + hn
    int64_t big_cap;
    unsigned long long *big_n;  // [R]
    int64_t *res;        // [R][16] what the host reads after a step, packed by vgx_tau_finish_kernel: tau (bits),
                         // globalInfectious, counters[8], multievent row range of the step, error
    int32_t *host_flags; // [3 R] accepted, grow as the decide kernel left them, and "the front pass alone found nothing", in pinned HOST memory (or null)
    int64_t *host_res;   // [R][16] the finish kernel's record per replicate, in pinned host memory (or null)
    int32_t *grow;       // [R] the try overflowed the cross-compartment list: the host enlarges it and the SAME try runs again
    int32_t *attempt;    // [R]
    const int64_t *seeds;  // [R]
    int32_t has_mig;
    int32_t mut_uniform;     // every haplotype has the same mRate / hapMutType rows: use mutp[][] below
    double mutp[16][3];      // mRate[s] * w[s][i] / (w[s][0]+w[s][1]+w[s][2]) of the uniform mutation model
    double mut_total;        // sum of mutp
    const double *mutcum;    // [3*sites] running sums of mutp (device)
    double *migcdf;          // [R][P][CB][P*S] running sums of the out-migration channel weights
    int64_t *counters;   // [R][8]: births, recoveries, samples, mutations, immunity, migrations, lockdown switches, events drawn
    int64_t *cnt_try;    // [R][8] tallies of the retry being validated
    // Occupied compartments (sparse states; vgx_tau_drift8_kernel lists them while it streams the bytes, vgx_tau_listscan_kernel draws a
    // try's buckets for them instead of streaming all P x H compartments).  One region per (population, tile of the drift pass, wavefront).
    int32_t *occ;            // [R][P][occ_nreg][VGX_OCC_CAP] haplotype numbers
    unsigned int *occ_n;     // [R][P][occ_nreg] occupied compartments per region (beyond VGX_OCC_CAP the list is incomplete: the region is swept)
    unsigned long long *occ_pop;   // [R][P] occupied compartments of the population as the drift pass saw them (summed and cleared by the finish kernel)
    int32_t occ_nreg, build_occ, use_list, occ_pad;
    // The drift pass over those lists (vgx_tau_drift8s_*: sparse states with uniform migration): the column sums' pass lists the occupied
    // compartments, the drift of the listed ones is formed from their neighbours' bytes, and of the EMPTY compartments only those that a
    // large neighbour or a large column sum could bring below the smallest candidate so far (ChooseTau's minimum is the same minimum).
    int32_t drift_sparse, ds_pad;
    unsigned long long *tI_pt;   // [R][P][nt8] infectious hosts per (population, tile): exact integer sums (any order)
    double *d8s_pk;              // [R][P][8] per population: Bsum, c1, c2, kI, F, weight, kmig, -
    unsigned long long *d8s_bc;  // [R][8] bit patterns of the largest |Bsum|, |c1|, |c2| of the populations (the bound of a column sum's term); [3]: compartments
                                 // whose empty neighbours were formed in this step; [4]: regions on d8s_ovf
    int32_t *d8s_ovf;            // [R][P * occ_nreg] the regions whose lists are incomplete (population * occ_nreg + region), d8s_bc[4] of them
    int32_t *d8s_regmax;         // [R][P][occ_nreg] largest count of a region
    double *d8s_tile;            // [R][2][nt8] hosts of all populations per tile of the rows, and weighted by cd / actualSizes
    int64_t *front;      // [R][P][front_cap] the front pass's lists: compartments that can fall below zero on their own in this try (vgx_tau_front_kernel)
    unsigned int *front_n;   // [R][P] their counts (may exceed front_cap: the rest is found by the try proper); cleared by vgx_tau_decide_kernel
    int32_t front_cap, front_on;
    // A whole ROUND of a step enqueued without the host in between (one replicate, vgx_api.hip): several tries' front passes back to
    // back, the try proper of the first one that finds nothing, the end of the step.  `spec` says on the device how far the round is:
    // 0 = front passes still look for a try that can succeed, 1 = one found nothing: its try proper runs, 2 = over (accepted, an error,
    // or the host has to enlarge a list).  `gate` (set per launch by the host) says which of these states a launch belongs to: 0 = none
    // (every launch runs: the host decides between them, as before), 1 = a front pass, 2 = the try proper, 3 = the end of the step
    // (runs once the try is accepted).  A launch whose state has passed returns at once.
    int32_t *spec;              // [R]
    int32_t gate, gate_pad;
    int32_t phase, phase_pad;   // of a try: 0 = front pass and try proper in one go; 1 = the front pass alone (the decide kernel rejects the try or
                                // reports that the pass found nothing); 2 = the try proper after such a front pass (one replicate: vgx_api.hip)
    unsigned long long *cnt_pop;   // [R][P][8] the events kernel's share of them per population, folded into cnt_try by vgx_tau_decide_kernel
    int64_t *mev;        // [R][mev_cap][6]  num, type, hap, pop, newHap, newPop (rows with num > 0 only)
    int64_t mev_cap;
    unsigned long long *mev_n;     // [R]
    unsigned long long *mev_base;  // [R] rows of the steps already accepted
    unsigned long long *loc_n;     // [R]
    int32_t *loc_rec;    // [R][VGX_LOC_CAP][2]
    double *loc_time;    // [R][VGX_LOC_CAP]
    // sieve of the halving loop (vgx_tau_sieve_kernel): lower bounds on the expected number of compartments that fail
    // the bounds check at tau * 2^-k, k = 0..VGX_SIEVE_K-1
    double *sieve_pop;   // [R][P][VGX_SIEVE_K] the populations' parts of it (histogram form), summed by vgx_tau_sieve_pick_kernel
    double *sieve;       // [R][VGX_SIEVE_K]
    int64_t *sieve_skipped;  // [R] tries skipped so far in this call
    int32_t sieve_on;
    // compartments whose own events alone would take them below zero in the current try (they may be rescued by incoming
    // mutants): re-examined after the scatter kernel (vgx_tau_suspect_kernel).  Overflow falls back to the dense check pass.
    int64_t *suspect;        // [R][suspect_cap][2] pn * H + hn, infectious + own delta (as the check books it)
    int64_t suspect_cap;
    unsigned long long *suspect_n;   // [R]
    int32_t dense_check;     // 1: run vgx_tau_check_kernel over all compartments after every try (validation / fallback)
    // sparse mode (the default): a try writes no dense delta arrays.  Own net changes go to the list `inc`, the bounds check
    // of a compartment's own change is made where it is drawn, compartments found below zero are entered in a small hash
    // table (key = compartment | gen << 38: entries of earlier tries are stale, nothing is ever cleared) in which
    // vgx_tau_arrivals_kernel adds the mutants that arrive there, and the upper bound is checked per population
    // (sum of a population's compartments <= size implies it for each of them; if the sum test fails the try is run again
    // in dense mode, `grow` = 2).  dense mode (sparse = 0): both delta arrays are written completely by every try.
    int32_t sparse;
    // queue of the compartments that may draw events in the current try (vgx_tau_scan_kernel -> vgx_tau_events_kernel):
    // haplotype | bucket << 32, sharded by (population, block of the scan kernel)
    int64_t *q;                      // [R][q_cap]
    int64_t q_cap;                   // a multiple of q_shards
    int64_t q_shards;
    unsigned long long *q_n;         // [R][q_shards]
    int32_t ev_split;                // blocks of the events kernel per shard of the queue
    double big_lam;                  // expected events of a compartment per leap from which its channels are drawn one by one
    uint32_t gen;                    // try counter of this call, 1 .. 2^25 - 1
    unsigned long long *st_key;      // [R][st_size]
    long long *st_val;               // [R][st_size] infectious + own delta + arrivals
    int64_t st_size;                 // power of two >= 2 * suspect_cap
    int64_t *dChkTot;                // [R][P] sum over the population's compartments of the deltas as the check books them
};
