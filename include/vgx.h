/*
 * vgx.h — C ABI of the MI355X-native forward epidemic simulation engine (libvgx.so).
 *
 * This is the drop-in boundary for ONE path of Genomics-HSE/VGsim: the two native entry points that
 * `Simulator.simulate` calls on its engine object (reference src/_interface.py:821-829):
 *     BirthDeathModel.SimulatePopulation      (src/_BirthDeath.pyx:396-429, direct Gillespie)
 *     BirthDeathModel.SimulatePopulation_tau  (src/_BirthDeath.pyx:2293-2346, Poisson tau-leaping)
 * plus what the caller needs to hand the model over and read the results back.  Plain pointers and
 * sizes only; every array is a C-contiguous host buffer with the reference's own name, dtype and shape
 * (numpy arrays of the reference's `cdef class BirthDeathModel`, pyx:47-68).  A binding for the
 * reference (ctypes, or `cdef extern` from Cython) is shown in INTEGRATION.md.
 *
 * One engine = one model shape and `n_replicates` independent trajectories of it (replicate = one
 * seeded run; the classic API uses 1).  Engines are independent; calls on one engine must not overlap.
 * Every function returns VGX_OK (0) or an error code; vgx_last_error() gives the message.
 *
 * Environment switches read by the library (diagnostics and tests; none is needed in production):
 *   VGX_LIST_CAP=n                 caps the capacity of every occupancy list at n entries (exercises the kernels' overflow paths)
 *   VGX_TIMING=1                   vgx_simulate_tau prints its host-side phases on stderr
 *   VGX_SOLO_PLAIN_DIV=1           single-trajectory kernel: x / actualSizes by the compiler's division instead of the reciprocal sequence
 *   VGX_SOLO_GENERAL=1             ... its general BirthRate layout where the compact one would be taken
 *   VGX_SOLO_NO_UNIT=1             ... no one-haplotype / one-population instantiation
 *   VGX_TAU_STEP_KERNELS=1 / 0     tau: always / never the step kernels (default: the on-device step loop for small models)
 *   VGX_TAU_NO_BYTE_DRIFT=1        tau: the two-pass drift instead of the pass on the one-byte counts
 *   VGX_TAU_NO_FRONT=1             tau: no front pass of a try (the compartments that can fall below zero on their own drawn first)
 *   VGX_TAU_NO_OCCLIST=1           tau: a try's scan and front pass always stream all compartments (no lists of the occupied ones)
 *   VGX_TAU_NO_FRONT_ALONE=1       tau, one replicate: the front pass is enqueued together with the try proper, not ahead of it
 *   VGX_TAU_LARGE_MODEL_THRESHOLDS=1  tau: the draw thresholds of large models on a small one
 */
#ifndef VGX_H
#define VGX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vgx_engine vgx_engine;

enum {
    VGX_OK = 0,
    VGX_ERR_ARG = 1,            /* bad argument / unsupported configuration */
    VGX_ERR_HIP = 2,            /* HIP runtime failure (no device, out of memory, launch failure) */
    VGX_ERR_ZERO_WEIGHT = 3,    /* fastChoose hit a zero weight: the reference prints and exits (fast_choose.pxi:5-13) */
    VGX_ERR_CAPACITY = 4,       /* occupancy list / lockdown log / multievent buffer capacity exceeded */
    VGX_ERR_LOOP_GUARD = 5,     /* iteration guard tripped (would be an endless loop upstream) */
    VGX_ERR_CLASSES = 6         /* more distinct per-haplotype rate rows than the engine supports */
};

/* Event types: src/events.pxi:2-8 */
enum { VGX_BIRTH = 0, VGX_DEATH = 1, VGX_SAMPLING = 2, VGX_MUTATION = 3, VGX_SUSCCHANGE = 4,
       VGX_MIGRATION = 5, VGX_MULTITYPE = 6 };

/* pyx:84-90: hapNum must equal 4^sites */
typedef struct vgx_dims {
    int64_t sites, hapNum, popNum, susNum;
} vgx_dims;

/* Model parameters (borrowed, copied by vgx_set_params).  Names, dtypes, shapes: pyx:157-204. */
typedef struct vgx_params {
    const double *bRate, *dRate, *sRate;         /* [H]            pyx:158-160 */
    const double *mRate;                         /* [H][sites]     pyx:161 */
    const double *hapMutType;                    /* [H][sites][3]  pyx:163 */
    const double *susceptibility;                /* [H][S]         pyx:162 */
    const int64_t *suscType;                     /* [H]            pyx:157 */
    const double *suscepTransition;              /* [S][S]         pyx:196 */
    const int64_t *sizes;                        /* [P]            pyx:173 */
    const double *contactDensityBeforeLockdown;  /* [P]            pyx:190 */
    const double *contactDensityAfterLockdown;   /* [P]            pyx:191 */
    const double *startLD, *endLD;               /* [P]            pyx:192-193 */
    const double *samplingMultiplier;            /* [P]            pyx:194 */
    const double *migrationRates;                /* [P][P]         pyx:198 (diagonal ignored: recomputed, pyx:290-295) */
} vgx_params;

/* Compartment state and scalars that persist between simulate() calls (pyx:34-38, 47-49).
 * In vgx_set_state the arrays are read; in vgx_get_state they are written (any may be NULL = skip). */
typedef struct vgx_state {
    int64_t *susceptible;          /* [P][S] */
    int64_t *infectious;           /* [P][H] */
    int64_t *initial_susceptible;  /* [P][S]  snapshot used by Restart (pyx:714-738) */
    int64_t *initial_infectious;   /* [P][H] */
    int64_t *totalSusceptible, *totalInfectious, *lockdownON; /* [P] */
    double *contactDensity;        /* [P]  current value (flips with lockdowns, pyx:698-710) */
    int64_t first_simulation;      /* pyx:34, 435-448 */
    int64_t globalInfectious;
    int64_t bCounter, dCounter, sCounter, mCounter, iCounter, swapLockdown, migPlus, migNonPlus;
    int64_t good_attempt;
    double currentTime, totalRate, totalMigrationRate, tau_l;
    int64_t ev_ptr, ev_size;       /* event-log position/capacity as maintained by Events.CreateEvents (events.pxi:52-68) */
} vgx_state;

/* Options of one simulate call beyond the reference's four arguments. */
typedef struct vgx_run_opts {
    int64_t record_events;   /* 1: keep the event log (default for the classic API); 0: counters/trajectories only */
    int64_t max_loop_factor; /* loop guard: at most max_loop_factor*iterations + 2^20 loop iterations (0 = 1024) */
    int64_t traj_points;     /* >0: bin totalInfectious/totalSusceptible[P] at this many uniform time points */
    double traj_t0, traj_t1; /* time window of the trajectory grid */
    int64_t mode;            /* direct path: 0 = EXACT (the reference's floating-point summation order, bit-exact log);
                                1 = FAST (order-free sums: class-aggregated infection rate, integer prefix search,
                                factored BirthRate, tree scans; same random stream and event semantics, identical
                                integer columns on the same seed); 2 = FAST with a counter-based random stream
                                (Philox4x32-10 keyed by the seed, counter = (draw index, attempt): every draw can be formed
                                on its own; other numbers than PCG64's, so another trajectory of the same law).  Ignored by
                                vgx_simulate_tau. */
    int64_t kernel;          /* direct path: 0 = automatic, 1 = one replicate per wavefront (vgx_direct.hip), 2 = one replicate
                                per lane (vgx_lanes.hip; small models: popNum <= 16, popNum*hapNum <= 1024, susNum <= 8, EXACT),
                                3 = four replicates per wavefront, one per 16-lane row (EXACT): vgx_quad.hip
                                for popNum <= 64, one susceptibility group, one rate class and no possible lockdown switch,
                                else the general form vgx_quadg.hip (popNum <= 128, susNum <= 8, <= 64 rate classes, <= 16
                                transmission/susceptibility classes; also models with a recombination probability); 4 = the general
                                form even where 3 would take the other;
                                5 = one replicate per wavefront with the whole DENSE model in LDS and registers (vgx_solo.hip: the
                                latency kernel of single trajectories; EXACT, hapNum <= 64, popNum <= 128, susNum <= 16).
                                Automatic: 5 for fewer than 2048 replicates of a model it takes (and for more where it beats 3 / 4:
                                small models with several classes, one-class models below 8192 replicates);
                                6 = one replicate per wavefront with the occupancy LISTS in LDS (vgx_lone.hip: single trajectories of
                                large haplotype spaces; EXACT, popNum <= 64; automatic up to 1536 replicates of models 5 does not take);
                                mode 1 on a model the FAST row kernel does not take runs the exact kernels 3 / 4 / 5 / 6 (their output
                                is what FAST promises), mode 2 likewise with those kernels' exact arithmetic on the counter-based
                                stream (4, 5 and 6 take it); 2 for P*H*S <= 4 from 131072 replicates, or when asked for */
    int64_t reserved[2];     /* [0] tau path: 1 = run every try of the halving loop (pyx:2316-2321) instead of starting at the
                                first try that is not certain to be rejected (same accepted steps either way);
                                [1] tau path, how a try's deltas are kept and checked (same draws and decisions in every mode):
                                0 = sparse (default): a list of moves, own deltas checked where they are drawn, no dense arrays;
                                2 = dense delta arrays written by every try, the fused own-delta / arrival tests;
                                1 = dense arrays and the bounds check (pyx:2522-2528) as one pass over all compartments */
} vgx_run_opts;

/* Per-replicate results of the last simulate call. */
typedef struct vgx_counters {
    int64_t ev_ptr;                /* events.ptr after the call */
    int64_t ev_first_new;          /* first log index written by this call (0 if a Restart rewound the log) */
    int64_t loop_iterations;       /* loop iterations incl. rejected migrations (each draws 2 uniforms) */
    int64_t restarts;
    int64_t lockdown_records;
    int64_t error;                 /* VGX_* code raised inside the kernel for this replicate */
    int64_t multievent_rows;       /* tau: rows appended to the multievent log by this call */
    int64_t reserved[5];           /* [0] tau: events drawn; direct: [1] index of the last attempt that drew random
                                      numbers (-1 none), [2] its loop iterations (2 uniforms each); [3] tau: tries of the
                                      halving loop left out as certain rejections */
} vgx_counters;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int vgx_create(const vgx_dims *dims, int64_t n_replicates, int device, vgx_engine **out);
void vgx_destroy(vgx_engine *e);
const char *vgx_last_error(const vgx_engine *e);   /* e may be NULL: message of the last failed vgx_create */
int vgx_device_count(void);

/* ---- model hand-over ------------------------------------------------------------------------ */
int vgx_set_params(vgx_engine *e, const vgx_params *p);
/* Recombination branch of Birth (pyx:575-596): `recombination_probability` (pyx:93, set_coinfection_parameters
 * pyx:1422-1426), `genome_length` (pyx:1409-1417) and sitesPosition[sites] (pyx:98-101, set_mutation_position
 * pyx:1516-1524).  Optional: without this call the probability is 0 and the branch is never taken.  With a non-zero
 * probability direct runs are exact mode only and take the single-trajectory kernel (few replicates of a small model), the general
 * four-replicates-per-wavefront kernel (its *_rec instantiations) or the one-replicate-per-wavefront kernel (any shape). */
int vgx_set_recombination(vgx_engine *e, double recombination_probability, int64_t genome_length,
                          const int64_t *sitesPosition /* [sites], may be NULL when the probability is 0 */);
/* The same state is given to every replicate; replicates differ by their seed only. */
int vgx_set_state(vgx_engine *e, const vgx_state *s);
int vgx_get_state(vgx_engine *e, int64_t replicate, vgx_state *out);
/* user_seed of each replicate: the RNG of attempt k is PCG64(SeedSequence(seed, spawn_key=(k,))),
 * the stream RndmWrapper(seed=(user_seed, k)) creates at pyx:403 / pyx:2310. */
int vgx_set_seeds(vgx_engine *e, const int64_t *seeds /* [n_replicates] */);

/* Optional: puts the state of vgx_set_state on the device in the tau kernels' layout ahead of vgx_simulate_tau (first-call snapshot of
 * PrepareParameters pyx:435-448, conversion and upload of the P x H counts of every replicate), so that a caller who times the simulate
 * call finds its inputs resident.  Valid until the next vgx_set_state / vgx_set_params / simulate call; vgx_simulate_tau does the same
 * work itself when this was not called. */
int vgx_stage_tau(vgx_engine *e);

/* ---- the hot path --------------------------------------------------------------------------- */
/* Replaces BirthDeathModel.SimulatePopulation(iterations, sample_size, float time, attempts), pyx:396. */
int vgx_simulate_direct(vgx_engine *e, int64_t iterations, int64_t sample_size, float time, int64_t attempts,
                        const vgx_run_opts *opts /* may be NULL */);
/* Replaces BirthDeathModel.SimulatePopulation_tau(iterations, sample_size, float time, attempts), pyx:2293. */
int vgx_simulate_tau(vgx_engine *e, int64_t iterations, int64_t sample_size, float time, int64_t attempts,
                     const vgx_run_opts *opts /* may be NULL */);

/* ---- results -------------------------------------------------------------------------------- */
int vgx_get_counters(vgx_engine *e, int64_t replicate, vgx_counters *out);
/* All replicates at once: out[replicate][4] = ev_ptr, loop_iterations, restarts, tau events drawn. */
int vgx_get_counters_all(vgx_engine *e, int64_t *out);
/* Copies log rows [first, first+count) into the caller's Events arrays (events.pxi:26-29). */
int vgx_get_events(vgx_engine *e, int64_t replicate, int64_t first, int64_t count, double *times,
                   int64_t *types, int64_t *haplotypes, int64_t *populations, int64_t *newHaplotypes,
                   int64_t *newPopulations);
/* Lockdown switches recorded by the last call (models.pxi:52-66): up to `cap` rows, returns the count in *n. */
int vgx_get_lockdowns(vgx_engine *e, int64_t replicate, int64_t cap, int64_t *states, int64_t *populations,
                      double *times, int64_t *n);
/* Forward recombination records of the last direct call (Recombination.AddRecombination_forward, models.pxi:82-89):
 * event index, parent haplotypes hi and hi2, recombinant haplotype, breakpoint.  Like upstream, records of failed
 * attempts stay in the list (Restart does not clear `rec`, pyx:714-738). */
int vgx_get_recombinations(vgx_engine *e, int64_t replicate, int64_t cap, int64_t *idevents, int64_t *his,
                           int64_t *hi2s, int64_t *nhis, int64_t *posRecombs, int64_t *n);
/* Tau multievents of the last call (events.pxi:105-152), rows with num > 0 only. */
/* Rejected tries (halvings of tau_l, pyx:2316-2321) of steps [first, first + count) of the last vgx_simulate_tau call: the leap a step
 * made times 2^tries is the tau ChooseTau (pyx:2432-2450) gave it. */
int vgx_get_tau_tries(vgx_engine *e, int64_t replicate, int64_t first, int64_t count, int32_t *out);
int vgx_get_multievents(vgx_engine *e, int64_t replicate, int64_t cap, int64_t *num, double *times, int64_t *types,
                        int64_t *haplotypes, int64_t *populations, int64_t *newHaplotypes,
                        int64_t *newPopulations, int64_t *n);
/* Summary trajectories of the last call: out[replicate][point][population][0=infectious,1=susceptible], f64.
 * `out` is a host pointer, or a device pointer when out_is_device != 0 (e.g. a torch tensor for an RCCL gather). */
int vgx_get_trajectories(vgx_engine *e, double *out, int out_is_device);
/* The same trajectories as 32-bit integers written to a DEVICE buffer of the same shape (compartment totals are whole numbers;
 * refused when a population size is 2^31 or more): the wire format of the ensemble gather, formed without an f64 copy. */
int vgx_get_trajectories_int(vgx_engine *e, int32_t *out_device);
/* Direct calls with a time limit take their `currentTime < time` stop decisions (pyx:407) on the device clock; event times are
 * rebuilt on the host with libm.  Number of replicates (fetched so far) for which the two clocks disagreed on one such
 * decision (an event time within rounding of the limit): the run reported is then the device clock's. */
int64_t vgx_clock_mismatches(const vgx_engine *e);

/* ---- backward pass ------------------------------------------------------------------------- */
/* Replaces BirthDeathModel.GetGenealogy(seed) (pyx:743-1000) with its recorders Mutations / Migrations
 * (models.pxi:1-48): one backward walk over the event log that coalesces the sampled lineages.  Host code (no
 * device, no engine handle needed).  `infectious` is walked back in place exactly as the reference does. */
typedef struct vgx_genealogy_io {
    int64_t popNum, hapNum;
    int64_t sCounter;                      /* number of samples; the tree has 2*sCounter-1 nodes */
    /* event log (events.pxi:24-68) */
    int64_t ev_ptr;
    const double *ev_times;
    const int64_t *ev_types, *ev_haplotypes, *ev_populations, *ev_newHaplotypes, *ev_newPopulations;
    /* multievent rows (events.pxi:105-152) referenced by MULTITYPE events as [haplotypes, populations); may be NULL */
    int64_t mev_rows;
    const int64_t *mev_num; const double *mev_times;
    const int64_t *mev_types, *mev_haplotypes, *mev_populations, *mev_newHaplotypes, *mev_newPopulations;
    int64_t *infectious;                   /* [P][H], state at the end of the simulation (in/out) */
    /* PCG64 position of the reference's self.seed: state hi, lo, increment hi, lo + numpy's buffered 32-bit half */
    uint64_t rng_state[4];
    int64_t rng_has_uint32;
    uint64_t rng_uinteger;
    /* outputs, caller-allocated */
    int64_t *tree, *tree_pop; double *times;            /* [2*sCounter-1]: parent (-1 = root), population, time */
    int64_t mut_cap, mut_n; int64_t *mut_node, *mut_AS, *mut_DS, *mut_site; double *mut_time;
    int64_t mig_cap, mig_n; int64_t *mig_node, *mig_old, *mig_new; double *mig_time;
    int64_t nodes_used;
} vgx_genealogy_io;
int vgx_get_genealogy(vgx_genealogy_io *io, char *errbuf, int64_t errcap);
/* (state, inc) of PCG64(SeedSequence(seed, spawn_key=(attempt,))) after `draws` outputs: where the reference's
 * self.seed stands when GetGenealogy(seed=None) continues the simulation's stream (pyx:766-767). */
void vgx_rng_position(int64_t seed, int64_t attempt, int64_t draws, uint64_t out[4]);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Device time of the last simulate call's kernels, from HIP events on the engine's stream (ms). */
double vgx_last_kernel_ms(const vgx_engine *e);
/* The kernel the last vgx_simulate_direct call ran on, as a value of vgx_run_opts.kernel (1 .. 5; 3 also for the FAST row kernel). */
int vgx_last_direct_kernel(const vgx_engine *e);
/* Number of kernel launches timed by the last simulate call. */
int64_t vgx_last_kernel_launches(const vgx_engine *e);
/* Bytes of device memory held by the engine. */
int64_t vgx_device_bytes(const vgx_engine *e);
/* Diagnostic build only (libvgx built with -DVGX_PROFILE): 16 per-phase shader-cycle sums of the last direct
 * call for one replicate (phase list: tools/profile_phases.py); all zeros in the product build. */
int vgx_get_profile(vgx_engine *e, int64_t replicate, int64_t *out16);

/* ---- the dense propensity row pass (K3) ------------------------------------------------------- */
/* For callers that hold the reference's dense per-population arrays: the infect branch of UpdateRates (pyx:518-528:
 * BirthRate, tEventHapPopRate, hapPopRate, infectPopRate) followed by fastChoose over hapPopRate (fast_choose.pxi:18-31)
 * for `rows` independent (replicate, population) rows at once.  FAST-mode arithmetic: BirthRate factored
 * through rowContact = sum_pn m[pi,pn]^2 * cd[pn] / actualSizes[pn], tree-order sums.  All pointers are host arrays. */
typedef struct vgx_rowscan {
    int64_t rows, H, S;
    const int64_t *infectious;       /* [rows][H]      infectious[pi, :]                                   in  */
    const double *eventRates123;     /* [rows][H][3]   eventHapPopRate[pi, :, 1:4] (pyx:311-314)          in  */
    const int64_t *numToHap;         /* [H]            pyx:105-125 (identity without memory_optimization)  in  */
    const double *bRate;             /* [H]                                                                in  */
    const double *susceptibility;    /* [H][S]                                                             in  */
    const double *rowSusceptible;    /* [rows][S]      susceptible[pi, :] as doubles                       in  */
    const double *rowContact;        /* [rows]                                                             in  */
    const double *u;                 /* [rows]         the random number of fastChoose                     in  */
    double *birthRate, *tEvent, *hapPopRate;   /* [rows][H]  eventHapPopRate[pi,:,0], tEventHapPopRate, hapPopRate  out */
    double *susceptHapPopRate;       /* [rows][H][S]                                                       out */
    double *rowTotal;                /* [rows]         infectPopRate[pi]                                   out */
    int64_t *chosen;                 /* [rows]         index returned by fastChoose                        out */
    double *rnOut;                   /* [rows]         rescaled random number (fc:31)                      out */
} vgx_rowscan;
int vgx_propensity_scan(const vgx_rowscan *io);
/* Measurement: `rows` copies of the caller's first row resident in HBM, `repeats` timed passes after a warm-up; average
 * device time (HIP events) of the row update and of the choice; the first row's outputs are returned. */
int vgx_propensity_scan_bench(const vgx_rowscan *first_row, int64_t rows, int repeats, double *ms_update, double *ms_choose);
const char *vgx_propensity_scan_error(void);

/* ---- test hooks: the samplers of the tau-leap kernels on their own -------------------------------- */
/* Philox4x32-10 (Salmon et al. 2011) for one (counter, key): the host build of the same function, or the device's. */
int vgx_test_philox(int on_device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* n independent draws of the device's Poisson(lam) sampler (what replaces numpy's random_poisson, pyx:2531-2532:
 * inversion below a mean of 10, PTRS from 10 on), draw i from the Philox stream of compartment i under `seed`. */
int vgx_test_poisson(double lam, int64_t n, uint64_t seed, int64_t *out);

/* count quotients n[i] / b[i] formed on the device three ways: q_seq = through the correctly rounded reciprocal of b with two residual
 * corrections (how the single-trajectory kernel divides BirthRate's terms by actualSizes, pyx:390), q_lean = its division sequence
 * without range scaling and special-case fix-up (fastChoose's rescalings, fast_choose.pxi:31), q_div = the division. */
int vgx_test_div_by_const(const double *n, const double *b, int64_t count, double *q_seq, double *q_lean, double *q_div);

#ifdef __cplusplus
}
#endif
#endif /* VGX_H */
