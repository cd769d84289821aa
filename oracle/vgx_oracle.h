/*
 * vgx_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the forward simulators of Genomics-HSE/VGsim
 * (src/_BirthDeath.pyx: SimulatePopulation pyx:396-429, SimulatePopulation_tau pyx:2293-2346 and
 * everything they call; src/fast_choose.pxi; src/events.pxi).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (vgsim_amd/) never does.
 *
 * PARITY STATUS
 *   - Everything downstream of the uniform stream (all arithmetic of /root/reference/src) is pinned:
 *     the .npz files under tests/golden/ were produced by the reference itself, compiled from the sources where they
 *     lie (tests/golden/build_reference.sh + make_golden.py), and this oracle reproduces them bit for
 *     bit (all six rows, sha256-equal on the nine getting_reference.py models).
 *   - The mapping seed -> uniform stream lives in a third-party module that is absent
 *     (mc_lib.rndm.RndmWrapper, mc_lib v0.4.1, pyproject.toml:9,33).  It is restated here from its
 *     published behaviour (numpy PCG64 seeded by SeedSequence(seed, spawn_key=(attempt,))) and checked
 *     against numpy, but the reference's own goldens (testing/reference_*.npy) are missing from the
 *     checkout, so for that one mapping: PARITY UNPINNED.
 *
 * All arrays are caller-owned, C-contiguous (numpy in the tests); the oracle allocates nothing except
 * small scratch vectors.
 *
 * memory_optimization (pyx:105-125, AddMemory pyx:264-274, AddHaplotype pyx:355-377, the lookup in Mutation pyx:651-660) is
 * restated op for op on the direct path, bugs included (AddHaplotype shifts the counts of ONE population only), with one
 * difference: the arrays indexed by program number keep hapNum columns here, so the places where upstream reads or
 * writes past its maxHapNum columns with bounds checking off (PrepareParameters pyx:444-446, Restart pyx:732-735,
 * the lookup's slot currentHapNum when the table is full) see zeros / are skipped instead of foreign memory; AddMemory is
 * then pure bookkeeping (maxHapNum, addMemoryNum).  Upstream has no goldens for the option (check_simulator.py:153-180
 * are commented out): this part of the oracle is a restatement read against the source, not pinned by fixtures.
 */
#ifndef VGX_ORACLE_H
#define VGX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VGO_BIRTH = 0, VGO_DEATH = 1, VGO_SAMPLING = 2, VGO_MUTATION = 3, VGO_SUSCCHANGE = 4,
       VGO_MIGRATION = 5, VGO_MULTITYPE = 6 }; /* ev:2-8 */

enum { VGO_OK = 0, VGO_ERR_ZERO_WEIGHT = 1, VGO_ERR_LOCKDOWN_LOG_FULL = 2, VGO_ERR_MULTIEVENTS_FULL = 3,
       VGO_ERR_BAD_ARG = 4 };

enum { VGO_LOG_LIBM = 0,     /* host libm log(): what the reference calls (pyx:477) */
       VGO_LOG_PORTABLE = 1, /* self-contained fdlibm-style log, bit-identical to the device kernel's */
       /* flag in the same field (not a reference behaviour): the direct path draws its uniforms from the counter-based
          Philox stream of the engine's FAST mode 2 (include/vgx.h) instead of PCG64, to check that mode draw for draw */
       VGO_RNG_PHILOX = 16 };

typedef struct vgo_model {
    /* ---- dimensions (pyx:84-90) ---- */
    int64_t sites, hapNum, popNum, susNum;
    int64_t user_seed;
    /* ---- parameters (pyx:157-204) ---- */
    double *bRate, *dRate, *sRate;            /* [H] */
    double *mRate;                            /* [H][sites] */
    double *hapMutType;                       /* [H][sites][3] */
    double *susceptibility;                   /* [H][S] */
    int64_t *suscType;                        /* [H] */
    double *suscepTransition;                 /* [S][S] */
    int64_t *sizes;                           /* [P] */
    double *contactDensity, *contactDensityBeforeLockdown, *contactDensityAfterLockdown;
    double *startLD, *endLD, *samplingMultiplier; /* [P] */
    double *migrationRates;                   /* [P][P]; diagonal rewritten by UpdateAllRates pyx:290-295 */
    /* ---- compartment state ---- */
    int64_t *susceptible;                     /* [P][S] */
    int64_t *infectious;                      /* [P][H] */
    int64_t *initial_susceptible;             /* [P][S] */
    int64_t *initial_infectious;              /* [P][H] */
    int64_t *totalSusceptible, *totalInfectious, *lockdownON; /* [P] */
    /* ---- rate caches (pyx:150-155, 182-199) ---- */
    double *tmRate;                           /* [H] */
    double *eventHapPopRate;                  /* [P][H][4] */
    double *tEventHapPopRate, *hapPopRate;    /* [P][H] */
    double *susceptHapPopRate;                /* [P][H][S] */
    double *suscepCumulTransition;            /* [S] */
    double *immuneSourcePopRate;              /* [P][S] */
    double *infectPopRate, *immunePopRate, *popRate, *migPopRate, *actualSizes,
           *maxEffectiveBirthMigration;       /* [P] */
    double *effectiveMigration;               /* [P][P] */
    /* ---- scalars (pyx:34-38) ---- */
    int64_t first_simulation, globalInfectious;
    int64_t bCounter, dCounter, sCounter, mCounter, iCounter, swapLockdown, migPlus, migNonPlus, good_attempt;
    double currentTime, totalRate, totalMigrationRate, rn, tau_l;
    /* ---- event log (ev:24-68); capacity managed by the caller exactly as CreateEvents does ---- */
    int64_t ev_size, ev_ptr;
    double *ev_times;
    int64_t *ev_types, *ev_haplotypes, *ev_populations, *ev_newHaplotypes, *ev_newPopulations;
    /* ---- multievents (ev:105-152); may be NULL (then rows are counted but not stored) ---- */
    int64_t mev_size, mev_ptr;
    int64_t *mev_num; double *mev_times;
    int64_t *mev_types, *mev_haplotypes, *mev_populations, *mev_newHaplotypes, *mev_newPopulations;
    /* ---- lockdown log (md:52-66) ---- */
    int64_t loc_cap, loc_n;
    int64_t *loc_states, *loc_populations; double *loc_times;
    /* ---- tau scratch (pyx:225-229); the per-channel arrays of pyx:210-223 are allocated internally ---- */
    double *infectiousAuxTau;                 /* [P][H] */
    double *susceptibleAuxTau;                /* [P][S] */
    int64_t *infectiousDelta;                 /* [P][H] */
    int64_t *susceptibleDelta;                /* [P][S] */
    /* ---- options / diagnostics ---- */
    int64_t sparse;     /* 1: visit only occupied haplotypes (bit-identical); 0: reference order */
    int64_t log_mode;   /* VGO_LOG_* */
    int64_t iterations_done; /* loop iterations incl. rejected migrations (diagnostic) */
    int64_t error;
    uint64_t *occ;      /* [P][ceil(H/64)] occupancy bitmap, used when sparse (caller allocates) */
    /* PCG64 (state, inc) of self.seed when the simulate call returns: GetGenealogy(seed=None) draws on from here */
    uint64_t rng_state_hi, rng_state_lo, rng_inc_hi, rng_inc_lo;
    /* ---- recombination (pyx:93-102, 575-596; recorder md:69-89) ---- */
    double recombination;                     /* recombination_probability; 0 = branch never taken */
    int64_t genome_length;
    int64_t *sitesPosition;                   /* [sites] */
    int64_t rec_cap, rec_n;                   /* forward records; never cleared by Restart (pyx:714-738) */
    int64_t *rec_idevents, *rec_his, *rec_hi2s, *rec_nhis, *rec_posRecombs;
    /* ---- memory_optimization (pyx:105-125); 0: program number == haplotype, the fields below are ignored ---- */
    int64_t memory_optimization;
    int64_t currentHapNum, maxHapNum, addMemoryNum;
    int64_t *hapToNum;                        /* [H] haplotype -> program number */
    int64_t *numToHap;                        /* [H] program number -> haplotype (upstream: maxHapNum entries, grown by AddMemory) */
} vgo_model;

/* pyx:396-429.  `time` is a C float exactly as in the reference signature. */
int vgo_simulate_direct(vgo_model *m, int64_t iterations, int64_t sample_size, float time, int64_t attempts);
/* pyx:2293-2346. */
int vgo_simulate_tau(vgo_model *m, int64_t iterations, int64_t sample_size, float time, int64_t attempts);
/* pyx:279-351 (exposed for intermediate-value tests). */
void vgo_update_all_rates(vgo_model *m);
/* pyx:2301 */
int64_t vgo_prop_num(const vgo_model *m);
/* test hook: rejected tries (halvings of tau_l, pyx:2316-2321) of step `step` (0-based, counted over all attempts) of the last
 * vgo_simulate_tau call; -1 beyond the steps made (or the first 4096) */
int64_t vgo_tau_tries(int64_t step);

/* RNG restatement (mc_lib.rndm.RndmWrapper + numpy PCG64/SeedSequence). */
typedef struct { uint64_t state_hi, state_lo, inc_hi, inc_lo; } vgo_pcg64;
void vgo_pcg64_seed(vgo_pcg64 *g, uint64_t entropy, uint32_t spawn_key);
double vgo_pcg64_double(vgo_pcg64 *g);
uint64_t vgo_pcg64_next64(vgo_pcg64 *g);
void vgo_pcg64_advance(vgo_pcg64 *g, uint64_t delta_hi, uint64_t delta_lo);
int64_t vgo_poisson(vgo_pcg64 *g, double lam); /* numpy random_poisson restated */
double vgo_portable_log(double x);

/* ---- backward pass: GetGenealogy (pyx:743-1000), vgx_oracle_genealogy.c ---- */
typedef struct { vgo_pcg64 g; int64_t has_uint32; uint64_t uinteger; } vgo_gen_rng; /* numpy bitgen front end on PCG64 */
int64_t vgo_hypergeometric(vgo_gen_rng *r, int64_t good, int64_t bad, int64_t sample); /* numpy random_hypergeometric */
typedef struct vgo_genealogy {
    int64_t popNum, hapNum, sCounter;
    /* event chain (ev:24-68) and, for MULTITYPE events, the multievent rows they refer to (ev:105-152) */
    int64_t ev_ptr;
    const double *ev_times;
    const int64_t *ev_types, *ev_haplotypes, *ev_populations, *ev_newHaplotypes, *ev_newPopulations;
    const int64_t *mev_num; const double *mev_times;
    const int64_t *mev_types, *mev_haplotypes, *mev_populations, *mev_newHaplotypes, *mev_newPopulations;
    int64_t *infectious;       /* [P][H] state at the end of the simulation; walked back in place like upstream */
    int64_t *infectiousDelta;  /* [P][H] scratch */
    vgo_gen_rng rng;           /* in: stream position; out: position after the pass */
    /* outputs, caller-allocated */
    int64_t *tree, *tree_pop; double *times;           /* [2*sCounter-1] */
    int64_t mut_cap, mut_n; int64_t *mut_node, *mut_AS, *mut_DS, *mut_site; double *mut_time;
    int64_t mig_cap, mig_n; int64_t *mig_node, *mig_old, *mig_new; double *mig_time;
    int64_t nodes_used;
} vgo_genealogy;
int vgo_get_genealogy(vgo_genealogy *G);

#ifdef __cplusplus
}
#endif
#endif
