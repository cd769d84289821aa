// vgx_taus.h — arguments of the on-device tau-leaping step loop of small models (vgx_taus.hip), shared with vgx_api.hip.
#pragma once
#include <stdint.h>
#include "vgx_dev.h"

#define VGX_TAUS_MAX_CELLS 8192    // popNum * hapNum: the compartments and their two delta arrays live in LDS (12 B each)
#define VGX_TAUS_MAX_P 64
#define VGX_TAUS_MAX_S 8
#ifndef VGX_TAUS_TB
#define VGX_TAUS_TB 512
#endif

struct VgxTausArgs {
    VgxDevParams p;
    int64_t R;
    // state of every replicate, read at the start and written back at the end (the arrays the getters read)
    int32_t *I;              // [R][P][H]
    int64_t *S;              // [R][P][S]
    int64_t *totInf;         // [R][P]
    double *cd;              // [R][P]
    int32_t *lock;           // [R][P]
    const int32_t *i_I;      // [P][H] initial state (Restart, pyx:714-738), shared by the replicates
    const int64_t *i_S;      // [P][S]
    const int64_t *seeds;    // [R]
    int64_t iterations, sample_size, attempts;
    float time;
    int32_t start_ok;                 // pyx:2311 on the start state: totalRate + totalMigrationRate != 0 and globalInfectious != 0
    int32_t rates_nonzero_initial;    // ... on the state a Restart restores
    int64_t ev_ptr0, ev_size;
    double t0;
    int64_t gI0, good0;
    int64_t base_cnt[8];              // bCounter, dCounter, sCounter, mCounter, iCounter, migPlus, swapLockdown, 0 before the call
    int32_t mut_uniform;
    double mutp[16][3];               // uniform mutation model: mRate[s] * w[s][i] / (w[s][0] + w[s][1] + w[s][2])
    int64_t *mev;                     // [R][mev_cap][6] multievent rows with num > 0: num, type, hap, pop, newHap, newPop
    int64_t mev_cap;
    int64_t *slog;                    // [R][slog_cap][3] accepted steps: time (bits), first row (+ the step's rejected tries << 56), one past the last row
    int64_t slog_cap;
    int32_t *loc_rec;                 // [R][VGX_LOC_CAP][2]
    double *loc_time;                 // [R][VGX_LOC_CAP]
    unsigned long long *loc_n;        // [R]
    int64_t *res;                     // [R][24] what the host reads after the call (see vgx_taus.hip)
};

enum { TS_TAU = 0, TS_GI, TS_CNT0, TS_EVPTR = 10, TS_ATT, TS_GOOD, TS_RESTARTS, TS_STEPS, TS_MEVROWS, TS_ERROR, TS_TIME, TS_EVPTR0, TS_TRIES };

#define VGX_TAUS_MAX_C 256
#define VGX_TAUS_MAX_CB 64
static inline size_t vgx_taus_lds_bytes(int64_t P, int64_t H, int64_t S, int64_t C, int64_t CB) {
    const size_t PH = (size_t)(P * H);
    return PH * 12 + 8 + (size_t)(P * S) * 24 + (size_t)P * (8 * 7 + 8 + 4) + (size_t)(P * P) * 8 + (size_t)C * 32 + (size_t)CB * 8 * (size_t)(1 + S) +
           (size_t)H * 4 + 64;
}
