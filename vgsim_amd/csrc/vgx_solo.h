// vgx_solo.h — what the host (vgx_api.hip) and the latency kernel of small models (vgx_solo.hip) share: the kernel's extra
// arguments, the limits of the shapes it takes and its LDS layout.
#pragma once
#include <stdint.h>

#define VGX_SOLO_MAX_H 64      // haplotypes: one lane each
#define VGX_SOLO_MAX_P 128     // populations: one lane each in up to two registers
#define VGX_SOLO_MAX_S 16      // susceptibility groups: one lane each
#define VGX_SOLO_MAX_SEG 32    // distinct (group, non-zero susceptibility) pairs: one bit of a lane's path word each
#define VGX_SOLO_MAX_TSEG 64   // chain segments of the BirthRate program beyond 16 populations or haplotypes: one lane each
#define VGX_SOLO_MAX_PASS 64   // passes of two chains each
#define VGX_SOLO_MAX_LDS (160 * 1024)
#define VGX_SOLO_COLD 16
#define VGX_SOLO_ROWS 4        // 16-lane DPP rows of a wavefront

struct VgxSoloArgs {
    // BirthRate (pyx:382-392) as a list of SEGMENTS: segment s = the P terms ((S[pi, sn_s] * sig_s) * m * m * cd) / as of one
    // susceptibility group sn_s at one non-zero susceptibility value sig_s, sorted by group.  A haplotype's sum runs over the
    // segments whose value is its own (its "path"), in this order: the reference's (sn, pn) order with the zero terms left out.
    const int32_t *seg_sn;     // [nseg]
    const double *seg_sig;     // [nseg]
    int32_t nseg;
    const double *rcpAs;       // [P] 1 / actualSizes, correctly rounded (the host's IEEE division)
    // Susceptibility CLASSES: haplotypes with bit-identical susceptibility rows.  With at most VGX_SOLO_ROWS classes every class gets
    // a 16-lane DPP row: lane (c, j) holds term j of class c's BirthRate sum (its non-zero groups in order, P terms each) and group
    // j's susceptibility of class c ("compact" layout, terms = max over the classes of (non-zero groups) * P <= 32).
    const int32_t *hap_cls;    // [H] class of a haplotype
    const int32_t *cls_nnz;    // [VGX_SOLO_ROWS] non-zero groups of a class
    const int32_t *cls_tsn;    // [VGX_SOLO_ROWS][VGX_SOLO_MAX_S] their group numbers, in order
    const double *cls_tsig;    // [VGX_SOLO_ROWS][VGX_SOLO_MAX_S] their susceptibilities
    const double *cls_sigma;   // [VGX_SOLO_ROWS][VGX_SOLO_MAX_S] susceptibility of a class per group (zero padded)
    int32_t n_cls;             // classes (<= VGX_SOLO_ROWS in the compact layout)
    int32_t maxterms;          // max over the classes of (non-zero groups) * P
    int32_t compact;           // 0: general layout; 1 / 2: compact layout with one / two registers of terms
    int32_t exact_rcp_div;     // 1: x / actualSizes through the reciprocal with two exact residual corrections (see vgx_solo.hip);
                               // 0: the compiler's division (validation: VGX_SOLO_PLAIN_DIV=1)
    int32_t mig_in_lds;        // migrationRates [P][P] fits the LDS budget
    // General layout beyond 16 populations or 16 haplotypes: BirthRate per birth class as the program of chain segments of vgx_quadg.h
    // (segment sg continues the sum of segment tseg_par[sg] with the P terms of group tseg_sn[sg] at susceptibility tseg_sig[sg]; birth
    // class cb ends with segment cb_seg[cb]), each segment ONE chain over the population lanes for all haplotypes at once, and two chains
    // that do not depend on each other per pass (vgx_flat.h flat_two_sums).  pass[0..1][k]: the two chains of pass k without the
    // migration rates' sum, pass[2..3][k]: with it (-1: none, -2: the sum of migPopRate, pyx:541-546).
    const int32_t *tseg_par, *tseg_sn;
    const double *tseg_sig;
    const int32_t *cb_seg;     // [CB]
    const int32_t *pass;       // [4][VGX_SOLO_MAX_PASS]
    int32_t tnseg, npass0, npass1, pad_;
};


struct VgxSoloLayout {
    int rowI, rowCum, rowHpr, rowBirth, rowTE;   // [P][H] f64: counts and rate caches of every population's row
    int susS, susSt, susImm;                     // [P][S] f64: susceptible counts, their copy as of the row's last infect-update, immuneSourcePopRate
    int sigma, trans, mrate, hmt;                // [H][S], [S][S], [H][sites], [H][sites][3] f64
    int cd, as;                                  // [P] f64 (uniform reads of the rebuild)
    int mig;                                     // [P][P] f64 or -1
    int rng, stage;                              // 64 f64; 64 staged log records of 32 bytes
    int rngk, rngs;                              // PCG64 jump constants [64][4] u64; stream position and increment [4] u64
    int cold;                                    // [VGX_SOLO_COLD] 8-byte slots: the call's bookkeeping that the event loop does not touch
    int smult;                                   // [P] f64 samplingMultiplier
    int total;
};

static inline __host__ __device__ VgxSoloLayout vgx_solo_layout(int P, int H, int S, int sites, int mig_in_lds) {
    VgxSoloLayout L;
    int o = 0;
    L.rng = o; o += 64 * 8;
    L.stage = o; o += 64 * 32;
    L.rngk = o; o += 64 * 32;
    L.rngs = o; o += 32;
    L.cold = o; o += 8 * VGX_SOLO_COLD;
    L.rowI = o; o += 8 * P * H;   L.rowCum = o; o += 8 * P * H;   L.rowHpr = o; o += 8 * P * H;
    L.rowBirth = o; o += 8 * P * H;   L.rowTE = o; o += 8 * P * H;
    L.susS = o; o += 8 * P * S;   L.susSt = o; o += 8 * P * S;   L.susImm = o; o += 8 * P * S;
    L.sigma = o; o += 8 * H * S;  L.trans = o; o += 8 * S * S;
    L.mrate = o; o += 8 * H * (sites > 0 ? sites : 1);   L.hmt = o; o += 24 * H * (sites > 0 ? sites : 1);
    L.cd = o; o += 8 * P;         L.as = o; o += 8 * P;         L.smult = o; o += 8 * P;
    L.mig = -1;
    if (mig_in_lds) { L.mig = o; o += 8 * P * P; }
    L.total = (o + 15) & ~15;
    return L;
}
