// vgx_quadg.h — what the host (vgx_api.hip) and the general row-per-replicate kernel (vgx_quadg.hip) share: the kernel's
// extra arguments, the limits of the shapes it takes and its LDS layout.
#pragma once
#include <stdint.h>

#define VGX_QG_MAX_P 128      // populations: eight register slots of 16
#define VGX_QG_MAX_S 8        // susceptibility groups
#define VGX_QG_MAX_C 64       // rate classes
#define VGX_QG_MAX_CB 16      // birth classes (transmission rate x susceptibility row): one lane each
#define VGX_QG_MAX_W 32       // words of a population's cold record, 3 S + CB
#define VGX_QG_MAX_SEG 64     // chain segments of the BirthRate program

struct VgxQuadgArgs {
    // effectiveMigration / maxEffectiveBirthMigration (pyx:327-338) for the contact densities cd0 of the call's start state,
    // shared by every replicate whose densities still equal them (vgx_quad_prep_kernel)
    const double *effMig0;    // [P][P]
    const double *mebm0;      // [P]
    const int32_t *has_mig0;  // [1]
    const double *cd0;        // [P]
    // BirthRate program (see vgx_quadg.hip): segment sg continues the sum of segment seg_par[sg] (-1: starts at 0.0) with the
    // P terms of susceptibility group seg_sn[sg] at susceptibility seg_sig[sg]; birth class cb ends with segment cb_seg[cb]
    // (-1: every susceptibility of the class is zero)
    const int32_t *seg_par, *seg_sn;
    const double *seg_sig;
    const int32_t *cb_seg;    // [CB]
    int32_t nseg;
    int32_t W;                // 3 S + CB
    int64_t *cold;            // [R][P][W] cold records
};

struct VgxQuadgLayout {
    int k_as, k_thS, k_thE, k_mult, k_d, k_s, k_tm, k_cbb, k_sig, k_cumul, k_trans, k_segsig, k_jump;
    int k_bidx, k_stype, k_segpar, k_segsn, k_cbseg;
    int rows, row_bytes;
    int s_inf, s_imm, s_mebm, s_cd, s_cc, s_seg, s_rec, s_ts, s_ti, s_cnt, s_inc, s_nocc, s_lock;   // inside a row's block
    int total;
};

// byte offsets of the LDS arrays (PL = 16 x slots); everything 8-byte aligned, the four replicates' blocks at `rows`
static inline __host__ __device__ VgxQuadgLayout vgx_quadg_layout(int PL, int S, int C, int CB, int NSEG, int W = VGX_QG_MAX_W) {
    VgxQuadgLayout L;
    int o = 0;
    L.k_as = o; o += 8 * PL;   L.k_thS = o; o += 8 * PL;   L.k_thE = o; o += 8 * PL;   L.k_mult = o; o += 8 * PL;
    L.k_d = o; o += 8 * C;     L.k_s = o; o += 8 * C;      L.k_tm = o; o += 8 * C;
    L.k_cbb = o; o += 8 * CB;  L.k_sig = o; o += 8 * CB * S;
    L.k_cumul = o; o += 8 * S; L.k_trans = o; o += 8 * S * S;
    L.k_segsig = o; o += 8 * (NSEG > 0 ? NSEG : 1);
    L.k_jump = o; o += 512;
    L.k_bidx = o; o += 4 * C;  L.k_stype = o; o += 4 * C;
    L.k_segpar = o; o += 4 * (NSEG > 0 ? NSEG : 1);   L.k_segsn = o; o += 4 * (NSEG > 0 ? NSEG : 1);
    L.k_cbseg = o; o += 4 * CB;
    o = (o + 15) & ~15;
    L.rows = o;
    int q = 0;
    L.s_inf = q; q += 8 * PL;  L.s_imm = q; q += 8 * PL;  L.s_mebm = q; q += 8 * PL;  L.s_cd = q; q += 8 * PL;
    L.s_cc = q; q += 8 * 8;
    L.s_seg = q; q += 8 * (NSEG > 0 ? NSEG : 1);
    L.s_rec = q; q += 8 * W;     // (the model's own record length: at 64 populations the 192 bytes a row saves are the eighth wavefront of a CU)
    L.s_ts = q; q += 8 * PL;   L.s_ti = q; q += 8 * PL;
    L.s_cnt = q; q += 8 * 8;   L.s_inc = q; q += 8 * 2;
    L.s_nocc = q; q += 4 * PL; L.s_lock = q; q += 4 * PL;
    q = (q + 15) & ~15;
    L.row_bytes = q;
    L.total = L.rows + 4 * q;
    return L;
}
