// vgx_lone.h — host/device interface of vgx_lone.hip: direct Gillespie for ONE trajectory of a LARGE haplotype space on one wavefront,
// the occupancy lists of every population resident in LDS for the whole call (BASELINE config 3 as a single Simulator.simulate(), and
// config 5's 256 replicates: one wavefront per CU each).
#pragma once
#include <stdint.h>

#define VGX_LONE_MAX_LDS (160 * 1024)
#define VGX_LONE_COLD 16
#define VGX_LONE_ROW 16            // list slots per heap row (one DPP row); a tile of the chains = four rows
#define VGX_LONE_FULL_SITE 77      // error = capacity | this << 8: the LDS heap cannot hold the lists any longer (the host runs the call again
                                   // on the four-replicates-per-wavefront kernel, vgx_api.hip)

struct VgxLoneArgs {
    const double *effMig;      // [P][P] effectiveMigration (vgx_quad_prep_kernel: no population can switch its lockdown state)
    const double *maxEBM;      // [P]    maxEffectiveBirthMigration
    const int32_t *has_mig;    // [1]
    const double *rcpAs;       // [P]    correctly rounded 1 / actualSizes
    int32_t lds_bytes;         // dynamic LDS of the launch
    int32_t exact_rcp_div;     // 1: x / actualSizes through the reciprocal (vgx_flat.h div_by_const); 0: the compiler's division (validation)
    int32_t mut_uniform;       // every haplotype has the same mRate / hapMutType rows: the kernel keeps row 0 in registers
    int32_t general;           // 1: the general form (several rate classes / susceptibility groups, lockdown switches); 0: one class, one group
    // general form: BirthRate as a program of chain segments (vgx_quadg.h: segment sg continues the sum of segment seg_par[sg] with the P
    // terms of group seg_sn[sg] at susceptibility seg_sig[sg]; birth class cb ends with segment cb_seg[cb])
    const int32_t *seg_par, *seg_sn;
    const double *seg_sig;
    const int32_t *cb_seg;     // [CB]
    int32_t nseg;
    int32_t pad_;
};
#define VGX_LONE_MAX_S 16          // general form: susceptibility groups (one lane each), rate classes (64, one lane each), birth classes (16),
#define VGX_LONE_MAX_C 64          // chain segments (64, one lane each); haplotype numbers below 2^26 (the class of a list entry rides in the
#define VGX_LONE_MAX_CB 16         // top six bits of its haplotype word)
#define VGX_LONE_MAX_SEG 64
#define VGX_LONE_HAP_BITS 26

struct VgxLoneLayout {
    int mig;                   // [P][P] f64 migrationRates
    int rng, stage;            // 64 f64; 64 staged log records of 32 bytes
    int rngk, rngs;            // PCG64 jump constants [64][4] u64; stream position and increment [4] u64
    int cold;                  // [VGX_LONE_COLD] 8-byte slots: bookkeeping of the call that the event loop does not touch
    int sus, sst, imm, bc, sig, trans;   // general form: [P][S] susceptible, their copy as of the population's last infect-update, immuneSourcePopRate;
                               // [P][CB] eventHapPopRate[.,.,0] per birth class; [CB][S] susceptibilities; [S][S] suscepTransition (all f64)
    int cum, tend, hap, cnt;   // the heap: per slot the serial prefix sum of hapPopRate as of the population's last infect-update (f64), per
                               // row of 16 slots one f64 (used at the first row of every tile of 64: the prefix sum at the tile's end),
                               // haplotype (i32), infectious count (i32)
    int nrows;                 // rows of VGX_LONE_ROW slots
    int total;
};

static inline __host__ __device__ VgxLoneLayout vgx_lone_layout(int P, int lds_bytes, int S = 0, int CB = 0) {   // S = 0: the one-class form
    VgxLoneLayout L;
    int o = 0;
    L.rng = o; o += 64 * 8;
    L.stage = o; o += 64 * 32;
    L.rngk = o; o += 64 * 32;
    L.rngs = o; o += 32;
    L.cold = o; o += 8 * VGX_LONE_COLD;
    L.mig = o; o += 8 * P * P;
    L.sus = o; o += 8 * P * S;   L.sst = o; o += 8 * P * S;   L.imm = o; o += 8 * P * S;
    L.bc = o; o += 8 * P * CB;   L.sig = o; o += 8 * CB * S;  L.trans = o; o += 8 * S * S;
    const int per_row = VGX_LONE_ROW * 16 + 8;
    int nrows = (lds_bytes - o - 64) / per_row;
    if (nrows < 0) nrows = 0;
    L.nrows = nrows;
    L.cum = o; o += 8 * VGX_LONE_ROW * nrows;
    L.tend = o; o += 8 * nrows;
    L.hap = o; o += 4 * VGX_LONE_ROW * nrows;
    L.cnt = o; o += 4 * VGX_LONE_ROW * nrows;
    L.total = (o + 15) & ~15;
    return L;
}
// rows a list of n entries needs at least (an empty list keeps one row: its first arrival needs no new layout)
static inline __host__ __device__ int vgx_lone_min_rows(int n) { return n > 0 ? (n + VGX_LONE_ROW - 1) / VGX_LONE_ROW : 1; }
