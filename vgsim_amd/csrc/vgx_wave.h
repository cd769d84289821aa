// vgx_wave.h — wavefront-level primitives of the gfx950 kernels (64-lane waves, one wave per workgroup).
//
// seq_sum / seq_scan add f64 values held one-per-lane STRICTLY IN LANE ORDER, i.e. with the rounding
// sequence of the reference's serial loops (fast_choose.pxi:25-28, src/_BirthDeath.pyx:519-528, 537-546).
// Both are chains of v_fmac_f64 with a DPP row_newbcast source (acc = fma(w[k], 1.0, acc) rounds exactly like
// acc + w[k]), one VALU instruction per step; the scan additionally narrows EXEC to lanes >= k
// before the add, so lane L stops accumulating after its own term and ends with the serial prefix
// w[0]+...+w[L] (no compare/select instructions on the dependent chain).  The blocks must be reached with all 64
// lanes active (wave-uniform control flow); EXEC is restored to all-ones at the end of every block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VGX_LANES 64

// compiler-level ordering of cross-lane traffic through LDS / global memory inside one wave; the hardware
// executes a wave's LDS and vector-memory instructions in issue order.
#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

static __device__ __forceinline__ double bcast(double v, int k) {  // value of lane k (k wave-uniform)
    int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ int64_t bcast_i64(int64_t v, int k) {
    int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, k);
    int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), k);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// ---- seq_sum: ONE VALU instruction per chain step ------------------------------------------------------------
// v_fmac_f64 with a DPP source (gfx90a+ "DP ALU DPP", row_newbcast only): acc = fma(v[lane k of the row], 1.0, acc),
// which rounds exactly like acc + v[k].  row_newbcast:k broadcasts inside a row of 16 lanes, so the chain runs row
// by row: row_mask enables only the lanes of the current row (they all hold the same running sum), and between
// rows the sum moves on with row_bcast:15.  The result is made wave-uniform at the end.
#define VGX_FM(K, RM) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:" #RM " bank_mask:0xf\n\t"
#define VGX_FSUM8_LO(RM) "s_nop 1\n\t" VGX_FM(0, RM) VGX_FM(1, RM) VGX_FM(2, RM) VGX_FM(3, RM) VGX_FM(4, RM) VGX_FM(5, RM) VGX_FM(6, RM) VGX_FM(7, RM)
#define VGX_FSUM8_HI(RM) "s_nop 1\n\t" VGX_FM(8, RM) VGX_FM(9, RM) VGX_FM(10, RM) VGX_FM(11, RM) VGX_FM(12, RM) VGX_FM(13, RM) VGX_FM(14, RM) VGX_FM(15, RM)
#define VGX_FSUM_ROW(R, RM, RMNEXT)                                                                       \
    if (n > 16 * R) {                                                                                     \
        asm volatile(VGX_FSUM8_LO(RM) : "+v"(acc) : "v"(v), "v"(one));                                     \
        if (n > 16 * R + 8) asm volatile(VGX_FSUM8_HI(RM) : "+v"(acc) : "v"(v), "v"(one));                 \
        if (R < 3 && n > 16 * R + 16) {                                                                   \
            int alo = __double2loint(acc), ahi = __double2hiint(acc);                                     \
            alo = __builtin_amdgcn_update_dpp(alo, alo, 0x142, RMNEXT, 0xf, false);                       \
            ahi = __builtin_amdgcn_update_dpp(ahi, ahi, 0x142, RMNEXT, 0xf, false);                       \
            acc = __hiloint2double(ahi, alo);                                                             \
        }                                                                                                 \
    }

// One row (16 lanes, R = 0..3) of the chain: acc + v[16R] + ... + v[16R + m - 1], wave-uniform in and out
// (m = entries of the row that count, 1..16; lanes beyond MUST hold +0.0).
#define VGX_ROW_CASE(R, RM)                                                                  \
    if (row == R) {                                                                          \
        asm volatile(VGX_FSUM8_LO(RM) : "+v"(acc) : "v"(v), "v"(one));                        \
        if (m > 8) asm volatile(VGX_FSUM8_HI(RM) : "+v"(acc) : "v"(v), "v"(one));             \
    }
static __device__ __forceinline__ double row_chain(double v, int row, int m, double acc) {
    const double one = 1.0;
    VGX_ROW_CASE(0, 0x1) VGX_ROW_CASE(1, 0x2) VGX_ROW_CASE(2, 0x4) VGX_ROW_CASE(3, 0x8)
    int lo = __builtin_amdgcn_readlane(__double2loint(acc), row * 16);
    int hi = __builtin_amdgcn_readlane(__double2hiint(acc), row * 16);
    return __hiloint2double(hi, lo);
}

// acc + v[0] + v[1] + ... + v[n-1] in lane order (wave-uniform result).  Lanes >= n MUST hold +0.0 (steps run in
// groups of 8); acc must be wave-uniform; all 64 lanes active.
static __device__ __forceinline__ double seq_sum(double v, int n, double acc) {
    n = __builtin_amdgcn_readfirstlane(n);
    if (n <= 0) return acc;
    const double one = 1.0;
    VGX_FSUM_ROW(0, 0x1, 0x2) VGX_FSUM_ROW(1, 0x2, 0x4) VGX_FSUM_ROW(2, 0x4, 0x8) VGX_FSUM_ROW(3, 0x8, 0x0)
    int src = ((n - 1) >> 4) << 4;   // a lane of the last row that ran
    int lo = __builtin_amdgcn_readlane(__double2loint(acc), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(acc), src);
    return __hiloint2double(hi, lo);
}

// ---- seq_scan: the same fmac chain with EXEC narrowed to lanes >= k before step k ---------------------------------
// All lanes of a row run the identical chain; narrowing EXEC (s_lshl_b64 exec, -1, k) before step k makes lane L stop
// after its own term, so it ends with the serial prefix w[0] + ... + w[L].  A DPP instruction needs 5 wait states after
// an EXEC write (s_nop 4); other waves of the SIMD issue meanwhile, so at >= 3 waves/SIMD a step costs one VALU slot
// (measured, tools/scan_dpp_test.hip: 3.8 ns per step and SIMD against 6.9 ns for the 2 x v_readlane + v_add_f64 step
// used before, bit-identical results).
#define VGX_FX(K, LANE, RM) "s_lshl_b64 exec, -1, " #LANE "\n\ts_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:" #RM " bank_mask:0xf\n\t"
#define VGX_FSCAN8_LO(B, RM) "s_nop 1\n\t" VGX_FX(0, B+0, RM) VGX_FX(1, B+1, RM) VGX_FX(2, B+2, RM) VGX_FX(3, B+3, RM) \
    VGX_FX(4, B+4, RM) VGX_FX(5, B+5, RM) VGX_FX(6, B+6, RM) VGX_FX(7, B+7, RM) "s_mov_b64 exec, -1\n\t"
#define VGX_FSCAN8_HI(B, RM) "s_nop 1\n\t" VGX_FX(8, B+8, RM) VGX_FX(9, B+9, RM) VGX_FX(10, B+10, RM) VGX_FX(11, B+11, RM) \
    VGX_FX(12, B+12, RM) VGX_FX(13, B+13, RM) VGX_FX(14, B+14, RM) VGX_FX(15, B+15, RM) "s_mov_b64 exec, -1\n\t"
#define VGX_FSCAN_ROW(R, B, RM, RMNEXT)                                                                     \
    if (n > B && k0 < B + 8) asm volatile(VGX_FSCAN8_LO(B, RM) : "+v"(acc) : "v"(v), "v"(one) : "scc");      \
    if (n > B + 8 && k0 < B + 16) asm volatile(VGX_FSCAN8_HI(B, RM) : "+v"(acc) : "v"(v), "v"(one) : "scc"); \
    if (R < 3 && n > B + 16) {                                                                              \
        int alo = __double2loint(acc), ahi = __double2hiint(acc);                                           \
        alo = __builtin_amdgcn_update_dpp(alo, alo, 0x142, RMNEXT, 0xf, false);                             \
        ahi = __builtin_amdgcn_update_dpp(ahi, ahi, 0x142, RMNEXT, 0xf, false);                             \
        acc = __hiloint2double(ahi, alo);                                                                   \
    }

// lane L (k0 <= L < n) gets carry + v[k0'] + ... + v[L] with k0' = k0 rounded down to a multiple of 8:
// lanes in [k0', k0) and lanes >= n MUST hold +0.0.  Lanes >= n-1 end with the total.  carry must be wave-uniform;
// all 64 lanes active.
static __device__ __forceinline__ double seq_scan(double v, int n, double carry, int k0 = 0) {
    n = __builtin_amdgcn_readfirstlane(n);
    k0 = __builtin_amdgcn_readfirstlane(k0);
    double acc = carry;
    if (n <= 0) return acc;
    const double one = 1.0;
    VGX_FSCAN_ROW(0, 0, 0x1, 0x2) VGX_FSCAN_ROW(1, 16, 0x2, 0x4) VGX_FSCAN_ROW(2, 32, 0x4, 0x8) VGX_FSCAN_ROW(3, 48, 0x8, 0x0)
    // rows after the last one that ran still hold the carry: give every lane >= n the total
    int tlo = __builtin_amdgcn_readlane(__double2loint(acc), n - 1);
    int thi = __builtin_amdgcn_readlane(__double2hiint(acc), n - 1);
    return ((int)__lane_id() >= n) ? __hiloint2double(thi, tlo) : acc;
}

// ---- order-free wave scans on the DPP network (no LDS traffic) ----------------------------------------------
// Inclusive prefix over the 64 lanes in 6 steps: row_shr 1/2/4/8 inside each row of 16 lanes (lanes without a
// source receive 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (the gfx9 scan).
#define VGX_DPP_SHR(v, d) __builtin_amdgcn_update_dpp(0, (v), 0x110 + (d), 0xf, 0xf, true)
#define VGX_DPP_BC15(v) __builtin_amdgcn_update_dpp(0, (v), 0x142, 0xa, 0xf, false)
#define VGX_DPP_BC31(v) __builtin_amdgcn_update_dpp(0, (v), 0x143, 0xc, 0xf, false)
#define VGX_SCAN_STEPS(STEP) STEP(VGX_DPP_SHR, 1) STEP(VGX_DPP_SHR, 2) STEP(VGX_DPP_SHR, 4) STEP(VGX_DPP_SHR, 8)
#define VGX_I64_STEP(F, D)                                                                  \
    {                                                                                       \
        int lo = F((int)(uint32_t)v, D), hi = F((int)(uint32_t)((uint64_t)v >> 32), D);      \
        v += (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);                      \
    }
#define VGX_I64_STEP0(F)                                                                    \
    {                                                                                       \
        int lo = F((int)(uint32_t)v), hi = F((int)(uint32_t)((uint64_t)v >> 32));           \
        v += (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);                      \
    }
#define VGX_F64_STEP(F, D)                                                                  \
    {                                                                                       \
        int lo = F(__double2loint(v), D), hi = F(__double2hiint(v), D);                     \
        v += __hiloint2double(hi, lo);                                                      \
    }
#define VGX_F64_STEP0(F)                                                                    \
    {                                                                                       \
        int lo = F(__double2loint(v)), hi = F(__double2hiint(v));                           \
        v += __hiloint2double(hi, lo);                                                      \
    }

// inclusive integer scan across the wave (int64 addition is associative: any order is exact)
static __device__ __forceinline__ int64_t iscan(int64_t v, int lane) {
    (void)lane;
    VGX_SCAN_STEPS(VGX_I64_STEP)
    VGX_I64_STEP0(VGX_DPP_BC15)
    VGX_I64_STEP0(VGX_DPP_BC31)
    return v;
}

// inclusive f64 scan across the wave in tree order (FAST mode only: NOT the reference's rounding sequence)
static __device__ __forceinline__ double fscan(double v) {
    VGX_SCAN_STEPS(VGX_F64_STEP)
    VGX_F64_STEP0(VGX_DPP_BC15)
    VGX_F64_STEP0(VGX_DPP_BC31)
    return v;
}
