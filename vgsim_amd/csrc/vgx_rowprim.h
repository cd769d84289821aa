// vgx_rowprim.h — primitives of the row-per-replicate kernels (vgx_quad.hip, vgx_quadg.hip): a wavefront runs FOUR replicates,
// one per 16-lane DPP row; lane (row, l) holds entry l of whatever 16-wide chunk its replicate is processing.  The chains below
// add f64 values STRICTLY IN LANE ORDER inside each row (the reference's serial loops: fast_choose.pxi:25-28,
// src/_BirthDeath.pyx:519-528, 537-546) with one v_fmac_f64 (DPP row_newbcast source) per term, every instruction serving the
// four rows.  All of them must be reached with all 64 lanes active (wave-uniform control flow).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_wave.h"

namespace {

// ---- row primitives --------------------------------------------------------------------------------------------
// value of lane (row, j) for a row-uniform j in 0..15 (LDS crossbar, no memory)
static __device__ __forceinline__ int rowget_i32(int v, int j) {
    return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 48u) | (unsigned)j) << 2), v);
}
static __device__ __forceinline__ double rowget_f64(double v, int j) {
    int lo = rowget_i32(__double2loint(v), j), hi = rowget_i32(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ int64_t rowget_i64(int64_t v, int j) {
    int lo = rowget_i32((int)(uint32_t)v, j), hi = rowget_i32((int)(uint32_t)((uint64_t)v >> 32), j);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
// minimum over the 16 lanes of each row (row rotations: every lane ends with its row's minimum)
#define QDPP_ROR(v, n) __builtin_amdgcn_update_dpp(0, (v), 0x120 + (n), 0xf, 0xf, false)
static __device__ __forceinline__ int row_min(int v) {
    v = min(v, QDPP_ROR(v, 8));
    v = min(v, QDPP_ROR(v, 4));
    v = min(v, QDPP_ROR(v, 2));
    v = min(v, QDPP_ROR(v, 1));
    return v;
}
static __device__ __forceinline__ int row_max(int v) { return -row_min(-v); }
// inclusive integer prefix inside each row (lanes without a source receive 0)
static __device__ __forceinline__ int64_t row_iscan(int64_t v) {
    VGX_SCAN_STEPS(VGX_I64_STEP)
    return v;
}
// maximum over the four rows of a row-uniform value (wave-uniform result)
static __device__ __forceinline__ int rows_max(int v) {
    int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// acc + v[0] + ... + v[15] of each row, in lane order: 16 dependent v_fmac_f64 (acc = fma(v[k], 1.0, acc) rounds like
// acc + v[k]), every one of them serving the four rows.  acc row-uniform in and out; all 64 lanes active.
#define QFM(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
static __device__ __forceinline__ double row_sum16(double v, double acc) {
    const double one = 1.0;
    asm volatile("s_nop 1\n\t" QFM(0) QFM(1) QFM(2) QFM(3) QFM(4) QFM(5) QFM(6) QFM(7) QFM(8) QFM(9) QFM(10) QFM(11) QFM(12)
                     QFM(13) QFM(14) QFM(15)
                 : "+v"(acc)
                 : "v"(v), "v"(one));
    return acc;
}
// lane l of each row gets carry + v[0] + ... + v[l] (the serial prefix): the same chain, each lane keeping the running
// sum of its own step (the selects are off the chain's critical path); `total` receives carry + v[0] + ... + v[15].
#define QSTEP(K)                                                                                        \
    asm volatile(QFM(K) : "+v"(acc) : "v"(v), "v"(one));                                                 \
    res = rl_ == K ? acc : res;
static __device__ __forceinline__ double row_scan16(double v, double carry, double &total) {
    const double one = 1.0;
    const int rl_ = threadIdx.x & 15;
    double acc = carry, res = carry;
    asm volatile("s_nop 1\n\t" QFM(0) : "+v"(acc) : "v"(v), "v"(one));
    res = rl_ == 0 ? acc : res;
    QSTEP(1) QSTEP(2) QSTEP(3) QSTEP(4) QSTEP(5) QSTEP(6) QSTEP(7) QSTEP(8) QSTEP(9) QSTEP(10) QSTEP(11) QSTEP(12)
    QSTEP(13) QSTEP(14) QSTEP(15)
    total = acc;
    return res;
}

}  // namespace
