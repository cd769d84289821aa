// vgx_flat.h — chains over values held ONE PER LANE of a lone wavefront (the latency kernels vgx_solo.hip and vgx_lone.hip).
//
// Every sequential f64 sum of the reference (fast_choose.pxi:22-28, src/_BirthDeath.pyx:385-392, 519-528, 537-546) is one
// v_fmac_f64 with a DPP row_newbcast source per term: acc = fma(w[k], m, acc) with m = 1.0 rounds exactly like acc + w[k].  Serial
// PREFIX sums use a per-lane multiplier m[k] = (lane & 15) >= k ? 1.0 : 0.0 — lane l adds +0.0 from its own step on (x + 0.0 = x),
// so it ends with w[0] + ... + w[l] and no select sits on the chain.  Also here: the division sequences whose result is the IEEE
// quotient (checked on the device by vgx_test_div_by_const) and small wave-uniform helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vgx_wave.h"

struct Masks { double m[16]; };   // m[k] = (lane & 15) >= k ? 1.0 : 0.0

// ---- chains ---------------------------------------------------------------------------------------------------------------
// One asm statement per row of 16 steps, safe by construction: the leading s_nop 1 gives the two wait states a DPP read of a VGPR
// needs after a VALU write of it (the compiler's hazard recogniser cannot see into an asm statement, and whatever it schedules
// right before one — a spill reload, a phi copy — may write the chain's source), and nothing can be scheduled between the steps.
// (The other DPP hazard, five wait states after a VALU write of EXEC, needs a v_cmpx, which hipcc does not emit for gfx9 targets;
// tools/isa_hazard_scan.py checks both on the shipped code objects at every build.)  Steps run in groups of four, a row of
// n <= 4 / 8 / 12 terms leaves early (entries beyond n MUST hold +0.0: they are added when n is not a multiple of four).
#define SOLO_FM(K, MUL, RM) "v_fmac_f64_dpp %[acc], %[v], %[" MUL "] row_newbcast:" #K " row_mask:" RM " bank_mask:0xf\n\t"
#define SOLO_EXIT(N) "s_cmp_le_i32 %[n], " #N "\n\ts_cbranch_scc1 .Lsolo_done%=\n\t"
#define SOLO_SCAN16(RM)                                                                                                        \
    asm volatile("s_nop 1\n\t" SOLO_FM(0, "m0", RM) SOLO_FM(1, "m1", RM) SOLO_FM(2, "m2", RM) SOLO_FM(3, "m3", RM) SOLO_EXIT(4)  \
                 SOLO_FM(4, "m4", RM) SOLO_FM(5, "m5", RM) SOLO_FM(6, "m6", RM) SOLO_FM(7, "m7", RM) SOLO_EXIT(8)                 \
                 SOLO_FM(8, "m8", RM) SOLO_FM(9, "m9", RM) SOLO_FM(10, "m10", RM) SOLO_FM(11, "m11", RM) SOLO_EXIT(12)            \
                 SOLO_FM(12, "m12", RM) SOLO_FM(13, "m13", RM) SOLO_FM(14, "m14", RM) SOLO_FM(15, "m15", RM)                      \
                 ".Lsolo_done%=:\n\t"                                                                                             \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [m0] "v"(M.m[0]), [m1] "v"(M.m[1]), [m2] "v"(M.m[2]), [m3] "v"(M.m[3]), [m4] "v"(M.m[4]),          \
                   [m5] "v"(M.m[5]), [m6] "v"(M.m[6]), [m7] "v"(M.m[7]), [m8] "v"(M.m[8]), [m9] "v"(M.m[9]), [m10] "v"(M.m[10]),  \
                   [m11] "v"(M.m[11]), [m12] "v"(M.m[12]), [m13] "v"(M.m[13]), [m14] "v"(M.m[14]), [m15] "v"(M.m[15]), [n] "s"(nn) \
                 : "scc")
#define SOLO_SUM16(RM)                                                                                                         \
    asm volatile("s_nop 1\n\t" SOLO_FM(0, "mu", RM) SOLO_FM(1, "mu", RM) SOLO_FM(2, "mu", RM) SOLO_FM(3, "mu", RM) SOLO_EXIT(4)  \
                 SOLO_FM(4, "mu", RM) SOLO_FM(5, "mu", RM) SOLO_FM(6, "mu", RM) SOLO_FM(7, "mu", RM) SOLO_EXIT(8)                 \
                 SOLO_FM(8, "mu", RM) SOLO_FM(9, "mu", RM) SOLO_FM(10, "mu", RM) SOLO_FM(11, "mu", RM) SOLO_EXIT(12)              \
                 SOLO_FM(12, "mu", RM) SOLO_FM(13, "mu", RM) SOLO_FM(14, "mu", RM) SOLO_FM(15, "mu", RM)                          \
                 ".Lsolo_done%=:\n\t"                                                                                             \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [mu] "v"(mu), [n] "s"(nn)                                                                          \
                 : "scc")

static __device__ __forceinline__ bool any_lane(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
static __device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
static __device__ __forceinline__ int64_t uni_i64(int64_t v) {
    int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
static __device__ __forceinline__ double uni_f64(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the same with exactly four steps and no way out: shapes of at most four terms (TINY instantiations) save the compare and the branch
#define SOLO_SCAN4(RM)                                                                                                        \
    asm volatile("s_nop 1\n\t" SOLO_FM(0, "m0", RM) SOLO_FM(1, "m1", RM) SOLO_FM(2, "m2", RM) SOLO_FM(3, "m3", RM)            \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [m0] "v"(M.m[0]), [m1] "v"(M.m[1]), [m2] "v"(M.m[2]), [m3] "v"(M.m[3]))
#define SOLO_SUM4(RM)                                                                                                         \
    asm volatile("s_nop 1\n\t" SOLO_FM(0, "mu", RM) SOLO_FM(1, "mu", RM) SOLO_FM(2, "mu", RM) SOLO_FM(3, "mu", RM)            \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [mu] "v"(mu))

// the running sum of row r-1 (its lane 15) moves to the lanes of row r
static __device__ __forceinline__ double row_carry(double acc, int which) {
    int lo = __double2loint(acc), hi = __double2hiint(acc);
    if (which == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x2, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x2, 0xf, false); }
    else if (which == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x4, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x4, 0xf, false); }
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x8, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x8, 0xf, false); }
    return __hiloint2double(hi, lo);
}

// ---- whole rows, no way out ----------------------------------------------------------------------------------------------
// The same chains in whole rows of 16 steps: `rows` (1..4, wave-uniform) rows run completely, entries beyond n MUST hold +0.0 (a row
// that is partly beyond n adds zeros: x + 0.0 = x).  No compare-and-branch inside a row and one test for the usual case of four
// rows: a wavefront that runs alone pays four to five cycles per INSTRUCTION whatever it is, and the early exits of flat_chain are
// twelve scalar instructions per row.  Only the first row's statement carries the s_nop: between the rows nothing writes the chain's
// DPP source (the value stays live in its register; tools/isa_hazard_scan.py checks the shipped code).
#define FLAT_ROWSCAN(RM, NOP)                                                                                                  \
    asm volatile(NOP SOLO_FM(0, "m0", RM) SOLO_FM(1, "m1", RM) SOLO_FM(2, "m2", RM) SOLO_FM(3, "m3", RM)                         \
                 SOLO_FM(4, "m4", RM) SOLO_FM(5, "m5", RM) SOLO_FM(6, "m6", RM) SOLO_FM(7, "m7", RM)                             \
                 SOLO_FM(8, "m8", RM) SOLO_FM(9, "m9", RM) SOLO_FM(10, "m10", RM) SOLO_FM(11, "m11", RM)                         \
                 SOLO_FM(12, "m12", RM) SOLO_FM(13, "m13", RM) SOLO_FM(14, "m14", RM) SOLO_FM(15, "m15", RM)                     \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [m0] "v"(M.m[0]), [m1] "v"(M.m[1]), [m2] "v"(M.m[2]), [m3] "v"(M.m[3]), [m4] "v"(M.m[4]),          \
                   [m5] "v"(M.m[5]), [m6] "v"(M.m[6]), [m7] "v"(M.m[7]), [m8] "v"(M.m[8]), [m9] "v"(M.m[9]), [m10] "v"(M.m[10]),  \
                   [m11] "v"(M.m[11]), [m12] "v"(M.m[12]), [m13] "v"(M.m[13]), [m14] "v"(M.m[14]), [m15] "v"(M.m[15]))
#define FLAT_ROWSUM(RM, NOP)                                                                                                   \
    asm volatile(NOP SOLO_FM(0, "mu", RM) SOLO_FM(1, "mu", RM) SOLO_FM(2, "mu", RM) SOLO_FM(3, "mu", RM)                         \
                 SOLO_FM(4, "mu", RM) SOLO_FM(5, "mu", RM) SOLO_FM(6, "mu", RM) SOLO_FM(7, "mu", RM)                             \
                 SOLO_FM(8, "mu", RM) SOLO_FM(9, "mu", RM) SOLO_FM(10, "mu", RM) SOLO_FM(11, "mu", RM)                           \
                 SOLO_FM(12, "mu", RM) SOLO_FM(13, "mu", RM) SOLO_FM(14, "mu", RM) SOLO_FM(15, "mu", RM)                         \
                 : [acc] "+v"(acc)                                                                                               \
                 : [v] "v"(v), [mu] "v"(mu))
#define FLAT_ROW(RM, NOP) do { if (SCAN) FLAT_ROWSCAN(RM, NOP); else FLAT_ROWSUM(RM, NOP); } while (0)
// SCAN: lane l ends with carry + v[0] + ... + v[l] (lanes of rows that did not run keep the carry).  SUM: every lane of the last row
// that ran ends with carry + v[0] + ... + v[16 rows - 1].
template <bool SCAN>
static __device__ __forceinline__ double flat_rows(double v, int rows, double carry, const Masks &M) {
    double acc = carry;
    const double mu = 1.0;
    rows = uni_i32(rows);
    FLAT_ROW("0x1", "s_nop 1\n\t");
    if (rows == 4) {
        acc = row_carry(acc, 1); FLAT_ROW("0x2", "");
        acc = row_carry(acc, 2); FLAT_ROW("0x4", "");
        acc = row_carry(acc, 3); FLAT_ROW("0x8", "");
    } else if (rows > 1) {
        acc = row_carry(acc, 1); FLAT_ROW("0x2", "");
        if (rows > 2) { acc = row_carry(acc, 2); FLAT_ROW("0x4", ""); }
    }
    return acc;
}

// SCAN: lane l < n ends with carry + v[0] + ... + v[l] (the serial prefix); lanes >= n-1 of the last row that ran hold the total.
// SUM: every lane of the last row that ran ends with carry + v[0] + ... + v[n-1].  One value per lane, lanes >= n hold +0.0,
// carry wave-uniform, n wave-uniform (1..64), all lanes active.
template <bool SCAN, bool TINY = false, bool UNIT = false>
static __device__ __forceinline__ double flat_chain(double v, int n, double carry, const Masks &M) {
    if (UNIT) return carry + v;   // one term (lane 0 holds it; the other lanes' sums are never read)
    double acc = carry;
    const double mu = 1.0;
    if (TINY) {   // n <= 4
        if (SCAN) SOLO_SCAN4("0x1"); else SOLO_SUM4("0x1");
        return acc;
    }
    n = uni_i32(n);
    int nn = n;
    if (SCAN) SOLO_SCAN16("0x1"); else SOLO_SUM16("0x1");
    if (n > 16) {
        acc = row_carry(acc, 1); nn = n - 16;
        if (SCAN) SOLO_SCAN16("0x2"); else SOLO_SUM16("0x2");
        if (n > 32) {
            acc = row_carry(acc, 2); nn = n - 32;
            if (SCAN) SOLO_SCAN16("0x4"); else SOLO_SUM16("0x4");
            if (n > 48) {
                acc = row_carry(acc, 3); nn = n - 48;
                if (SCAN) SOLO_SCAN16("0x8"); else SOLO_SUM16("0x8");
            }
        }
    }
    return acc;
}

// ---- two independent chains of 64 steps, interleaved ----------------------------------------------------------------------------
// A dependent v_fmac_f64 is ready after ~7 cycles, a lone wavefront can issue one every 4: two chains that do not depend on each other
// (BirthRate's sum and the migPopRate prefix sums after a birth: vgx_lone.hip) run in the time of one when their steps alternate.  ONE
// asm statement: the accumulators are fixed registers (the row-to-row moves address their halves, which an operand cannot), every DPP
// read keeps its two wait states by construction (tools/isa_hazard_scan.py checks the shipped code).
// A: every lane of row 3 ends with va[0] + ... + va[63] (SUM).  B: lane l ends with vb[0] + ... + vb[l] (SCAN).  Both from +0.0.
#define FLAT2_A(K, RM) "v_fmac_f64_dpp v[200:201], %[va], %[mu] row_newbcast:" #K " row_mask:" RM " bank_mask:0xf\n\t"
#define FLAT2_B(K, RM) "v_fmac_f64_dpp v[202:203], %[vb], %[m" #K "] row_newbcast:" #K " row_mask:" RM " bank_mask:0xf\n\t"
#define FLAT2_S(K, RM) FLAT2_A(K, RM) FLAT2_B(K, RM)
#define FLAT2_ROW(RM) FLAT2_S(0, RM) FLAT2_S(1, RM) FLAT2_S(2, RM) FLAT2_S(3, RM) FLAT2_S(4, RM) FLAT2_S(5, RM) FLAT2_S(6, RM) FLAT2_S(7, RM) \
                      FLAT2_S(8, RM) FLAT2_S(9, RM) FLAT2_S(10, RM) FLAT2_S(11, RM) FLAT2_S(12, RM) FLAT2_S(13, RM) FLAT2_S(14, RM) FLAT2_S(15, RM)
#define FLAT2_MV(R, RM) "v_mov_b32_dpp " R ", " R " row_bcast:15 row_mask:" RM " bank_mask:0xf\n\t"
#define FLAT2_NEXT(RM) "s_nop 0\n\t" FLAT2_MV("v200", RM) FLAT2_MV("v201", RM) FLAT2_MV("v202", RM) FLAT2_MV("v203", RM)
static __device__ __forceinline__ void flat_two64(double va, double vb, const Masks &M, double &sumA, double &scanB) {
    const double mu = 1.0;
    double oa, ob;
    asm volatile("v_mov_b64 v[200:201], 0\n\tv_mov_b64 v[202:203], 0\n\ts_nop 1\n\t"
                 FLAT2_ROW("0x1") FLAT2_NEXT("0x2") FLAT2_ROW("0x2") FLAT2_NEXT("0x4") FLAT2_ROW("0x4") FLAT2_NEXT("0x8") FLAT2_ROW("0x8")
                 "v_mov_b64 %[oa], v[200:201]\n\tv_mov_b64 %[ob], v[202:203]\n\t"
                 : [oa] "=&v"(oa), [ob] "=&v"(ob)
                 : [va] "v"(va), [vb] "v"(vb), [mu] "v"(mu), [m0] "v"(M.m[0]), [m1] "v"(M.m[1]), [m2] "v"(M.m[2]), [m3] "v"(M.m[3]),
                   [m4] "v"(M.m[4]), [m5] "v"(M.m[5]), [m6] "v"(M.m[6]), [m7] "v"(M.m[7]), [m8] "v"(M.m[8]), [m9] "v"(M.m[9]),
                   [m10] "v"(M.m[10]), [m11] "v"(M.m[11]), [m12] "v"(M.m[12]), [m13] "v"(M.m[13]), [m14] "v"(M.m[14]), [m15] "v"(M.m[15])
                 : "v200", "v201", "v202", "v203");
    sumA = oa; scanB = ob;
}

// The same for two SUMS over `rows` whole rows (1..4, wave-uniform) from wave-uniform carries: every lane of the last row that ran ends with
// ca + va[0] + ... + va[16 rows - 1] / cb + vb[0] + ...; entries beyond the chains' lengths MUST hold +0.0.
#define FLAT2S_B(K, RM) "v_fmac_f64_dpp v[202:203], %[vb], %[mu] row_newbcast:" #K " row_mask:" RM " bank_mask:0xf\n\t"
#define FLAT2S_S(K, RM) FLAT2_A(K, RM) FLAT2S_B(K, RM)
#define FLAT2S_ROW(RM) FLAT2S_S(0, RM) FLAT2S_S(1, RM) FLAT2S_S(2, RM) FLAT2S_S(3, RM) FLAT2S_S(4, RM) FLAT2S_S(5, RM) FLAT2S_S(6, RM) FLAT2S_S(7, RM) \
                       FLAT2S_S(8, RM) FLAT2S_S(9, RM) FLAT2S_S(10, RM) FLAT2S_S(11, RM) FLAT2S_S(12, RM) FLAT2S_S(13, RM) FLAT2S_S(14, RM) FLAT2S_S(15, RM)
#define FLAT2S_EXIT(N) "s_cmp_le_i32 %[rows], " #N "\n\ts_cbranch_scc1 .Lflat2s_done%=\n\t"
static __device__ __forceinline__ void flat_two_sums(double va, double vb, int rows, double ca, double cb, double &sumA, double &sumB) {
    const double mu = 1.0;
    double oa, ob;
    rows = uni_i32(rows);
    asm volatile("v_mov_b64 v[200:201], %[ca]\n\tv_mov_b64 v[202:203], %[cb]\n\ts_nop 1\n\t"
                 FLAT2S_ROW("0x1") FLAT2S_EXIT(1) FLAT2_NEXT("0x2") FLAT2S_ROW("0x2") FLAT2S_EXIT(2) FLAT2_NEXT("0x4") FLAT2S_ROW("0x4")
                 FLAT2S_EXIT(3) FLAT2_NEXT("0x8") FLAT2S_ROW("0x8")
                 ".Lflat2s_done%=:\n\t"
                 "v_mov_b64 %[oa], v[200:201]\n\tv_mov_b64 %[ob], v[202:203]\n\t"
                 : [oa] "=&v"(oa), [ob] "=&v"(ob)
                 : [va] "v"(va), [vb] "v"(vb), [mu] "v"(mu), [ca] "v"(ca), [cb] "v"(cb), [rows] "s"(rows)
                 : "v200", "v201", "v202", "v203", "scc");
    sumA = oa; sumB = ob;
}

// every lane: acc += mu * (v[lane 0 of its row] + ... in order ... + v[lane n-1 of its row]) term by term, mu = 1.0 or 0.0 per
// lane; v must hold the same 16 values in every row (nn <= 16, entries beyond nn +0.0)
template <bool TINY = false, bool UNIT = false>
static __device__ __forceinline__ double rows_chain(double acc, double v, double mu, int nn) {
    if (UNIT) return acc + v;     // one term per row, in the row's lane 0 (the lanes that read the sum: one haplotype, lane 0)
    if (TINY) { SOLO_SUM4("0xf"); return acc; }
    nn = uni_i32(nn);
    SOLO_SUM16("0xf");
    return acc;
}
// every lane (row, l): v[lane 0 of its row] + ... + v[lane l of its row], the serial prefix inside each row (nn <= 16 terms, entries
// beyond nn +0.0); lanes >= nn - 1 of a row end with the row's total
template <bool TINY = false>
static __device__ __forceinline__ double rows_scan(double v, int nn, const Masks &M) {
    double acc = 0.0;
    if (TINY) { SOLO_SCAN4("0xf"); return acc; }
    nn = uni_i32(nn);
    SOLO_SCAN16("0xf");
    return acc;
}

// a / b without the range scaling and the special-case fix-up of the compiler's division sequence (v_div_scale / v_div_fmas /
// v_div_fixup): the same reciprocal refinement and the same final correction step, so for operands whose quotient and intermediate
// products stay clear of overflow and underflow — rates, host counts and numbers in [0, 1) here — it is the same correctly rounded
// quotient (vgx_test_div_by_const compares it with the division on the device).  b = 0 gives NaN instead of +-inf: only lanes whose
// result is never read divide by zero.
// (in two halves: the refined reciprocal of a divisor can be formed as soon as the divisor exists, the quotient when the dividend does)
static __device__ __forceinline__ double refined_rcp(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
static __device__ __forceinline__ double fdiv_y(double a, double b, double y) {
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
static __device__ __forceinline__ double fdiv(double a, double b) { return fdiv_y(a, b, refined_rcp(b)); }

static __device__ __forceinline__ double bperm_f64(double v, int src_lane) {
    int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// n / b for a divisor b whose correctly rounded reciprocal y = RN(1 / b) is known (actualSizes: a parameter).  q0 = RN(n y) is
// within 2 ulp of n / b; q1 = RN(q0 + r0 y) with r0 = RN(n - q0 b) is within 1 ulp (the residual of a quotient that close is
// formed with a relative error of 2^-53 at most, its product with y corrects q0 to 2^-100 before the rounding); r1 = n - q1 b is then
// exact and q2 = RN(q1 + r1 y) is the correctly rounded quotient (Markstein's theorem on the correction of a faithful quotient
// with a correctly rounded reciprocal: the final step of the Itanium division sequences).  No overflow / underflow in this range
// (rates and host counts).  vgx_test_div_by_const runs the sequence on the device against the division for the tests.
static __device__ __forceinline__ double div_by_const(double n, double b, double y) {
    double q = n * y;
    double r = __builtin_fma(-q, b, n);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-q, b, n);
    return __builtin_fma(r, y, q);
}

// lane `k` (a constant) of v takes the wave-uniform value val
#define SOLO_WRITELANE(v, val, k) asm("v_writelane_b32 %0, %1, " #k : "+v"(v) : "s"(val))
