// vgx_rowscan.hip — K3: the propensity row pass in the reference's DENSE layout, as a streaming kernel.
//
// What the reference does for every infect-type event in population pi:
//   UpdateRates(pi, infect=True) (src/_BirthDeath.pyx:518-528): for every haplotype hn of the row
//       eventHapPopRate[pi,hn,0] = BirthRate(pi,hn)                    (pyx:382-392, writes susceptHapPopRate[pi,hn,:])
//       tEventHapPopRate[pi,hn]  = r0 + r1 + r2 + r3                    (pyx:522-525)
//       hapPopRate[pi,hn]        = tEventHapPopRate * infectious[pi,hn] (pyx:527);  infectPopRate[pi] = their sum
//   and, at the next event drawn in pi, fastChoose(hapPopRate[pi,:], infectPopRate[pi], rn) (fast_choose.pxi:18-31):
//       the first index whose running sum reaches rn * total, and the rescaled random number.
// Per row visit that is H*(84+16S) bytes of arrays: pure HBM streaming plus a prefix sum.  The product's
// exact kernels never form these dense rows (they keep ordered occupancy lists, vgx_direct.hip / vgx_quad.hip); this file
// is the same row pass for callers that DO hold the reference's dense arrays, and the kernel the bench measures against
// the HBM roofline.  Arithmetic is FAST-mode: BirthRate factored through the row's contact sum
// K[pi] = sum_pn m[pi,pn]^2 cd[pn]/as[pn] (constant between lockdown switches) and tree-order sums — every per-haplotype
// rate is the reference's value up to that factoring (<= 1e-15 relative), sums agree to ~1e-13.
//
// Phase A (vgx_rowscan_update_kernel): one workgroup per row streams it in tiles of 1024 haplotypes, 4 consecutive
// haplotypes per thread (every access 16 or 32 contiguous bytes per lane, 1 KiB per wave instruction), writes the four rate
// arrays and reduces the row total: per-thread sums -> 64-lane DPP scan -> per-wave totals in LDS.
// Phase B (vgx_rowscan_choose_kernel): one workgroup per row scans hapPopRate with the same per-wavefront LDS-staged
// prefix sum, tile after tile with the running carry, and stops at the tile in which the sum first reaches r: on average
// half the row is read, as in the reference's linear scan.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../include/vgx.h"
#include "vgx_wave.h"

#define RS_TB 256
#define RS_PER 4
#define RS_TILE (RS_TB * RS_PER)

struct RowScanDev {
    int64_t rows, H, S;
    const int64_t *infectious;      // [rows][H]
    const double *rates123;         // [rows][H][3]
    const int64_t *numToHap;        // [H]
    const double *bRate;            // [H]
    const double *susceptibility;   // [H][S]
    const double *rowSus;           // [rows][S]
    const double *rowContact;       // [rows]
    const double *u;                // [rows]
    double *birth, *tEvent, *hapPopRate;   // [rows][H]
    double *suscept;                // [rows][H][S]
    double *rowTotal;               // [rows]
    int64_t *chosen;                // [rows]
    double *rnOut;                  // [rows]
};

// inclusive prefix over the workgroup of one value per thread (tree order); returns the prefix, `total` the group's sum
static __device__ __forceinline__ double block_scan(double v, double *wsum, double &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double pre = fscan(v);
    if (lane == 63) wsum[wave] = pre;
    __syncthreads();
    double before = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < RS_TB / 64; ++w) {
        const double s = wsum[w];
        if (w < wave) before += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return pre + before;
}

extern "C" __global__ void __launch_bounds__(RS_TB) vgx_rowscan_update_kernel(RowScanDev a) {
    const int64_t row = blockIdx.x;
    const int64_t H = a.H;
    const int S = (int)a.S;
    __shared__ double wsum[RS_TB / 64];
    const double K = a.rowContact[row];
    const int64_t *I = a.infectious + row * H;
    const double *r123 = a.rates123 + row * H * 3;
    double *birth = a.birth + row * H, *tE = a.tEvent + row * H, *hpr = a.hapPopRate + row * H;
    double *sus = a.suscept + row * H * S;
    const double *rs = a.rowSus + row * S;
    double acc = 0.0;
    for (int64_t base = 0; base < H; base += RS_TILE) {
        const int64_t h0 = base + (int64_t)threadIdx.x * RS_PER;
        double part = 0.0;
        if (h0 + RS_PER <= H && S == 1) {
            // the common shape: whole groups of four, one susceptibility group — wide loads and stores only
            const longlong2 i01 = *(const longlong2 *)(I + h0), i23 = *(const longlong2 *)(I + h0 + 2);
            const longlong2 n01 = *(const longlong2 *)(a.numToHap + h0), n23 = *(const longlong2 *)(a.numToHap + h0 + 2);
            const double2 ra = *(const double2 *)(r123 + h0 * 3), rb = *(const double2 *)(r123 + h0 * 3 + 2),
                          rc = *(const double2 *)(r123 + h0 * 3 + 4), rd = *(const double2 *)(r123 + h0 * 3 + 6),
                          re = *(const double2 *)(r123 + h0 * 3 + 8), rf = *(const double2 *)(r123 + h0 * 3 + 10);
            const int64_t hh[4] = {n01.x, n01.y, n23.x, n23.y};
            const int64_t cnt[4] = {i01.x, i01.y, i23.x, i23.y};
            const double r1[4] = {ra.x, rb.y, rd.x, re.y}, r2[4] = {ra.y, rc.x, rd.y, rf.x}, r3[4] = {rb.x, rc.y, re.x, rf.y};
            const double s0 = rs[0];
            double xb[4], bb[4], te[4], hp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xb[j] = s0 * a.susceptibility[hh[j]];            // susceptHapPopRate (pyx:386)
                bb[j] = a.bRate[hh[j]] * (xb[j] * K);            // BirthRate, factored
                te[j] = ((bb[j] + r1[j]) + r2[j]) + r3[j];       // pyx:522-525
                hp[j] = te[j] * (double)cnt[j];                  // pyx:527
                part += hp[j];
            }
            *(double2 *)(sus + h0) = make_double2(xb[0], xb[1]); *(double2 *)(sus + h0 + 2) = make_double2(xb[2], xb[3]);
            *(double2 *)(birth + h0) = make_double2(bb[0], bb[1]); *(double2 *)(birth + h0 + 2) = make_double2(bb[2], bb[3]);
            *(double2 *)(tE + h0) = make_double2(te[0], te[1]); *(double2 *)(tE + h0 + 2) = make_double2(te[2], te[3]);
            *(double2 *)(hpr + h0) = make_double2(hp[0], hp[1]); *(double2 *)(hpr + h0 + 2) = make_double2(hp[2], hp[3]);
        } else {
            for (int j = 0; j < RS_PER; ++j) {
                const int64_t h = h0 + j;
                if (h >= H) break;
                const int64_t hh = a.numToHap[h];
                double ws = 0.0;
                for (int sn = 0; sn < S; ++sn) {
                    const double x = rs[sn] * a.susceptibility[hh * S + sn];
                    sus[h * S + sn] = x;
                    ws += x;
                }
                const double b = a.bRate[hh] * (ws * K);
                const double te = ((b + r123[h * 3]) + r123[h * 3 + 1]) + r123[h * 3 + 2];
                const double hp = te * (double)I[h];
                birth[h] = b; tE[h] = te; hpr[h] = hp;
                part += hp;
            }
        }
        double tot;
        (void)block_scan(part, wsum, tot);
        acc += tot;
    }
    if (threadIdx.x == 0) a.rowTotal[row] = acc;
}

extern "C" __global__ void __launch_bounds__(RS_TB) vgx_rowscan_choose_kernel(RowScanDev a) {
    const int64_t row = blockIdx.x;
    const int64_t H = a.H;
    __shared__ double wsum[RS_TB / 64];
    __shared__ int64_t s_hit;
    __shared__ double s_pre, s_w, s_wincl[RS_TB / 64];
    __shared__ int s_first[RS_TB / 64];
    const double *hpr = a.hapPopRate + row * H;
    const double total = a.rowTotal[row];
    const double r = total * a.u[row];                  // fast_choose.pxi:22
    if (threadIdx.x == 0) s_hit = -1;
    __syncthreads();
    double carry = 0.0;
    for (int64_t base = 0; base < H; base += RS_TILE) {
        const int64_t h0 = base + (int64_t)threadIdx.x * RS_PER;
        double w[RS_PER] = {0.0, 0.0, 0.0, 0.0};
        if (h0 + RS_PER <= H) {
            const double2 a01 = *(const double2 *)(hpr + h0), a23 = *(const double2 *)(hpr + h0 + 2);
            w[0] = a01.x; w[1] = a01.y; w[2] = a23.x; w[3] = a23.y;
        } else {
            for (int j = 0; j < RS_PER; ++j) if (h0 + j < H) w[j] = hpr[h0 + j];
        }
        const double part = ((w[0] + w[1]) + w[2]) + w[3];
        double tot;
        const double incl = block_scan(part, wsum, tot) + carry;      // running sum through this thread's four entries
        // the first entry whose running sum reaches r lies in the first thread whose inclusive sum does
        const unsigned long long hitmask = __ballot(!(incl < r));
        if ((threadIdx.x & 63) == 0) s_first[threadIdx.x >> 6] = hitmask ? (int)(threadIdx.x + __ffsll((long long)hitmask) - 1) : RS_TB;
        if ((threadIdx.x & 63) == 63) s_wincl[threadIdx.x >> 6] = incl;   // for the first lane of the next wavefront
        double prev = __shfl_up(incl, 1);
        __syncthreads();
        int first = RS_TB;
#pragma unroll
        for (int w_ = RS_TB / 64 - 1; w_ >= 0; --w_) if (s_first[w_] < RS_TB) first = s_first[w_];
        if ((int)threadIdx.x == first) {
            // running sum before this thread's entries: the previous thread's inclusive sum (the same bits it compared)
            double run = (threadIdx.x & 63) ? prev : (threadIdx.x ? s_wincl[(threadIdx.x >> 6) - 1] : carry);
            int j = 0;
            for (; j < RS_PER - 1; ++j) { if (!(run + w[j] < r)) break; run += w[j]; }
            // zero weights never stop the reference's scan either (total < rn is strict, fc:26): skip forward to a positive one
            while (j < RS_PER - 1 && w[j] == 0.0) ++j;
            s_hit = h0 + j; s_pre = run + w[j]; s_w = w[j];
        }
        __syncthreads();
        if (s_hit >= 0) break;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        int64_t hit = s_hit;
        double pre = s_pre, wi = s_w;
        if (hit < 0) {   // rounding left the tree-order total below r: clamp at the end like fc:26
            hit = H - 1; pre = carry; wi = hpr[H - 1];
        }
        a.chosen[row] = hit;
        a.rnOut[row] = (r - (pre - wi)) / wi;           // fc:31
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
static std::string g_rowscan_error;
extern "C" const char *vgx_propensity_scan_error(void) { return g_rowscan_error.c_str(); }

#define RSCHECK(call)                                                                  \
    do {                                                                               \
        hipError_t err__ = (call);                                                     \
        if (err__ != hipSuccess) {                                                     \
            g_rowscan_error = std::string(#call) + ": " + hipGetErrorString(err__);    \
            for (void *q : bufs) (void)hipFree(q);                                     \
            return VGX_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

static int rowscan_run(const vgx_rowscan *io, int repeats, double *ms_update, double *ms_choose, bool synthetic) {
    std::vector<void *> bufs;
    if (!io || io->rows < 1 || io->H < 1 || io->S < 1) { g_rowscan_error = "vgx_propensity_scan: bad dimensions"; return VGX_ERR_ARG; }
    const int64_t R = io->rows, H = io->H, S = io->S;
    RowScanDev d{};
    d.rows = R; d.H = H; d.S = S;
    auto dev = [&](size_t bytes, const void *src, void **out) -> hipError_t {
        hipError_t e = hipMalloc(out, bytes ? bytes : 8);
        if (e != hipSuccess) return e;
        bufs.push_back(*out);
        if (src && bytes) return hipMemcpy(*out, src, bytes, hipMemcpyHostToDevice);
        return hipSuccess;
    };
    RSCHECK(dev((size_t)(R * H) * 8, synthetic ? nullptr : io->infectious, (void **)&d.infectious));
    RSCHECK(dev((size_t)(R * H * 3) * 8, synthetic ? nullptr : io->eventRates123, (void **)&d.rates123));
    RSCHECK(dev((size_t)H * 8, io->numToHap, (void **)&d.numToHap));
    RSCHECK(dev((size_t)H * 8, io->bRate, (void **)&d.bRate));
    RSCHECK(dev((size_t)(H * S) * 8, io->susceptibility, (void **)&d.susceptibility));
    RSCHECK(dev((size_t)(R * S) * 8, io->rowSusceptible, (void **)&d.rowSus));
    RSCHECK(dev((size_t)R * 8, io->rowContact, (void **)&d.rowContact));
    RSCHECK(dev((size_t)R * 8, io->u, (void **)&d.u));
    RSCHECK(dev((size_t)(R * H) * 8, nullptr, (void **)&d.birth));
    RSCHECK(dev((size_t)(R * H) * 8, nullptr, (void **)&d.tEvent));
    RSCHECK(dev((size_t)(R * H) * 8, nullptr, (void **)&d.hapPopRate));
    RSCHECK(dev((size_t)(R * H * S) * 8, nullptr, (void **)&d.suscept));
    RSCHECK(dev((size_t)R * 8, nullptr, (void **)&d.rowTotal));
    RSCHECK(dev((size_t)R * 8, nullptr, (void **)&d.chosen));
    RSCHECK(dev((size_t)R * 8, nullptr, (void **)&d.rnOut));
    if (synthetic) {   // every row gets the caller's first row (the measurement needs resident bytes, not distinct ones)
        for (int64_t r = 0; r < R; ++r) {
            RSCHECK(hipMemcpy((void *)(d.infectious + r * H), io->infectious, (size_t)H * 8, hipMemcpyHostToDevice));
            RSCHECK(hipMemcpy((void *)(d.rates123 + r * H * 3), io->eventRates123, (size_t)(H * 3) * 8, hipMemcpyHostToDevice));
        }
    }
    hipEvent_t e0, e1, e2;
    RSCHECK(hipEventCreate(&e0)); RSCHECK(hipEventCreate(&e1)); RSCHECK(hipEventCreate(&e2));
    float tu = 0.f, tc = 0.f;
    for (int it = 0; it < repeats + 1; ++it) {   // the first pass warms up
        RSCHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(vgx_rowscan_update_kernel, dim3((unsigned)R), dim3(RS_TB), 0, 0, d);
        RSCHECK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(vgx_rowscan_choose_kernel, dim3((unsigned)R), dim3(RS_TB), 0, 0, d);
        RSCHECK(hipEventRecord(e2, 0));
        RSCHECK(hipEventSynchronize(e2));
        if (it > 0 || repeats == 0) {
            float a_ = 0.f, b_ = 0.f;
            RSCHECK(hipEventElapsedTime(&a_, e0, e1));
            RSCHECK(hipEventElapsedTime(&b_, e1, e2));
            tu += a_; tc += b_;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    const int n = repeats > 0 ? repeats : 1;
    if (ms_update) *ms_update = tu / n;
    if (ms_choose) *ms_choose = tc / n;
    const int64_t Rout = synthetic ? 1 : R;   // synthetic runs return the first row only
    if (io->birthRate) RSCHECK(hipMemcpy(io->birthRate, d.birth, (size_t)(Rout * H) * 8, hipMemcpyDeviceToHost));
    if (io->tEvent) RSCHECK(hipMemcpy(io->tEvent, d.tEvent, (size_t)(Rout * H) * 8, hipMemcpyDeviceToHost));
    if (io->hapPopRate) RSCHECK(hipMemcpy(io->hapPopRate, d.hapPopRate, (size_t)(Rout * H) * 8, hipMemcpyDeviceToHost));
    if (io->susceptHapPopRate) RSCHECK(hipMemcpy(io->susceptHapPopRate, d.suscept, (size_t)(Rout * H * S) * 8, hipMemcpyDeviceToHost));
    if (io->rowTotal) RSCHECK(hipMemcpy(io->rowTotal, d.rowTotal, (size_t)Rout * 8, hipMemcpyDeviceToHost));
    if (io->chosen) RSCHECK(hipMemcpy(io->chosen, d.chosen, (size_t)Rout * 8, hipMemcpyDeviceToHost));
    if (io->rnOut) RSCHECK(hipMemcpy(io->rnOut, d.rnOut, (size_t)Rout * 8, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return VGX_OK;
}

extern "C" int vgx_propensity_scan(const vgx_rowscan *io) { return rowscan_run(io, 0, nullptr, nullptr, false); }

extern "C" int vgx_propensity_scan_bench(const vgx_rowscan *first_row, int64_t rows, int repeats, double *ms_update,
                                         double *ms_choose) {
    if (!first_row) return VGX_ERR_ARG;
    vgx_rowscan io = *first_row;
    io.rows = rows;
    return rowscan_run(&io, repeats < 1 ? 1 : repeats, ms_update, ms_choose, true);
}
