// Development aid: exactness and cost of a lane-ordered f64 prefix scan built from v_fmac_f64 (DPP row_newbcast) with
// EXEC narrowed before every step (s_lshl_b64 exec + the 5 wait states a DPP instruction needs after an EXEC write),
// against the serial sum it has to reproduce bit for bit and against the readlane-based scan of vgx_wave.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vgsim_amd/csrc tools/scan_dpp_test.hip -o tools/scan_dpp_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vgx_wave.h"

#define FX(K, LANE, RM) "s_lshl_b64 exec, -1, " #LANE "\n\ts_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:" #RM " bank_mask:0xf\n\t"
#define FX16(B, RM) FX(0, B+0, RM) FX(1, B+1, RM) FX(2, B+2, RM) FX(3, B+3, RM) FX(4, B+4, RM) FX(5, B+5, RM) FX(6, B+6, RM) FX(7, B+7, RM) \
                    FX(8, B+8, RM) FX(9, B+9, RM) FX(10, B+10, RM) FX(11, B+11, RM) FX(12, B+12, RM) FX(13, B+13, RM) FX(14, B+14, RM) FX(15, B+15, RM)

static __device__ __forceinline__ double scan_dpp(double v, double carry) {
    double acc = carry;
    const double one = 1.0;
    asm volatile("s_nop 1\n\t" FX16(0, 0x1) "s_mov_b64 exec, -1\n\t" : "+v"(acc) : "v"(v), "v"(one) : "scc");
    {   // rows 1..3 start from the total of the previous row (lane 15 of that row)
        int lo = __double2loint(acc), hi = __double2hiint(acc);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x2, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x2, 0xf, false);
        acc = __hiloint2double(hi, lo);
    }
    asm volatile("s_nop 1\n\t" FX16(16, 0x2) "s_mov_b64 exec, -1\n\t" : "+v"(acc) : "v"(v), "v"(one) : "scc");
    {
        int lo = __double2loint(acc), hi = __double2hiint(acc);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x4, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x4, 0xf, false);
        acc = __hiloint2double(hi, lo);
    }
    asm volatile("s_nop 1\n\t" FX16(32, 0x4) "s_mov_b64 exec, -1\n\t" : "+v"(acc) : "v"(v), "v"(one) : "scc");
    {
        int lo = __double2loint(acc), hi = __double2hiint(acc);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0x8, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0x8, 0xf, false);
        acc = __hiloint2double(hi, lo);
    }
    asm volatile("s_nop 1\n\t" FX16(48, 0x8) "s_mov_b64 exec, -1\n\t" : "+v"(acc) : "v"(v), "v"(one) : "scc");
    return acc;
}

template <int MODE>
__global__ void __launch_bounds__(64) kern(const double *in, double *out, unsigned long long *cyc, int iters) {
    double v = in[blockIdx.x * 64 + threadIdx.x], r = 0.0, carry = 0.125;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        r = MODE == 0 ? scan_dpp(v, carry) : seq_scan(v, 64, carry);
        carry = bcast(r, 63) * 1e-3;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256 * 4 * 4, iters = 200;
    std::vector<double> h(blocks * 64), o0(blocks * 64), o1(blocks * 64);
    srand(7);
    for (auto &x : h) x = (rand() % 5 == 0) ? 0.0 : (double)rand() / RAND_MAX * ((rand() & 1) ? 1e-3 : 1e3);
    double *din, *dout; unsigned long long *dc;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&dc, blocks * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) kern<0><<<blocks, 64>>>(din, dout, dc, iters); else kern<1><<<blocks, 64>>>(din, dout, dc, iters);
            hipEventRecord(e1); hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(mode == 0 ? o0.data() : o1.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.3f ns per step and SIMD at 4 waves/SIMD\n", mode, mode == 0 ? "fmac_dpp + exec narrowing" : "readlane scan", (double)ms * 1e6 / ((double)iters * 64 * 4));
    }
    // single-iteration exactness vs the serial sum on the host
    kern<0><<<blocks, 64>>>(din, dout, dc, 1); hipMemcpy(o0.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    kern<1><<<blocks, 64>>>(din, dout, dc, 1); hipMemcpy(o1.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    long bad0 = 0, bad1 = 0;
    for (int b = 0; b < blocks; ++b) {
        volatile double acc = 0.125;
        for (int l = 0; l < 64; ++l) {
            acc = acc + h[b * 64 + l];
            double a = acc;
            if (memcmp(&a, &o0[b * 64 + l], 8)) bad0++;
            if (memcmp(&a, &o1[b * 64 + l], 8)) bad1++;
        }
    }
    printf("mismatches vs serial sum: fmac_dpp scan %ld, readlane scan %ld (of %d)\n", bad0, bad1, blocks * 64);
    return 0;
}
