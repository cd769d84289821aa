// Development aid: cost of one step of a serial f64 summation chain on gfx950, for the three ways the direct
// kernel could feed it: (a) v_fmac_f64 with a DPP row_newbcast source (1 VALU/step), (b) 2x v_readlane + v_add_f64
// with an SGPR operand (3 VALU/step), (c) plain dependent v_add_f64 from a VGPR (lower bound).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_chain.hip -o gpurun_out/microbench_chain ; run on the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FM(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define FM16 FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
#define RL(K) "v_readlane_b32 s10, %1, " #K "\n\tv_readlane_b32 s11, %2, " #K "\n\ts_nop 1\n\tv_add_f64 %0, %0, s[10:11]\n\t"
#define RL16 RL(0) RL(1) RL(2) RL(3) RL(4) RL(5) RL(6) RL(7) RL(8) RL(9) RL(10) RL(11) RL(12) RL(13) RL(14) RL(15)
#define AD "v_add_f64 %0, %0, %1\n\t"
#define AD16 AD AD AD AD AD AD AD AD AD AD AD AD AD AD AD AD

template <int MODE>
__global__ void __launch_bounds__(64) chain(double *out, unsigned long long *cyc, int iters) {
    double w = 1.0 + threadIdx.x * 1e-9, acc = 0.0, one = 1.0;
    int lo = __double2loint(w), hi = __double2hiint(w);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile("s_nop 1\n\t" FM16 FM16 FM16 FM16 : "+v"(acc) : "v"(w), "v"(one));
        if (MODE == 1) asm volatile(RL16 RL16 RL16 RL16 : "+v"(acc) : "v"(lo), "v"(hi) : "s10", "s11");
        if (MODE == 2) asm volatile(AD16 AD16 AD16 AD16 : "+v"(acc) : "v"(w));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 2000;
    double *out;
    unsigned long long *cyc;
    const char *names[3] = {"v_fmac_f64_dpp row_newbcast", "2x v_readlane + v_add_f64 sgpr", "v_add_f64 vgpr (dependent)"};
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        int blocks = 256 * 4 * waves_per_simd;
        hipMalloc(&out, blocks * 64 * 8);
        hipMalloc(&cyc, blocks * 8);
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) chain<0><<<blocks, 64>>>(out, cyc, iters);
                if (mode == 1) chain<1><<<blocks, 64>>>(out, cyc, iters);
                if (mode == 2) chain<2><<<blocks, 64>>>(out, cyc, iters);
                hipEventRecord(e1);
                hipDeviceSynchronize();
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= blocks;
            double steps = (double)iters * 64;
            printf("waves/SIMD %d  %-34s  %.2f cycles/step per wave (counter), %.3f ns/step/SIMD (wall)\n", waves_per_simd,
                   names[mode], mean / steps, (double)ms * 1e6 / (steps * waves_per_simd));
        }
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
