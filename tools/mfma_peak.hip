// Sustained fp32-MFMA rate and shader clock of THIS box: a register-only v_mfma_f32_32x32x2_f32 loop (no memory traffic) on every CU,
// timed with HIP events; the shader clock comes from s_memtime (clock64, shader cycles) against s_memrealtime (wall_clock64, 100 MHz).
// What the GEMM kernels' roofline fraction (quoted against the data-sheet 157.3 TFLOP/s = 256 CUs x 256 flop/clk x 2.4 GHz) can reach
// when the part holds less than 2.4 GHz under a matrix load.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out, unsigned long long* clk) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int CHAINS>
static void run(int wgs_per_cu, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc((void**)&out, 64); hipMalloc((void**)&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    mfma_loop<CHAINS><<<grid, 256>>>(iters / 10, out, clk);   // warm-up (clock ramp)
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<CHAINS><<<grid, 256>>>(iters, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * CHAINS * (2.0 * 32 * 32 * 2);
        printf("chains %d, %d workgroups per CU, %d iterations: %.2f ms, %.1f TFLOP/s, shader clock %.0f MHz (clock64 / wall_clock64 at 100 MHz)\n",
               CHAINS, wgs_per_cu, iters, ms, flops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100.0));
    }
    hipFree(out); hipFree(clk);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    run<1>(1, iters);      // ONE dependent chain per wave (what a 32 x 32 output tile per wave is: attention.h phases A / C), 1 / 2 / 4 waves per SIMD
    run<1>(2, iters / 2);
    run<1>(4, iters / 4);
    run<2>(1, iters);      // one wave per SIMD, two dependent chains: 2 x 64-cycle MFMAs back to back (issue-bound check)
    run<4>(1, iters);
    run<4>(2, iters / 2);  // two waves per SIMD
    return 0;
}
