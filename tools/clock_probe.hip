// Shader clock WHILE the fp32 GEMM kernels run: a one-wave probe kernel on a second stream reads s_memtime (shader cycles) and
// s_memrealtime (100 MHz) over a fixed wall interval while libmtts.so's GEMM of a model shape loops on the first stream.
// hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -Iinclude -Lmeta_tts_amd -lmtts -Wl,-rpath,'$ORIGIN/../meta_tts_amd' -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mtts.h"

__global__ void probe(unsigned long long wall_ticks, unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(32); w1 = wall_clock64(); }
    out[0] = clock64() - c0; out[1] = w1 - w0;
}

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 17047, N = argc > 3 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 2304;
    const int tile = argc > 4 ? atoi(argv[4]) : 3064, reps = 60;
    float *A, *B, *C; unsigned long long* out;
    hipMalloc((void**)&A, (size_t)M * K * 4); hipMalloc((void**)&B, (size_t)N * K * 4); hipMalloc((void**)&C, (size_t)M * N * 4);
    hipMalloc((void**)&out, 16);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: probe alone (idle chip), pass 1: probe beside the GEMM loop
        if (pass) {
            for (int i = 0; i < 10; ++i) mtts_gemm_f32(0, M, N, K, A, K, B, K, C, N, nullptr, 1.f, 0, tile, sa);
            hipStreamSynchronize(sa);
            hipEventRecord(e0, sa);
            for (int i = 0; i < reps; ++i) mtts_gemm_f32(0, M, N, K, A, K, B, K, C, N, nullptr, 1.f, 0, tile, sa);
            hipEventRecord(e1, sa);
        }
        probe<<<1, 64, 0, sb>>>(pass ? 3000000ull : 1000000ull, out);   // 30 ms / 10 ms of wall time
        hipDeviceSynchronize();
        unsigned long long r[2]; hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
        const double mhz = (double)r[0] / ((double)r[1] / 100.0);
        if (!pass) printf("idle chip: shader clock %.0f MHz\n", mhz);
        else {
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tf = 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12;
            printf("NT %d x %d x %d, tile code %d: %.1f us per launch, %.1f TFLOP/s = %.3f of 157.3; shader clock beside it %.0f MHz -> %.3f of the %.1f TFLOP/s the matrix pipes peak at that clock\n",
                   M, N, K, tile, 1e3 * ms / reps, tf, tf / 157.3, mhz, tf / (157.3 * mhz / 2400.0), 157.3 * mhz / 2400.0);
        }
    }
    return 0;
}
