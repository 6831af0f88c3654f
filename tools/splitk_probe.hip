// Kernel-level check of the batch regime's split-K (gemm.h: gemm_batch_end + slab_combine) on the shape the 8-task step splits — the variance predictors'
// k = 3 convolutions on the phoneme rectangle (NT, M = 424 rows per task, N = 256, K = 768 over overlapping rows lda = 256, 8 groups) — single-source and
// DUAL-source (the tangent forward of second-order MAML), against a double-precision host reference.  MTTS_DBG_NOSPLIT=1 switches the split off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../meta_tts_amd/csrc/gemm.h"
#include "../meta_tts_amd/csrc/gemm_glds.h"
#include "../meta_tts_amd/csrc/gemm_bf16.h"
using namespace mtts;
static float frand() { return (float)rand() / RAND_MAX - 0.5f; }
int main(int argc, char** argv) {
    const int G = 8, M = 424, N = 256, CIN = 256, KT = 3, K = KT * CIN, PAD = 1;
    const int rows_alloc = M + 2 * PAD + 8;
    const long long a_gs = (long long)rows_alloc * CIN, b_gs = (long long)N * K, c_gs = (long long)M * N;
    std::vector<float> hA(G * a_gs), hB(G * b_gs), hA2(G * a_gs), hB2(G * b_gs), hbias(N), hC(G * c_gs);
    for (auto& v : hA) v = frand(); for (auto& v : hB) v = frand() * 0.1f; for (auto& v : hA2) v = frand(); for (auto& v : hB2) v = frand() * 0.1f; for (auto& v : hbias) v = frand();
    float *dA, *dB, *dA2, *dB2, *dbias, *dC;
    hipMalloc((void**)&dA, hA.size() * 4); hipMalloc((void**)&dB, hB.size() * 4); hipMalloc((void**)&dA2, hA.size() * 4); hipMalloc((void**)&dB2, hB.size() * 4);
    hipMalloc((void**)&dbias, N * 4); hipMalloc((void**)&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dA2, hA2.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB2, hB2.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice);
    GemmCtx cx;
    if (cx.alloc_workspace()) { printf("workspace alloc failed\n"); return 1; }
    for (int dual = 0; dual < 2; ++dual)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(dC, 0xff, hC.size() * 4);
            GemmArgs g;
            g.A = dA + (long long)(4 - PAD) * CIN; g.a_gs = a_gs; g.lda = CIN;   // row m of the im2col matrix = x[m - 1 .. m + 1] (rows 4 .. of the slab: 4 guard rows)
            g.B = dB; g.b_gs = b_gs; g.ldb = K;
            if (dual) { g.A2 = dA2 + (long long)(4 - PAD) * CIN; g.a2_gs = a_gs; g.B2 = dB2; g.b2_gs = b_gs; }
            g.C = dC; g.c_gs = c_gs; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = dbias; g.flags = 0;
            gemm_launch(cx, GEMM_NT, g, M, N, G, nullptr);
            hipDeviceSynchronize();
            if (cx.error) { printf("launcher error: %s\n", cx.error); return 1; }
            hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, scale = 0;
            for (int z = 0; z < G; ++z)
                for (int m = 0; m < M; m += 7)
                    for (int n = 0; n < N; n += 5) {
                        double s = hbias[n];
                        const float* a = hA.data() + z * a_gs + (long long)(4 + m - PAD) * CIN;   // contiguous span of K floats
                        const float* b = hB.data() + z * b_gs + (long long)n * K;
                        for (int k = 0; k < K; ++k) s += (double)a[k] * b[k];
                        if (dual) {
                            const float* a2 = hA2.data() + z * a_gs + (long long)(4 + m - PAD) * CIN; const float* b2 = hB2.data() + z * b_gs + (long long)n * K;
                            for (int k = 0; k < K; ++k) s += (double)a2[k] * b2[k];
                        }
                        const double e = fabs(s - hC[z * c_gs + (long long)m * N + n]);
                        if (e > worst) worst = e;
                        if (fabs(s) > scale) scale = fabs(s);
                    }
            printf("dual %d rep %d: max |err| %.3e of max |ref| %.3e (kind %d)\n", dual, rep, worst, scale, cx.last_kind);
        }
    return 0;
}
