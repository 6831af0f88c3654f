// Launcher of the split-bf16 GEMM (gemm_bf16.h).  Returns true when the 128x128 tile was used.
#pragma once
#include "gemm_bf16.h"

namespace mtts {

inline bool gemm_launch_bf16x3(int form, const GemmArgs& g, int max_M, int max_N, int groups, hipStream_t stream, int tile,
                               double rows) {
    auto ntiles = [&](int t) { return (long)((max_M + t - 1) / t) * ((max_N + t - 1) / t); };
    if (tile != 64 && tile != 128) {
        // same wave-quantisation model as the fp32 launcher; the 64x64 tile does 6 MFMAs per barrier here, so its
        // per-tile efficiency is lower
        auto eff = [&](int t, double base) {
            const double b = std::ceil(rows / t) * ((max_N + t - 1) / t) / 256.0;
            return base * b / std::ceil(b);
        };
        tile = eff(128, 1.0) >= eff(64, gemm_bf16_small_tile_eff()) ? 128 : 64;
    }
    dim3 block(256), grid((unsigned)ntiles(tile), 1, (unsigned)groups);
#define MTTS_GEMM16_CASE(F, T) \
    if (form == F && tile == T) { MTTS_LAUNCH((gemm_bf16x3_kernel<F, T, T, 3>), grid, block, stream, g); }
    MTTS_GEMM16_CASE(GEMM_NT, 128) MTTS_GEMM16_CASE(GEMM_NT, 64)
    MTTS_GEMM16_CASE(GEMM_NN, 128) MTTS_GEMM16_CASE(GEMM_NN, 64)
    MTTS_GEMM16_CASE(GEMM_TN, 128) MTTS_GEMM16_CASE(GEMM_TN, 64)
#undef MTTS_GEMM16_CASE
    return tile == 128;
}

}  // namespace mtts
