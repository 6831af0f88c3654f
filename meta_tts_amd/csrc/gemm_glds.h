// fp32-MFMA GEMM with LDS-DMA staging (gfx950 `global_load_lds_dwordx4`): the same three operand forms, grouped
// launch modes, split-K and fused epilogue as gemm.h's register-staged kernel, but the operand tiles go HBM/L2 -> LDS
// directly — no staging VGPRs, no ds_write pass (13 issue cycles per ds_write_b128 and a VGPR->LDS transfer path that
// loads do not hide), no per-slice wait on a just-issued load.
//
//  * block tile 64x64, 4 waves (2x2, 32x32 each, two accumulator chains per wave), BK = 32: a K-contiguous operand row
//    is one full 128-byte line per slice;
//  * 3-stage LDS ring (3 x 16 KB); slice c+2 is issued right after the barrier that retires slice c-1's reads, so two
//    slices (8 DMA instructions per wave) are in flight while slice c feeds the MFMAs; the wait for slice c is a
//    counted `s_waitcnt vmcnt(4)` followed by a raw `s_barrier`, and the fragments are read after that barrier
//    (the only ordering that makes LDS-DMA data visible to other waves);
//  * the DMA writes LDS linearly (wave-uniform base + lane * 16 B), so a K-contiguous tile is stored [64 rows][32 k]
//    unpadded with the 16-byte k-quads of row r XOR-permuted by (r >> 1) & 7 — applied to the SOURCE address on the way
//    in and to the fragment address on the way out (same involution): every 16-lane group of a ds_read_b128 then covers
//    all 64 banks.  Reduction-major tiles ([32 k][64 cols]) are stored as they are (ds_read_b32, conflict-free);
//  * rows / columns beyond the operand are clamped to the last valid one (their products only reach output elements the
//    epilogue never stores); a partial last K-slice cannot be clamped (both operands would be garbage), it is staged
//    through registers with zero fill;
//  * the DMA and its waits are inline asm: hipcc neither counts them nor drains them at barriers.
// GPU only: the SIMT emulator build keeps using gemm.h's kernel.
#pragma once
#include "gemm.h"

#if !defined(MTTS_EMU)
namespace mtts {

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

constexpr int kGldsBK = 32, kGldsStages = 3;
constexpr int kGldsTile = 64 * kGldsBK;         // floats per operand tile
constexpr int kGldsStage = 2 * kGldsTile;       // floats per ring stage (A + B)
constexpr int kGldsSmemFloats = kGldsStages * kGldsStage;

// K-loop of one 64x64 output tile over the BK=32 K-chunks [c_lo, c_hi) (see gemm.h: gemm_f32_kloop); the products are ADDED into acc.
template <int FORM>
__device__ __forceinline__ void gemm_glds_kloop(const GemmArgs& g, const GemmProb& pr, int z, int m0, int n0, bool cs_tile, int c_lo, int c_hi,
                                                float* smem, f32x16 (&acc)[1][1]) {
    constexpr int BK = kGldsBK, R = kGldsStages;
    constexpr bool A_KC = (FORM != GEMM_TN);
    constexpr bool B_KC = (FORM == GEMM_NT);
    const float* A = pr.A;
    const float* B = pr.B;
    const int M = pr.M, N = pr.N, K = pr.K;
    const int lda = pr.lda;
    int ldb = pr.ldb;
    (void)R;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int l31 = lane & 31, h = lane >> 5;
    const int M4 = (M + 3) & ~3, N4 = (N + 3) & ~3;
    int n0b = n0, N4b = N4;   // column window of the B operand
    if (cs_tile) { B = g.colsum_w + (long long)z * g.colsum_w_gs; ldb = 4; n0b = 0; N4b = 4; }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;

    // ---- per-lane source offsets of this wave's two DMA pieces per operand (k0 term added per slice) ----
    long long a_off[2], b_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * wave + q;  // piece index 0..7 (1 KB each)
        if (A_KC) {
            const int r = 8 * i + (lane >> 3), kq = (lane & 7) ^ ((r >> 1) & 7);
            const int gm = (m0 + r < M) ? m0 + r : M - 1;
            a_off[q] = (long long)gm * lda + 4 * kq;
        } else {
            const int krow = 4 * i + (lane >> 4);
            int col = m0 + (lane & 15) * 4;
            if (col > M4 - 4) col = M4 - 4 > 0 ? M4 - 4 : 0;
            a_off[q] = (long long)krow * lda + col;
        }
        if (B_KC) {
            const int r = 8 * i + (lane >> 3), kq = (lane & 7) ^ ((r >> 1) & 7);
            const int gn = (n0 + r < N) ? n0 + r : N - 1;
            b_off[q] = (long long)gn * ldb + 4 * kq;
        } else {
            const int krow = 4 * i + (lane >> 4);
            int col = n0b + (lane & 15) * 4;
            if (col > N4b - 4) col = N4b - 4 > 0 ? N4b - 4 : 0;
            b_off[q] = (long long)krow * ldb + col;
        }
    }

    const int nchunks = c_hi > c_lo ? c_hi - c_lo : 0;
    if (nchunks == 0) return;
    const int kb0 = c_lo * BK;

    auto is_full = [&](int c) { return kb0 + (c + 1) * BK <= K; };
    // running tap state instead of a runtime integer division per slice (gemm.h: a_tap_of / b_tap_of)
    int a_tap_i = 0, a_tap_base = 0, b_tap_i = 0, b_tap_base = 0;
    // (the tap geometry in registers: read through `g` inside the K-loop it is re-loaded from the kernel-argument segment every slice, and
    // the s_waitcnt lgkmcnt(0) behind each scalar load also drains the LDS fragment reads in flight)
    const int g_a_tap_k = g.a_tap_k, g_tap_k = g.tap_k, g_taps = g.taps, g_tap_bstride = g.tap_bstride, g_a_tap_rows = g.a_tap_rows;
    auto a_tap_of = [&](int k0) { while (k0 - a_tap_base >= g_a_tap_k) { a_tap_base += g_a_tap_k; ++a_tap_i; } return a_tap_i; };
    auto b_tap_of = [&](int k0) { while (k0 - b_tap_base >= g_tap_k) { b_tap_base += g_tap_k; ++b_tap_i; } return b_tap_i; };
    // stage slice c into ring slot st: 4 DMA instructions per wave, or (partial last slice) zero-filled register staging
    auto stage = [&](int c, int st) {
        const int k0 = kb0 + c * BK;
        float* As = smem + st * kGldsStage;
        float* Bs = As + kGldsTile;
        if (is_full(c)) {
            const unsigned sa = lds_base + (unsigned)(st * kGldsStage) * 4u, sb = sa + (unsigned)kGldsTile * 4u;
            const int atap = A_KC ? a_tap_of(k0) : 0;   // dilated taps of a K-contiguous A operand (gemm.h)
            const float* Ab = A_KC ? A + (long long)atap * g_a_tap_rows * lda + (k0 - a_tap_base) : A + (long long)k0 * lda;
            const float* Bb;
            if (B_KC) Bb = B + k0;
            else {
                const int tap = b_tap_of(k0), kin = k0 - b_tap_base;
                Bb = B + (long long)(g_taps - 1 - tap) * g_tap_bstride + (long long)kin * ldb;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                glds16(Ab + a_off[q], (unsigned)__builtin_amdgcn_readfirstlane((int)(sa + (unsigned)(2 * wave + q) * 1024u)));
                glds16(Bb + b_off[q], (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)(2 * wave + q) * 1024u)));
            }
        } else {
            // 64 x 32 floats per operand = 512 float4, 2 per thread, same LDS images as the DMA path
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = tid + 256 * q;
                {
                    float4 v = zero4();
                    if (A_KC) {
                        const int r = idx >> 3, pos = idx & 7, kq = pos ^ ((r >> 1) & 7), gk = k0 + 4 * kq;
                        if (m0 + r < M) {
                            const int atap = a_tap_of(k0);
                            const float* p = A + ((long long)(m0 + r) + (long long)atap * g_a_tap_rows) * lda + (gk - a_tap_base);
                            if (gk + 3 < K) v = ld4(p);
                            else { if (gk < K) v.x = p[0]; if (gk + 1 < K) v.y = p[1]; if (gk + 2 < K) v.z = p[2]; }
                        }
                        st4(As + r * 32 + pos * 4, v);
                    } else {
                        const int kk = idx >> 4, c4 = (idx & 15) * 4;
                        if (k0 + kk < K && m0 + c4 < M4) v = ld4(A + (long long)(k0 + kk) * lda + m0 + c4);
                        st4(As + kk * 64 + c4, v);
                    }
                }
                {
                    float4 v = zero4();
                    if (B_KC) {
                        const int r = idx >> 3, pos = idx & 7, kq = pos ^ ((r >> 1) & 7), gk = k0 + 4 * kq;
                        if (n0 + r < N) {
                            const float* p = B + (long long)(n0 + r) * ldb + gk;
                            if (gk + 3 < K) v = ld4(p);
                            else { if (gk < K) v.x = p[0]; if (gk + 1 < K) v.y = p[1]; if (gk + 2 < K) v.z = p[2]; }
                        }
                        st4(Bs + r * 32 + pos * 4, v);
                    } else {
                        const int kk = idx >> 4, c4 = (idx & 15) * 4;
                        const int tap = b_tap_of(k0), kin = k0 - b_tap_base;
                        const float* Bc = B + (long long)(g_taps - 1 - tap) * g_tap_bstride;
                        if (k0 + kk < K && n0b + c4 < N4b) v = ld4(Bc + (long long)(kin + kk) * ldb + n0b + c4);
                        st4(Bs + kk * 64 + c4, v);
                    }
                }
            }
        }
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    if (nchunks > 0) stage(0, 0);
    if (nchunks > 1) stage(1, 1);
    int st = 0;  // ring slot of slice c
    for (int c = 0; c < nchunks; ++c) {
        // slice c landed: only slice c+1's four DMA instructions may still be in flight (none if it was register-staged)
        if (c + 1 < nchunks && is_full(c + 1)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 2 < nchunks) stage(c + 2, st == 0 ? 2 : st - 1);  // (c + 2) % 3 == (st + 2) % 3
        const float* As = smem + st * kGldsStage;
        const float* Bs = As + kGldsTile;
        float fa[BK / 8][4], fb[BK / 8][4];
#pragma unroll
        for (int j2 = 0; j2 < BK / 8; ++j2) {
            if (A_KC) {
                const int row = wm0 + l31;
                const float4 v = ld4(As + row * 32 + (((2 * j2 + h) ^ ((row >> 1) & 7)) << 2));
                fa[j2][0] = v.x; fa[j2][1] = v.y; fa[j2][2] = v.z; fa[j2][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[j2][e] = As[(8 * j2 + 4 * h + e) * 64 + wm0 + l31];
            }
            if (B_KC) {
                const int row = wn0 + l31;
                const float4 v = ld4(Bs + row * 32 + (((2 * j2 + h) ^ ((row >> 1) & 7)) << 2));
                fb[j2][0] = v.x; fb[j2][1] = v.y; fb[j2][2] = v.z; fb[j2][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[j2][e] = Bs[(8 * j2 + 4 * h + e) * 64 + wn0 + l31];
            }
        }
#pragma unroll
        for (int j2 = 0; j2 < BK / 8; ++j2)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j2][e], fb[j2][e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j2][e + 1], fb[j2][e + 1], acc1, 0, 0, 0);
            }
        st = (st == 2) ? 0 : st + 1;
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc0[r] + acc1[r];
}

template <int FORM, bool DUAL = false>
__device__ __forceinline__ void gemm_glds_body(const GemmArgs& g, int z, int bxs, float* smem) {
    constexpr int BM = 64, BN = 64, BK = kGldsBK;
    const GemmProb pr = gemm_resolve(g, z);
    const bool has_cs = gemm_has_colsum<FORM>(g);
    const int tiles_nc = (pr.N + BN - 1) / BN, tiles_n = tiles_nc + (has_cs ? 1 : 0), tiles_m = (pr.M + BM - 1) / BM;
    const int S = g.splitk > 1 ? g.splitk : 1;
    const int tile_lin = bxs / S, split = bxs - tile_lin * S;
    if (tile_lin >= tiles_m * tiles_n || pr.K <= 0) return;
    const int m0 = (tile_lin / tiles_n) * BM, n0 = (tile_lin % tiles_n) * BN;
    const bool cs_tile = has_cs && n0 == tiles_nc * BN;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int nch_all = (pr.K + BK - 1) / BK, cps = (nch_all + S - 1) / S;
    const int c_lo = split * cps, c_hi = (c_lo + cps < nch_all) ? c_lo + cps : nch_all;
    gemm_glds_kloop<FORM>(g, pr, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
    if (DUAL) {   // second source of a dual-source problem (gemm.h: gemm_f32_body)
        if (g.A2 != nullptr && !cs_tile) {
            __syncthreads();   // every wave is done with the ring slots of the first source
            const GemmProb p2 = gemm_resolve2(g, z, pr);
            gemm_glds_kloop<FORM>(g, p2, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
        }
    }
    if (S > 1 && !splitk_combine<1, 1, 256>(g, z, tile_lin, split, S, acc)) return;
    gemm_finish<1, 1, 2, 2>(g, pr, z, m0, n0, cs_tile, acc);
}

template <int FORM>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[kGldsSmemFloats];
    int bx = blockIdx.x, z = blockIdx.z;
    if (g.xs.on) { if (!xcd_sched_locate(g.xs, bx, z, bx)) return; }
    else if (g.swizzle == 2) { if (!xcd_panel_locate(bx, g.po_tiles_m, (g.N + 63) / 64 + (gemm_has_colsum<FORM>(g) ? 1 : 0), g.splitk > 1 ? g.splitk : 1, bx)) return; }
    else if (g.swizzle) bx = xcd_group_remap(bx, (int)gridDim.x, xcd_group_size((g.N + 63) / 64, g.splitk));
    gemm_glds_body<FORM>(g, z, bx, smem);
}

__global__ __launch_bounds__(256) void gemm_glds_multi_kernel(GemmMulti mp) {
    __shared__ __attribute__((aligned(16))) float smem[kGldsSmemFloats];
    int p, z, bx;
    if (!gemm_multi_locate(mp, p, z, bx)) return;
    const int form = mp.form[p];
    if (form == GEMM_NT) gemm_glds_body<GEMM_NT>(mp.g[p], z, bx, smem);
    else if (form == GEMM_NN) gemm_glds_body<GEMM_NN>(mp.g[p], z, bx, smem);
    else gemm_glds_body<GEMM_TN>(mp.g[p], z, bx, smem);
}

__global__ __launch_bounds__(256) void gemm_glds_multi_dual_kernel(GemmMulti mp) {
    __shared__ __attribute__((aligned(16))) float smem[kGldsSmemFloats];
    int p, z, bx;
    if (!gemm_multi_locate(mp, p, z, bx)) return;
    const int form = mp.form[p];
    if (form == GEMM_NT) gemm_glds_body<GEMM_NT, true>(mp.g[p], z, bx, smem);
    else if (form == GEMM_NN) gemm_glds_body<GEMM_NN, true>(mp.g[p], z, bx, smem);
    else gemm_glds_body<GEMM_TN, true>(mp.g[p], z, bx, smem);
}

inline void gemm_glds_launch(int form, const GemmArgs& g, dim3 grid, hipStream_t stream) {
    if (form == GEMM_NT) hipLaunchKernelGGL((gemm_glds_kernel<GEMM_NT>), grid, dim3(256), 0, stream, g);
    else if (form == GEMM_NN) hipLaunchKernelGGL((gemm_glds_kernel<GEMM_NN>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_glds_kernel<GEMM_TN>), grid, dim3(256), 0, stream, g);
}
inline void gemm_glds_multi_launch(const GemmMulti& mp, dim3 grid, hipStream_t stream, bool dual) {
    if (dual) hipLaunchKernelGGL(gemm_glds_multi_dual_kernel, grid, dim3(256), 0, stream, mp);
    else hipLaunchKernelGGL(gemm_glds_multi_kernel, grid, dim3(256), 0, stream, mp);
}

}  // namespace mtts
#endif  // !MTTS_EMU
