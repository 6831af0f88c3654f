// bf16 operand family of the grouped GEMM (the optional reduced-precision numerics mode, mtts_set_numerics(h, 1); BASELINE.json
// configs[1] "bf16 on 1xMI355X").  Same problems, operand forms, grouped / multi-problem / dual-source launches, split-K rendezvous
// and fused epilogue as gemm.h — only the K-loop differs:
//
//  * the matrix instruction is v_mfma_f32_32x32x16_bf16 (dense peak ~2.5 PFLOP/s, 16x the fp32-input MFMA): operands rounded to
//    bf16 (round-to-nearest-even), products exact, fp32 accumulation.  Master weights, optimizer state, LayerNorm / BatchNorm /
//    softmax statistics, losses and every tensor in HBM stay fp32 — this mode changes the arithmetic of the contractions only;
//  * operands are read from HBM / L2 as fp32 and rounded ONCE, in the global -> LDS staging pass (one v_cvt_pk_bf16_f32 per two
//    elements, hidden behind the previous slice's MFMAs); the K-loop itself is ds_read_b128 + MFMA with no VALU work per product
//    (round 2's "bf16x3" split every operand into three planes inside the K-loop and ran at 1.13x the fp32 step — removed);
//  * both LDS tiles are K-contiguous [rows][BK + 8] bf16 whatever the source layout: a K-contiguous fp32 operand (NT: A and B, NN: A)
//    is staged float4 -> 4 bf16 (ds_write_b64); a reduction-major one (NN: B = the [Cout][k][Cin] weight image walked by taps, TN:
//    both = row-major activations reduced over rows) is staged as 2(k) x 4(cols) register blocks whose column pairs (k, k+1) are
//    exactly what v_cvt_pk packs — the transpose costs nothing beyond the conversion;
//  * fragments: lane l feeds row / column l & 31 and the 8 consecutive k of half l >> 5 of a 16-deep MFMA step — ONE ds_read_b128
//    per 32-row subtile per step for either operand (row stride 2 * (BK + 8) bytes: 80 for BK = 32, conflict-free for the b128
//    lane groups); A and B use the same k order, so the in-instruction k permutation is immaterial;
//  * the accumulator tile and its C/D lane mapping are those of the fp32 kernels (dtype-independent on gfx950), so gemm_epilogue,
//    gemm_finish, slab_combine are shared.
//
// Reference ops served: the same as gemm.h (SubLayers.py:39-41,54,86; Modules.py:16,23; modules.py:253-296; Layers.py:33-64;
// fastspeech2.py:97 and their autograd backward) under torch.autocast(bfloat16)-style operand rounding (BASELINE.md section 2 probe).
#pragma once
#include <type_traits>

#include "gemm.h"

namespace mtts {

struct alignas(8) u32x2 { unsigned x, y; };   // two packed bf16 pairs: one ds_write_b64
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // eight bf16 of a plane: one 16-byte load / ds_write_b128 (a register value: a struct here ended up in scratch memory behind a pointer select)

#if defined(MTTS_EMU)
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) { return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16); }
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {   // -> one v_cvt_pk_bf16_f32
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) — register-set indices must be constants BEFORE the optimiser
// looks at the staging arrays (a run-time set index that only becomes constant after unrolling left them in scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int BM, int BN, int BK>
struct GemmBf16Smem {
    static constexpr int kLDK = BK + 8;                           // bf16 elements per LDS row
    static constexpr int TILE_ELEMS = (BM + BN) * kLDK;           // one stage (A rows then B rows)
    static constexpr int FLOATS = (2 * TILE_ELEMS + 1) / 2;       // two stages, expressed in floats
};

template <int TM, int TN>
struct FragsBf16 {
#if defined(MTTS_EMU)
    float a[TM][16][16];   // the 16 A rows this lane's accumulators need x the 16 k of a step
    float b[TN][16];       // the B column this lane's accumulators need
#else
    bf16x8 a[TM], b[TN];
#endif
};

// K-loop of one output tile over the K-slices [c_lo, c_hi) of BK elements each (same contract as gemm_f32_kloop).
// PF: K-slices kept IN FLIGHT per workgroup (register sets of the global -> LDS staging).  An under-filled launch (1-2 workgroups per
// CU: every GEMM of a single-task rank, of C2, of few-shot adaptation) is bound by latency x bytes in flight per CU, not by the matrix
// pipes: with one slice in flight a workgroup moves 16 KB per ~0.7 us round trip (load -> convert -> LDS -> barrier -> MFMA).
// the MFMA steps of one staged slice: LDS stage `buf` ([BM + BN rows][BK + 8] bf16) -> this wave's accumulator tiles
template <int BM, int BN, int BK, int WGM, int WGN>
__device__ __forceinline__ void bf16_compute_slice(const bf16_t* smem, int buf, int wm0, int wn0, int lane,
                                                   f32x16 (&acc)[(BM / WGM) / 32][(BN / WGN) / 32]) {
    constexpr int kLDK = BK + 8, TM = (BM / WGM) / 32, TN = (BN / WGN) / 32, STAGE = (BM + BN) * kLDK;
    const int l31 = lane & 31, h = lane >> 5;
    const bf16_t* As = smem + buf * STAGE;
    const bf16_t* Bs = As + BM * kLDK;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
        FragsBf16<TM, TN> f;
#if defined(MTTS_EMU)
        for (int i = 0; i < TM; ++i)
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                for (int k = 0; k < 16; ++k) f.a[i][r][k] = bf16_to_f32(As[row * kLDK + ks * 16 + k]);
            }
        for (int j = 0; j < TN; ++j)
            for (int k = 0; k < 16; ++k) f.b[j][k] = bf16_to_f32(Bs[(wn0 + j * 32 + l31) * kLDK + ks * 16 + k]);
        for (int i = 0; i < TM; ++i)
            for (int j = 0; j < TN; ++j)
                for (int r = 0; r < 16; ++r) {
                    float s = acc[i][j][r];
                    for (int k = 0; k < 16; ++k) s = fmaf(f.a[i][r][k], f.b[j][k], s);
                    acc[i][j][r] = s;
                }
#else
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + i * 32 + l31) * kLDK + ks * 16 + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + j * 32 + l31) * kLDK + ks * 16 + 8 * h);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
#endif
    }
}

// NT problem staged from bf16 planes (GemmArgs::Ah / Bh: the operands as their producers rounded them).  The K-loop above without the
// conversion pass: one 16-byte load per 8 k-values, written to the LDS tile as it is — half the bytes per workgroup through L2, the bound
// of the under-filled launches (profiles/r03_gemm_variants.md section 1), and no VALU work at all between the load and the MFMA.
template <int BM, int BN, int BK, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void gemm_bf16_kloop_h(const GemmArgs& g, const GemmProb& pr, int z, int m0, int n0, int c_lo, int c_hi,
                                                  float* smem_f, f32x16 (&acc)[(BM / WGM) / 32][(BN / WGN) / 32]) {
    constexpr int NTH = 64 * WGM * WGN, kLDK = BK + 8, KQ = BK / 8, RPP = NTH / KQ, STAGE = (BM + BN) * kLDK;
    constexpr int A_LD = (BM * KQ) / NTH, B_LD = (BN * KQ) / NTH;   // 16-byte loads per thread and slice (64-tile: 1, 128-tile: 2)
    constexpr int WM = BM / WGM, WN = BN / WGN;
    static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for one 16-byte load per thread");
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_f);
    const bf16_t* A = g.Ah + (pr.A - g.A);   // (an activation plane shares its fp32 twin's layout: same element offset)
    const bf16_t* B = g.Bh + (long long)z * g.bh_gs;   // (TASK mode only: gemm_bf16_planes_ok)
    const int M = pr.M, N = pr.N, K = pr.K, K8 = (K + 7) & ~7;
    const int tid = MTTS_OPAQUE_TID(), lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int kq = (tid % KQ) * 8;
    const bf16_t* a_ptr[A_LD];
    const bf16_t* b_ptr[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) { int gm = m0 + tid / KQ + RPP * i; gm = gm < M ? gm : M - 1; a_ptr[i] = A + (long long)gm * pr.lda + kq; }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) { int gn = n0 + tid / KQ + RPP * i; gn = gn < N ? gn : N - 1; b_ptr[i] = B + (long long)gn * pr.ldb + kq; }
    u32x4 areg[A_LD], breg[B_LD];
    // the K tail: 8-element groups beyond K are read from the slice's first group (in range) and zeroed on their way into LDS — not
    // at the load, where the select would wait for the data in front of the MFMAs
    auto load = [&](int k0) {
        const int ko = (k0 + kq < K8) ? k0 : k0 - kq;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) areg[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + ko);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) breg[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + ko);
    };
    auto store = [&](int buf, int k0) {
        const unsigned keep = (k0 + kq < K8) ? 0xffffffffu : 0u;
        bf16_t* As = smem + buf * STAGE;
        bf16_t* Bs = As + BM * kLDK;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<u32x4*>(As + (tid / KQ + RPP * i) * kLDK + kq) = areg[i] & keep;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *reinterpret_cast<u32x4*>(Bs + (tid / KQ + RPP * i) * kLDK + kq) = breg[i] & keep;
    };
    const int nchunks = c_hi > c_lo ? c_hi - c_lo : 0;
    if (nchunks == 0) return;
    const int kb0 = c_lo * BK;
    load(kb0);
    store(0, kb0);
    if (nchunks > 1) load(kb0 + BK);
    __syncthreads();
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < nchunks) store(buf ^ 1, kb0 + (cc + 1) * BK);
        if (cc + 2 < nchunks) load(kb0 + (cc + 2) * BK);
        bf16_compute_slice<BM, BN, BK, WGM, WGN>(smem, buf, wm0, wn0, lane, acc);
        __syncthreads();
    }
}

template <int FORM, int BM, int BN, int BK, int PF, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void gemm_bf16_kloop(const GemmArgs& g, const GemmProb& pr, int z, int m0, int n0, bool cs_tile, int c_lo, int c_hi,
                                                float* smem_f, f32x16 (&acc)[(BM / WGM) / 32][(BN / WGN) / 32]) {
    constexpr int NTH = 64 * WGM * WGN;
    constexpr int kLDK = BK + 8;
    constexpr int KQ = BK / 4;               // float4 per K-contiguous source row
    constexpr bool A_KC = (FORM != GEMM_TN);
    constexpr bool B_KC = (FORM == GEMM_NT);
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int STAGE = (BM + BN) * kLDK;  // bf16 elements per stage
    constexpr int A_LD4 = (BM * KQ) / NTH;   // float4 loads per thread and slice (either source layout)
    constexpr int B_LD4 = (BN * KQ) / NTH;
    constexpr int RPP = NTH / KQ;            // K-contiguous rows covered per pass
    static_assert(A_LD4 >= 2 && B_LD4 >= 2 && A_LD4 % 2 == 0 && B_LD4 % 2 == 0, "tile too small for the 2(k) x 4(col) staging blocks");
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_f);
    const float* A = pr.A;
    const float* B = pr.B;
    const int M = pr.M, N = pr.N, K = pr.K;
    const int lda = pr.lda;
    int ldb = pr.ldb;

    const int tid = MTTS_OPAQUE_TID(), lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int M4 = (M + 3) & ~3, N4 = (N + 3) & ~3, K4 = (K + 3) & ~3;
    int n0b = n0, N4b = N4;   // column window of the B operand
    if (cs_tile) { B = g.colsum_w + (long long)z * g.colsum_w_gs; ldb = 4; n0b = 0; N4b = 4; }

    float4 areg[PF][A_LD4], breg[PF][B_LD4];
    int a_tap_i = 0, a_tap_base = 0, b_tap_i = 0, b_tap_base = 0;
    // (the tap geometry in registers: read through `g` inside the K-loop it is re-loaded from the kernel-argument segment every slice, and
    // the s_waitcnt lgkmcnt(0) behind each scalar load also drains the LDS fragment reads in flight)
    const int g_a_tap_k = g.a_tap_k, g_tap_k = g.tap_k, g_taps = g.taps, g_tap_bstride = g.tap_bstride, g_a_tap_rows = g.a_tap_rows;
    auto a_tap_of = [&](int k0) { while (k0 - a_tap_base >= g_a_tap_k) { a_tap_base += g_a_tap_k; ++a_tap_i; } return a_tap_i; };
    auto b_tap_of = [&](int k0) { while (k0 - b_tap_base >= g_tap_k) { b_tap_base += g_tap_k; ++b_tap_i; } return b_tap_i; };

    // per-thread source pointers, hoisted out of the K-loop and advanced by one uniform distance per slice (see gemm_f32_kloop);
    // rows / columns beyond the operand are clamped to the last valid one (their products only reach accumulator entries the epilogue
    // never stores).  Reduction-major operands: load 2q + r of a thread is k-row 2 * kk2 + r of its (k-pair kk2, column quad c4) block.
    const float* a_ptr[A_LD4];
    const float* b_ptr[B_LD4];
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) {
        if (A_KC) {
            int gm = m0 + tid / KQ + RPP * i;
            gm = gm < M ? gm : M - 1;
            a_ptr[i] = A + (long long)gm * lda + (tid % KQ) * 4;
        } else {
            const int u = tid + NTH * (i >> 1), kk = 2 * (u / (BM / 4)) + (i & 1), c4 = u % (BM / 4);
            int gc = m0 + c4 * 4;
            gc = gc < M4 ? gc : M4 - 4;
            a_ptr[i] = A + (long long)kk * lda + gc;
        }
    }
#pragma unroll
    for (int i = 0; i < B_LD4; ++i) {
        if (B_KC) {
            int gn = n0 + tid / KQ + RPP * i;
            gn = gn < N ? gn : N - 1;
            b_ptr[i] = B + (long long)gn * ldb + (tid % KQ) * 4;
        } else {
            const int u = tid + NTH * (i >> 1), kk = 2 * (u / (BN / 4)) + (i & 1), c4 = u % (BN / 4);
            int gc = n0b + c4 * 4;
            gc = gc < N4b ? gc : N4b - 4;
            b_ptr[i] = B + (long long)kk * ldb + gc;
        }
    }
    long long a_koff = 0, b_koff = 0;
    const long long a_tap_stride = (long long)g_a_tap_rows * lda;
    auto load_a = [&](int k0, auto set_c) {
        constexpr int set = decltype(set_c)::value;
        const int atap = A_KC ? a_tap_of(k0) : 0;
        const long long koff = A_KC ? (long long)atap * a_tap_stride + (k0 - a_tap_base) : (long long)k0 * lda;
        const long long delta = koff - a_koff;
        a_koff = koff;
#pragma unroll
        for (int i = 0; i < A_LD4; ++i) a_ptr[i] += delta;
        if (k0 + BK <= K) {
#pragma unroll
            for (int i = 0; i < A_LD4; ++i) areg[set][i] = ld4(a_ptr[i]);
        } else {
#pragma unroll
            for (int i = 0; i < A_LD4; ++i) {
                const bool ok = A_KC ? (k0 + (tid % KQ) * 4 < K4) : (k0 + 2 * ((tid + NTH * (i >> 1)) / (BM / 4)) + (i & 1) < K);
                areg[set][i] = ok ? ld4(a_ptr[i]) : zero4();
            }
        }
    };
    auto load_b = [&](int k0, auto set_c) {
        constexpr int set = decltype(set_c)::value;
        const int tap = B_KC ? 0 : b_tap_of(k0);
        const long long koff = B_KC ? (long long)k0 : (long long)(g_taps - 1 - tap) * g_tap_bstride + (long long)(k0 - b_tap_base) * ldb;
        const long long delta = koff - b_koff;
        b_koff = koff;
#pragma unroll
        for (int i = 0; i < B_LD4; ++i) b_ptr[i] += delta;
        if (k0 + BK <= K) {
#pragma unroll
            for (int i = 0; i < B_LD4; ++i) breg[set][i] = ld4(b_ptr[i]);
        } else {
#pragma unroll
            for (int i = 0; i < B_LD4; ++i) {
                const bool ok = B_KC ? (k0 + (tid % KQ) * 4 < K4) : (k0 + 2 * ((tid + NTH * (i >> 1)) / (BN / 4)) + (i & 1) < K);
                breg[set][i] = ok ? ld4(b_ptr[i]) : zero4();
            }
        }
    };
    // fp32 registers -> bf16 LDS image [row][kLDK] (the one rounding of this mode)
    auto store_ab = [&](int buf, auto set_c) {
        constexpr int set = decltype(set_c)::value;
        bf16_t* As = smem + buf * STAGE;
        bf16_t* Bs = As + BM * kLDK;
        if (A_KC) {
#pragma unroll
            for (int i = 0; i < A_LD4; ++i) {
                u32x2 v; v.x = pack2_bf16(areg[set][i].x, areg[set][i].y); v.y = pack2_bf16(areg[set][i].z, areg[set][i].w);
                *reinterpret_cast<u32x2*>(As + (tid / KQ + RPP * i) * kLDK + (tid % KQ) * 4) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < A_LD4 / 2; ++q) {
                const int u = tid + NTH * q, kk = 2 * (u / (BM / 4)), c = (u % (BM / 4)) * 4;
                const float4 r0 = areg[set][2 * q], r1 = areg[set][2 * q + 1];
                *reinterpret_cast<unsigned*>(As + (c + 0) * kLDK + kk) = pack2_bf16(r0.x, r1.x);
                *reinterpret_cast<unsigned*>(As + (c + 1) * kLDK + kk) = pack2_bf16(r0.y, r1.y);
                *reinterpret_cast<unsigned*>(As + (c + 2) * kLDK + kk) = pack2_bf16(r0.z, r1.z);
                *reinterpret_cast<unsigned*>(As + (c + 3) * kLDK + kk) = pack2_bf16(r0.w, r1.w);
            }
        }
        if (B_KC) {
#pragma unroll
            for (int i = 0; i < B_LD4; ++i) {
                u32x2 v; v.x = pack2_bf16(breg[set][i].x, breg[set][i].y); v.y = pack2_bf16(breg[set][i].z, breg[set][i].w);
                *reinterpret_cast<u32x2*>(Bs + (tid / KQ + RPP * i) * kLDK + (tid % KQ) * 4) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < B_LD4 / 2; ++q) {
                const int u = tid + NTH * q, kk = 2 * (u / (BN / 4)), c = (u % (BN / 4)) * 4;
                const float4 r0 = breg[set][2 * q], r1 = breg[set][2 * q + 1];
                *reinterpret_cast<unsigned*>(Bs + (c + 0) * kLDK + kk) = pack2_bf16(r0.x, r1.x);
                *reinterpret_cast<unsigned*>(Bs + (c + 1) * kLDK + kk) = pack2_bf16(r0.y, r1.y);
                *reinterpret_cast<unsigned*>(Bs + (c + 2) * kLDK + kk) = pack2_bf16(r0.z, r1.z);
                *reinterpret_cast<unsigned*>(Bs + (c + 3) * kLDK + kk) = pack2_bf16(r0.w, r1.w);
            }
        }
    };
    auto compute = [&](int buf) { bf16_compute_slice<BM, BN, BK, WGM, WGN>(smem, buf, wm0, wn0, lane, acc); };

    const int nchunks = c_hi > c_lo ? c_hi - c_lo : 0;
    if (nchunks == 0) return;
    const int kb0 = c_lo * BK;
    // slice 0 goes straight to LDS; slices 1 .. PF are put in flight (slice s >= 1 lives in register set (s - 1) % PF)
    const std::integral_constant<int, 0> set0;
    load_a(kb0, set0);
    load_b(kb0, set0);
    store_ab(0, set0);
    static_for<0, PF>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if (1 + d < nchunks) { load_a(kb0 + (1 + d) * BK, dc); load_b(kb0 + (1 + d) * BK, dc); }
    });
    __syncthreads();
    // steady state, slice cc: its successor (in flight since PF iterations) is converted into the other LDS stage, the register set it
    // leaves is refilled with slice cc + 1 + PF, then the MFMAs of slice cc run — PF slices' loads stay in flight behind them; one
    // barrier per slice hands the stage over.
    for (int c = 0; c < nchunks; c += PF) {
        static_for<0, PF>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            const int cc = c + d;
            if (cc < nchunks) {
                const int buf = cc & 1;
                if (cc + 1 < nchunks) store_ab(buf ^ 1, dc);
                if (cc + 1 + PF < nchunks) { load_a(kb0 + (cc + 1 + PF) * BK, dc); load_b(kb0 + (cc + 1 + PF) * BK, dc); }
                compute(buf);
                __syncthreads();
            }
        });
    }
}

// One workgroup's share of one problem (the bf16 twin of gemm_f32_body: same tile / split-K / column-sum / dual-source logic).
template <int FORM, int BM, int BN, int BK, int PF, bool DUAL = false>
__device__ __forceinline__ void gemm_bf16_body(const GemmArgs& g, int z, int bxs, float* smem) {
    constexpr int WGM = 2, WGN = 2, NTH = 256;
    constexpr int TM = (BM / WGM) / 32, TN = (BN / WGN) / 32;
    if (g.wave_prio > 0) MTTS_SETPRIO_HIGH();   // (the critical stream's launches beside side-stream work: GemmArgs::wave_prio)
    const GemmProb pr = gemm_resolve(g, z);
    const bool has_cs = gemm_has_colsum<FORM>(g);
    const int tiles_nc = (pr.N + BN - 1) / BN, tiles_n = tiles_nc + (has_cs ? 1 : 0), tiles_m = (pr.M + BM - 1) / BM;
    const int S = g.splitk > 1 ? g.splitk : 1;
    const int tile_lin = bxs / S, split = bxs - tile_lin * S;
    if (tile_lin >= tiles_m * tiles_n || pr.K <= 0) return;
    const int m0 = (tile_lin / tiles_n) * BM, n0 = (tile_lin % tiles_n) * BN;
    const bool cs_tile = has_cs && n0 == tiles_nc * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nch_all = (pr.K + BK - 1) / BK, cps = (nch_all + S - 1) / S;
    const int c_lo = split * cps, c_hi = (c_lo + cps < nch_all) ? c_lo + cps : nch_all;
    if constexpr (FORM == GEMM_NT_H) gemm_bf16_kloop_h<BM, BN, BK, WGM, WGN>(g, pr, z, m0, n0, c_lo, c_hi, smem, acc);
    else gemm_bf16_kloop<FORM, BM, BN, BK, PF, WGM, WGN>(g, pr, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
    if constexpr (DUAL && FORM != GEMM_NT_H) {
        if (g.A2 != nullptr && !cs_tile) {
            const GemmProb p2 = gemm_resolve2(g, z, pr);   // (the first K-loop ends on a barrier: its LDS stages are free)
            gemm_bf16_kloop<FORM, BM, BN, BK, PF, WGM, WGN>(g, p2, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
        }
    }
    if (S > 1 && !splitk_combine<TM, TN, NTH>(g, z, tile_lin, split, S, acc)) return;
    gemm_finish<TM, TN, WGM, WGN>(g, pr, z, m0, n0, cs_tile, acc);
}

template <int FORM, int BM, int BN, int BK, int PF>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[GemmBf16Smem<BM, BN, BK>::FLOATS];
    int z = blockIdx.z;
    int bxs = blockIdx.x;
    if (g.xs.on) { if (!xcd_sched_locate(g.xs, bxs, z, bxs)) return; }
    else if (g.swizzle == 2) { if (!xcd_panel_locate(bxs, g.po_tiles_m, (g.N + BN - 1) / BN + (gemm_has_colsum<FORM>(g) ? 1 : 0), g.splitk > 1 ? g.splitk : 1, bxs)) return; }
    else if (g.swizzle) bxs = xcd_group_remap(bxs, (int)gridDim.x, xcd_group_size((g.N + BN - 1) / BN, g.splitk));
    gemm_bf16_body<FORM, BM, BN, BK, PF>(g, z, bxs, smem);
}

// several independent problems in ONE launch (see gemm_f32_multi_kernel); DUAL: the launch may carry dual-source problems
template <int BM, int BN, int BK, int PF, bool DUAL>
__global__ __launch_bounds__(256) void gemm_bf16_multi_kernel(GemmMulti mp) {
    __shared__ __attribute__((aligned(16))) float smem[GemmBf16Smem<BM, BN, BK>::FLOATS];
    int p, z, bx;
    if (!gemm_multi_locate(mp, p, z, bx)) return;
    const int form = mp.form[p];
    if (form == GEMM_NT) {
        if (mp.g[p].Ah != nullptr) gemm_bf16_body<GEMM_NT_H, BM, BN, BK, PF, DUAL>(mp.g[p], z, bx, smem);   // (set by the launcher only when both planes exist)
        else gemm_bf16_body<GEMM_NT, BM, BN, BK, PF, DUAL>(mp.g[p], z, bx, smem);
    } else if (form == GEMM_NN) gemm_bf16_body<GEMM_NN, BM, BN, BK, PF, DUAL>(mp.g[p], z, bx, smem);
    else gemm_bf16_body<GEMM_TN, BM, BN, BK, PF, DUAL>(mp.g[p], z, bx, smem);
}

// a multi-problem launch that carries plane problems (non-dual, 64x64): the plane K-loop runs at HBK (64: whole 128-byte lines per row and
// slice) in an LDS allocation sized for it; the launch's other problems keep BK.  A kernel of its own: launches without planes keep the
// smaller allocation (7 instead of 4 workgroups per CU).
template <int BM, int BN, int BK, int HBK, int PF>
__global__ __launch_bounds__(256) void gemm_bf16_multi_planes_kernel(GemmMulti mp) {
    constexpr int F0 = GemmBf16Smem<BM, BN, BK>::FLOATS, F1 = GemmBf16Smem<BM, BN, HBK>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[F0 > F1 ? F0 : F1];
    int p, z, bx;
    if (!gemm_multi_locate(mp, p, z, bx)) return;
    const int form = mp.form[p];
    if (form == GEMM_NT) {
        if (mp.g[p].Ah != nullptr) gemm_bf16_body<GEMM_NT_H, BM, BN, HBK, PF, false>(mp.g[p], z, bx, smem);
        else gemm_bf16_body<GEMM_NT, BM, BN, BK, PF, false>(mp.g[p], z, bx, smem);
    } else if (form == GEMM_NN) gemm_bf16_body<GEMM_NN, BM, BN, BK, PF, false>(mp.g[p], z, bx, smem);
    else gemm_bf16_body<GEMM_TN, BM, BN, BK, PF, false>(mp.g[p], z, bx, smem);
}

constexpr int kBf16BK = 32;
// slices in flight per workgroup.  Measured (profiles/r04_bf16_gemm_microbench.md): 1 / 2 / 4 slices in flight make NO difference on
// under-filled launches (hipcc drains vmcnt(0) in front of every conversion pass, and a CU sustains ~64 outstanding lines whatever is
// queued behind them) and the extra registers cost the chip-filling launches 10-25 % (occupancy 7 -> 4 workgroups per CU): 1.
constexpr int kBf16PF64 = 1, kBf16PF128 = 1;
// a problem the bf16 K-loop can take: every conv tap must cover whole K-slices (the others — PostNet's 80-channel output layer in
// dgrad form, the vocoder's dilated taps — keep the fp32 kernels: a few per cent of the step's flops)
inline bool gemm_bf16_ok(const GemmArgs& g) {
    if (g.taps > 1 && g.tap_k % kBf16BK != 0) return false;
    if (g.a_tap_rows != 0) return false;
    return true;
}
// an NT problem whose operand planes the plane-staged K-loop can take: 16-byte loads (leading dimensions and K in whole groups of 8), plain
// K-contiguous operands (an implicit conv over overlapping rows is one), single source.  The launcher clears Ah / Bh otherwise.
inline bool gemm_bf16_planes_ok(int form, const GemmArgs& g) {
    return form == GEMM_NT && g.Ah != nullptr && g.Bh != nullptr && g.A2 == nullptr && g.table == nullptr && g.a_tap_rows == 0 &&
           g.K % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0;
}
// stand-alone launch of one problem; T = 64 / 128; pf: 0 = the default depth, else an explicit one (MTTS_BF16_PF_SWEEP builds only)
inline int gemm_bf16_launch(int form, const GemmArgs& g, int T, dim3 grid, hipStream_t stream, int pf_req = 0) {
    int pf = 0; (void)pf_req;
    dim3 block(256);
    pf = T == 128 ? kBf16PF128 : kBf16PF64;   // (explicit depths exist in MTTS_BF16_PF_SWEEP builds only)
#if defined(MTTS_BF16_PF_SWEEP)
    if (pf_req) pf = pf_req;
#endif
    if (form == GEMM_NT && g.Ah != nullptr) {   // both planes exist (gemm_bf16_planes_ok): the plane-staged K-loop
        constexpr int plane_bk = 64;   // (BK = 32 measured slower on the long convolutions: profiles/r04_c2_bf16_planes.md)
        if (T == 128) { MTTS_LAUNCH((gemm_bf16_kernel<GEMM_NT_H, 128, 128, kBf16BK, 1>), grid, block, stream, g); return GK_BF16_128_H; }
        if (plane_bk == 64 && g.K >= 512) { MTTS_LAUNCH((gemm_bf16_kernel<GEMM_NT_H, 64, 64, 64, 1>), grid, block, stream, g); return GK_BF16_64_H64; }   // whole 128-byte lines per row and slice
        MTTS_LAUNCH((gemm_bf16_kernel<GEMM_NT_H, 64, 64, kBf16BK, 1>), grid, block, stream, g);
        return GK_BF16_64_H32;
    }
#define MTTS_BF16_CASE(F, TT, PP) if (form == F && T == TT && pf == PP) { MTTS_LAUNCH((gemm_bf16_kernel<F, TT, TT, kBf16BK, PP>), grid, block, stream, g); return (TT == 128 ? GK_BF16_128 : GK_BF16_64) + F; }
#define MTTS_BF16_FORMS(TT, PP) MTTS_BF16_CASE(GEMM_NT, TT, PP) MTTS_BF16_CASE(GEMM_NN, TT, PP) MTTS_BF16_CASE(GEMM_TN, TT, PP)
    MTTS_BF16_FORMS(64, 1) MTTS_BF16_FORMS(128, 1)
#if defined(MTTS_BF16_PF_SWEEP)   // explicit slices-in-flight variants for micro-benchmarks (tile code T + 1000 * PF)
    MTTS_BF16_FORMS(64, 2) MTTS_BF16_FORMS(64, 4) MTTS_BF16_FORMS(128, 2) MTTS_BF16_FORMS(128, 3)
#endif
#undef MTTS_BF16_FORMS
#undef MTTS_BF16_CASE
    return GK_OTHER;
}
inline int gemm_bf16_multi_launch(const GemmMulti& mp, int T, bool dual, dim3 grid, hipStream_t stream) {
    dim3 block(256);
    if (T == 128) {
        if (dual) MTTS_LAUNCH((gemm_bf16_multi_kernel<128, 128, kBf16BK, kBf16PF128, true>), grid, block, stream, mp);
        else MTTS_LAUNCH((gemm_bf16_multi_kernel<128, 128, kBf16BK, kBf16PF128, false>), grid, block, stream, mp);
        return dual ? GK_BF16_MULTI128_DUAL : GK_BF16_MULTI128;
    }
    bool planes = false;
    for (int i = 0; i < mp.n; ++i) planes = planes || mp.g[i].Ah != nullptr;
    constexpr int plane_bk = 64;
    if (planes && !dual && plane_bk == 64) {
        MTTS_LAUNCH((gemm_bf16_multi_planes_kernel<64, 64, kBf16BK, 64, kBf16PF64>), grid, block, stream, mp);
        return GK_BF16_MULTI64_PLANES;
    }
    if (dual) MTTS_LAUNCH((gemm_bf16_multi_kernel<64, 64, kBf16BK, kBf16PF64, true>), grid, block, stream, mp);
    else MTTS_LAUNCH((gemm_bf16_multi_kernel<64, 64, kBf16BK, kBf16PF64, false>), grid, block, stream, mp);
    return dual ? GK_BF16_MULTI64_DUAL : GK_BF16_MULTI64;
}

}  // namespace mtts
