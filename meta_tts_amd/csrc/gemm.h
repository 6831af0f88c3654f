// Grouped fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered
// fma chain, 157 TFLOP/s dense peak on MI355X) — the one contraction kernel behind every
// Linear / Conv1d / attention product of the FastSpeech2 hot path, forward and backward.
//
// Reference ops served: SubLayers.py:39-41,54 (Linear), SubLayers.py:86
// (Conv1d k=9 / k=1), Modules.py:16,23 (bmm), modules.py:253-296 (Conv k=3), Layers.py:33-64
// (ConvNorm k=5), fastspeech2.py:97 (mel_linear) and their autograd backward.
//
// Design (MI355X-first, not a port of anything):
//  * activations are channels-last row matrices [rows][C]; a Conv1d is an *implicit GEMM over
//    overlapping rows*: row m of the im2col matrix is the contiguous span x[m-pad .. m+pad][:]
//    (lda = C_in, K = k*C_in), sequences are separated by >= pad zero guard rows, so no im2col
//    buffer, no NCL transposes, no padding copies;
//  * three operand forms cover forward, dgrad and wgrad:
//      NT  C[M,N] = A[M,K] * B[N,K]^T        (Linear / Conv forward, Q K^T, dO V^T)
//      NN  C[M,N] = A[M,K] * B[K,N]          (dgrad — conv taps walk the same [Cout][k][Cin]
//                                             weight image backwards —, P V, dS K)
//      TN  C[M,N] = A[R,M]^T * B[R,N]        (wgrad, P^T dO, dS^T Q; reduction over rows R)
//  * grouped launch: blockIdx.z = group (task of the meta-batch, or (task, sequence, head) via a
//    descriptor table), so the 8 MAML tasks — each with its own fast weights — fill the 256 CUs
//    in one launch;
//  * 256 threads = 4 waves (2x2); the block tile the engine uses is 64x64 (wave 32x32 = one MFMA tile, 16 accumulator VGPRs; 4-6
//    workgroups per CU), K-slices of BK = 32 for long K-contiguous panels (full 128-byte lines per row) and 16 otherwise;
//    128x128 (wave 64x64 = 2x2 MFMA tiles) exists for explicit tile codes.  global -> registers -> LDS double buffering with one
//    barrier per K-slice, software-pipelined (the LDS store, the barrier and the fragment reads of slice c+1 sit between the two MFMA
//    halves of slice c).  K-contiguous operands live in LDS as [row][BK+4] (ds_read_b128 conflict-free), reduction-major operands
//    as [BK][cols].  Each lane half h feeds k = 8j+4h+e into MFMA step (j,e) of both operands, which lets a K-contiguous operand
//    be fetched with two ds_read_b128 per 32-row subtile per slice.  The LDS-DMA family (gemm_glds.h) takes the under-filled launches;
//    the bf16 operand family (gemm_bf16.h) is the optional reduced-precision mode (mtts_set_numerics);
//  * fused epilogue: alpha, bias, ReLU, ReLU-mask of a saved activation (dgrad through ReLU),
//    row mask (guard / padded rows), output row remap, accumulate.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "compat.h"

namespace mtts {

enum GemmForm { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2,
                GEMM_NT_H = 3 };   // (bf16 family, kernel-internal: NT staged from the operands' bf16 planes — GemmArgs::Ah / Bh)
enum GemmFlags { GEMM_RELU = 1, GEMM_ACCUM = 2, GEMM_LRELU = 4 };   // LRELU: v < 0 -> act_slope * v (MelGAN generator)

struct GemmGroupDesc {
    long long a_off, b_off, c_off;  // element offsets added to A / B / C
    int M, N, K;
    int lda, ldb, ldc;              // per-group leading dimensions (0: use GemmArgs')
};

// Task-per-XCD schedule of a launch of 8 groups (the 8 tasks of a meta-batch on the 8 XCDs of the part; opt-in also for 4 / 2 groups,
// whose tasks are cut into 2 / 4 parts that play the role of the tasks below).  The dispatcher deals
// consecutive workgroups round-robin to the XCDs, each with a private 4 MB L2; with the plain order every XCD touches every task and
// streams every task's operands (the weight image of a dgrad, the activation panels of a wgrad) from the Infinity Cache / HBM: ~6x the
// algorithmic bytes on the 8-task meta-step.  Here workgroup slot 8 j + x belongs to XCD x, which first runs its OWN task's units
// (own[x] of them, from unit 0) and then a contiguous piece [ps[x], ps[x+1]) of the POOL — the surplus units of the tasks that carry
// more than an eighth of the launch's work (task z's units own[z] .., pool positions [P[z], P[z+1])), so the ragged tasks do not
// unbalance the XCDs.  A unit is an m-tile (all its n-tiles; M-ragged problems: forward, dgrad) or a tile (K-ragged: wgrad).
// Built on the host per launch (xcd_sched_build), speed only: any placement gives the same results.
struct XcdSched {
    int on = 0;          // 1: units are m-tiles of `tn` tiles; 2: units are tiles
    int tn = 1;
    int maxlen = 0;      // longest per-XCD list, in units (grid = 8 * maxlen * (on == 1 ? tn : 1))
    int shift = 0;       // launches of 4 / 2 groups: every group is cut into 2 / 4 contiguous parts, one per XCD (part x of group x >> shift)
    int own[8] = {0};
    int base[8] = {0};   // first unit of part x inside its group
    int ps[9] = {0};
    int P[9] = {0};
};
// workgroup slot -> (group z, tile index inside the group); false: the slot is padding
__host__ __device__ __forceinline__ bool xcd_sched_locate(const XcdSched& s, int lin, int& z, int& tile) {
    const int x = lin & 7, j = lin >> 3;
    const int ju = s.on == 1 ? j / s.tn : j, n = s.on == 1 ? j - ju * s.tn : 0;
    int u, part;
    if (ju < s.own[x]) { part = x; u = s.base[x] + ju; }
    else {
        const int idx = s.ps[x] + (ju - s.own[x]);
        if (idx >= s.ps[x + 1]) return false;
        part = 0;
        while (part < 7 && idx >= s.P[part + 1]) ++part;
        u = s.base[part] + s.own[part] + (idx - s.P[part]);
    }
    z = part >> s.shift;
    tile = s.on == 1 ? u * s.tn + n : u;
    return true;
}
// dims[z]: rows of task z (M of an M-ragged problem, K of a K-ragged one).  cls 1: units = m-tiles of tile_m rows, tn tiles each;
// cls 2: units = tiles, units_per_group of them in every group, cost proportional to dims[z].
// groups = 8, 4 or 2: with fewer than 8 groups every group is cut into 8 / groups contiguous parts and the parts play the role of the tasks
// (a task's operands are then streamed into 2 or 4 L2s instead of 8).
inline void xcd_sched_build(XcdSched& s, const int* dims, int cls, int tn, int units_per_group, int tile_m = 64, int groups = 8) {
    long long cnt[8], w[8], W = 0;
    s.shift = groups == 8 ? 0 : (groups == 4 ? 1 : 2);
    const int v = 1 << s.shift;
    for (int x = 0; x < 8; ++x) {
        const int z = x >> s.shift, part = x & (v - 1);
        const long long units = dims[z] <= 0 ? 0 : (cls == 1 ? (dims[z] + tile_m - 1) / tile_m : units_per_group);
        const long long lo = units * part / v, hi = units * (part + 1) / v;
        s.base[x] = (int)lo;
        cnt[x] = hi - lo;
        w[x] = cls == 1 ? 1 : (dims[z] > 0 ? dims[z] : 1);
        W += cnt[x] * w[x];
    }
    const long long Q = (W + 7) / 8;
    s.on = cls; s.tn = cls == 1 ? tn : 1;
    s.P[0] = 0;
    for (int z = 0; z < 8; ++z) {
        s.own[z] = (int)std::min<long long>(cnt[z], Q / w[z]);
        s.P[z + 1] = s.P[z] + (int)(cnt[z] - s.own[z]);
    }
    const int pool = s.P[8];
    int idx = 0, zt = 0;   // next pool position and the task it belongs to
    s.maxlen = 0;
    for (int x = 0; x < 8; ++x) {
        s.ps[x] = idx;
        long long load = (long long)s.own[x] * w[x];
        while (idx < pool) {
            while (zt < 7 && idx >= s.P[zt + 1]) ++zt;
            if (x < 7 && load + w[zt] / 2 + (w[zt] & 1) > Q) break;   // (the last XCD takes what is left)
            load += w[zt];
            ++idx;
        }
        s.maxlen = std::max(s.maxlen, s.own[x] + idx - s.ps[x]);
    }
    s.ps[8] = idx;
}

// Row-complete epilogue of an NT problem whose N is the whole row (N = C <= 256 columns = up to four 64-column tiles): bias + dropout +
// residual + LayerNorm behind the GEMM without a tile that spans the row.  Every tile writes its 64 x 64 piece of a = A B^T + bias THROUGH
// the L2 (sc1), the workgroups of an m-tile count themselves on one agent-scope counter, and the LAST to arrive — one acquire — reads the
// 64 complete rows back (L2-hot), applies the dropout mask, adds the residual, normalises (one wavefront per row, DPP reductions) and writes
// z (in place of a: the activation the backward keeps), y and the row statistics: exactly layernorm_fwd_kernel's arithmetic, in the GEMM's
// launch (SubLayers.py:54-55,90-91: self.layer_norm(self.dropout(sublayer(x)) + residual)).  ctr == nullptr: off.
struct LnFuse {
    const float* res = nullptr;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const unsigned char* mask = nullptr;
    float* y = nullptr;
    float* stats = nullptr;
    int* ctr = nullptr;          // [groups][m-tiles] arrival counters, zero between launches (the last arriver re-arms its own)
    long long res_gs = 0, par_gs = 0, mask_gs = 0, y_gs = 0, st_gs = 0;
    unsigned drop_seed = 0, drop_thr16 = 0;
    float drop_scale = 1.f, eps = 1e-5f;
};

struct GemmArgs {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    long long a_gs = 0, b_gs = 0, c_gs = 0;  // per-group strides (elements), TASK mode
    int lda = 0, ldb = 0, ldc = 0;
    int M = 0, N = 0, K = 0;
    const int* dimptr = nullptr;  // per-group override of M (dim_sel 0) or K (dim_sel 2)
    int dim_stride = 1, dim_sel = 0, dim_mult = 1;  // value used = dimptr[group * dim_stride] * dim_mult
    const GemmGroupDesc* table = nullptr;  // TABLE mode: per-group offsets and sizes
    const float* bias = nullptr;
    long long bias_gs = 0;
    const unsigned char* rowmask = nullptr;  // per output row; 0 -> value forced to 0
    long long rowmask_gs = 0;
    const float* relu_ref = nullptr;  // value kept only where relu_ref[m][n] > 0
    long long relu_ref_gs = 0;
    int ld_relu = 0;
    const int* c_rowmap = nullptr;  // output row remap, < 0 -> row dropped
    long long c_rowmap_gs = 0;
    float alpha = 1.f;
    float act_slope = 0.2f;
    int flags = 0;
    int taps = 1, tap_k = 0x40000000, tap_bstride = 0;  // NN conv dgrad tap walk
    // dilated taps of a K-contiguous A operand (NT / NN): k = tap * a_tap_k + kin reads row m + tap * a_tap_rows — a dilated
    // Conv1d as ONE implicit GEMM (K = taps * C_in) instead of one accumulate pass per tap (vocoder.h).  K-slices must not
    // straddle taps: a_tap_k % 32 == 0.
    int a_tap_k = 0x40000000, a_tap_rows = 0;
    int swizzle = 0;                                     // XCD-aware tile order (set by the launcher): 1 = xcd_group_remap, 2 = panel order (xcd_panel_locate)
    int po_tiles_m = 0;                                  // panel order: m-tiles of the tile grid (from the launch's max_M)
    // split-K (set by the launcher for under-filled grids): `splitk` workgroups share one output tile, each reducing a
    // contiguous run of K-chunks; partial tiles go to `ws`, the last workgroup to arrive (tile counter in `tile_ctr`) sums
    // them in split order and runs the fused epilogue, so the result does not depend on the arrival order
    int splitk = 1;
    int tiles_pg = 0;  // output tiles per group (workspace slot = group * tiles_pg + tile)
    float* ws = nullptr;
    int* tile_ctr = nullptr;
    // TN form only: column sums of the A operand over the reduction rows (the bias gradient db = sum_rows dY beside the weight
    // gradient dW = dY^T X) as ONE EXTRA n-tile per m-tile whose B operand is `colsum_w`: [rows][4] floats, 1.0 on the rows
    // that count and 0.0 on masked ones, read with ldb = 4 — column 0 of that tile is the masked column sum, produced by the same
    // K-loop (k-ordered fp32 MFMA chain, so the order of summation is fixed), no extra registers or instructions in the loop and
    // no separate reduction launches.  Launchers add the n-tile to the grid (gemm_tiles_n) and keep splitk == 1.
    float* colsum = nullptr;
    long long colsum_gs = 0;
    const float* colsum_w = nullptr;
    long long colsum_w_gs = 0;
    // Dual source: C = epilogue(A·B + A2·B2) in ONE accumulator chain — the K-loop runs over (A, B) and then over (A2, B2), which share
    // every size, leading dimension, tap walk and (TABLE mode) per-group offset with the first pair; only the bases and the TASK-mode
    // group strides differ.  The tangent products of second-order MAML come in such pairs (t(XW) = tX·W + X·tW): one launch, one
    // epilogue and no read-modify-write of C instead of a plain launch followed by an accumulating one.  The column-sum tile
    // (colsum) belongs to the first pair only.
    const float* A2 = nullptr;
    const float* B2 = nullptr;
    long long a2_gs = 0, b2_gs = 0;
    // bf16 operand planes (bf16 numerics mode only, gemm_bf16.h).  Ah / Bh: the SAME operands as A / B, element for element (same leading
    // dimensions, group strides and offsets), already rounded to bf16 by their producer — an NT problem with both set is staged from the
    // planes (half the bytes through L2, no conversion pass) instead of from A / B.  Ch: bf16 twin of C, written by the epilogue beside
    // the fp32 value (the next GEMM's plane).
    // Bh has its own group stride (a weight shadow is packed differently from its fp32 master); plane_only: the problem exists in plane
    // form alone (an input-gradient conv as NT over the transposed shadow: A / B give offsets and sizes, B is never dereferenced).
    const bf16_t* Ah = nullptr;
    const bf16_t* Bh = nullptr;
    long long bh_gs = 0;
    bf16_t* Ch = nullptr;
    bool plane_only = false;
    LnFuse ln;   // row-complete LayerNorm epilogue (NT, N <= 256; the launcher never splits K of such a problem)
    // task-per-XCD schedule (launches of 8 — opt-in 4 / 2 — groups).  host_dims: HOST array of the groups' dimptr values, read by the
    // launcher to build `xs`; never dereferenced on the device.
    const int* host_dims = nullptr;
    XcdSched xs;
    // > 0: the launch's wavefronts raise their issue priority (s_setprio 3) — set by the launcher from GemmCtx::wave_prio for the launches of
    // the CRITICAL stream of a multi-stream step, so that on a SIMD they share with a side stream's wavefronts (priority 0) the critical
    // chain's MFMAs / LDS reads issue first and the side work takes the slots the chain leaves (speed only: never changes a result)
    int wave_prio = 0;
};
// K-loop length of a problem for the launch heuristics (both sources of a dual-source problem)
inline int gemm_keff(const GemmArgs& g) { return g.A2 ? 2 * g.K : g.K; }


// XCD-aware tile order (speed only, never correctness).  The dispatcher deals consecutive workgroups round-robin to the 8
// XCDs, each with a private 4 MB L2, so with the natural order the n-tiles of one m-tile — which read the same A panel —
// land on 8 different L2s.  Within every run of 8*G workgroup slots the G slots of residue r (mod 8) are mapped to G
// consecutive tiles instead: with G = tiles_n (x split-K factor) one m-tile's n-tiles share an XCD while every XCD still
// receives every eighth workgroup — no imbalance (a contiguous eighth of the grid per XCD measured -3 % on the ragged
// 8-task launches).  PMC: L2-miss traffic of the multi-problem launches 454 -> 286 MB per launch; time neutral (the kernels
// are MFMA-, not fetch-bound).  MTTS_XCD_GROUP=0 restores the natural order.
__device__ __forceinline__ int xcd_group_size(int tiles_n, int splitk) {
    const int G = tiles_n * (splitk > 1 ? splitk : 1);
    return G < 64 ? G : 64;
}
__device__ __forceinline__ int xcd_group_remap(int lin, int total, int G) {
    if (G <= 1) return lin;
    const int run = 8 * G, blk = lin / run;
    if ((blk + 1) * run > total) return lin;  // ragged last run: natural order
    const int r = lin - blk * run;
    return blk * run + (r & 7) * G + (r >> 3);
}

// Panel order (launches of fewer than 8 groups: a single-task rank, C2, few-shot adaptation — where no task-per-XCD schedule applies).
// With the m-tile-major grouping above an XCD walks a few m-tiles across ALL n-tiles, so it streams the whole B operand — the 9.4 MB
// weight image of a k = 9 conv against a 4 MB L2 — once per m-tile: every weight line crosses the fabric up to tiles_m times.  Here the
// tile list is ordered B-panel-major ((n, split) outermost, m innermost) and dealt to the XCDs in contiguous eighths (slot 8 j + x runs
// on XCD x): an XCD works through one or two (n, split) units, whose B panel (64 columns x its K range: 0.6-2.4 MB) stays in its L2
// while the m-tiles' A rows stream past — the weight image crosses the fabric once.  Placement only: any order gives the same results.
// Returns false for padding slots; bxs = (m * tiles_n + n) * S + split as gemm_*_body decodes it.
__host__ __device__ __forceinline__ int xcd_panel_slots(int tiles_m, int tiles_n, int S) { return 8 * ((tiles_m * tiles_n * S + 7) / 8); }
__host__ __device__ __forceinline__ bool xcd_panel_locate(int lin, int tiles_m, int tiles_n, int S, int& bxs) {
    const int total = tiles_m * tiles_n * S, R = (total + 7) / 8;
    const int x = lin & 7, j = lin >> 3;
    const int t = x * R + j;
    if (j >= R || t >= total) return false;
    const int m = t % tiles_m, u = t / tiles_m;
    const int sp = u % S, n = u / S;
    bxs = (m * tiles_n + n) * S + sp;
    return true;
}

// Register fragments of one BK=16 slice of the wave tile.  A_KC/B_KC: operand tile is [row][kLDK]
// (K-contiguous: two ds_read_b128 per 32-row subtile) or [k][LD] (reduction-major: ds_read_b32).
// Lane half h = lane>>5 holds k = 8*j2 + 4*h + e for MFMA step (j2, e) of both operands.
// acc[i][j] is the 32x32 MFMA C tile: (lane l, reg r) -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
template <int TM, int TN, int BK>
struct Frags {
#if defined(MTTS_EMU)
    float a[TM][16][BK];  // the 16 A rows this lane's accumulators need, all k
    float b[TN][BK];      // the B column this lane's accumulators need, all k
#else
    float a[BK / 8][TM][4], b[BK / 8][TN][4];
#endif
};

template <int TM, int TN, int BK, bool A_KC, bool B_KC, int LDA, int LDB>
__device__ __forceinline__ void read_frags(const float* As, const float* Bs, int wm0, int wn0, int lane, Frags<TM, TN, BK>& f) {
    const int l31 = lane & 31, h = lane >> 5;
#if defined(MTTS_EMU)
    for (int i = 0; i < TM; ++i)
        for (int r = 0; r < 16; ++r) {
            const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            for (int k = 0; k < BK; ++k) f.a[i][r][k] = A_KC ? As[row * LDA + k] : As[k * LDA + row];
        }
    for (int j = 0; j < TN; ++j) {
        const int col = wn0 + j * 32 + l31;
        for (int k = 0; k < BK; ++k) f.b[j][k] = B_KC ? Bs[col * LDB + k] : Bs[k * LDB + col];
    }
#else
#pragma unroll
    for (int j2 = 0; j2 < BK / 8; ++j2) {
        const int kb = 8 * j2 + 4 * h;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (A_KC) {
                const float4 v = ld4(As + (wm0 + i * 32 + l31) * LDA + kb);
                f.a[j2][i][0] = v.x; f.a[j2][i][1] = v.y; f.a[j2][i][2] = v.z; f.a[j2][i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.a[j2][i][e] = As[(kb + e) * LDA + wm0 + i * 32 + l31];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (B_KC) {
                const float4 v = ld4(Bs + (wn0 + j * 32 + l31) * LDB + kb);
                f.b[j2][j][0] = v.x; f.b[j2][j][1] = v.y; f.b[j2][j][2] = v.z; f.b[j2][j][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.b[j2][j][e] = Bs[(kb + e) * LDB + wn0 + j * 32 + l31];
            }
        }
    }
#endif
}

// MFMA steps of half `hf` (0/1) of the slice: k = hf*BK/2 .. (hf+1)*BK/2 - 1
template <int TM, int TN, int BK>
__device__ __forceinline__ void mma_half(const Frags<TM, TN, BK>& f, int hf, f32x16 (&acc)[TM][TN]) {
#if defined(MTTS_EMU)
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) {
                float s = acc[i][j][r];
                for (int k = hf * (BK / 2); k < (hf + 1) * (BK / 2); ++k) s = fmaf(f.a[i][r][k], f.b[j][k], s);
                acc[i][j][r] = s;
            }
#else
#pragma unroll
    for (int q = 0; q < BK / 16; ++q) {
        const int j2 = hf * (BK / 16) + q;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j2][i][e], f.b[j2][j][e], acc[i][j], 0, 0, 0);
    }
#endif
}

// ---- register-lean K-loop (KL >= 1, BK = 32, device builds) ----------------------------------------------------------------------
// Half fragments: the operands of ONE half (16 k-values) of a slice.  The rotating loop of gemm_f32_kloop keeps two of them (32 VGPRs for a
// 32x32 wave tile) instead of two whole-slice fragment sets (64): the dominant kernels drop from 124 + 16 to under 112 + 16 registers,
// i.e. from 3 to 4 resident waves per SIMD (the unified VGPR file holds 512 per lane).
template <int TM, int TN, int NQ>   // NQ = BK / 16 groups of 8 k-values per half
struct HalfFrags { float a[NQ][TM][4], b[NQ][TN][4]; };
template <int V> struct KlTag { static constexpr int value = V; };   // 3: store + load (steady state), 2: store only, 1: last slice, 0: generic (tested per slice)

#if !defined(MTTS_EMU)
template <int TM, int TN, int NQ, bool A_KC, bool B_KC, int LDA, int LDB>
__device__ __forceinline__ void read_half(const float* As, const float* Bs, int wm0, int wn0, int lane, int hf, HalfFrags<TM, TN, NQ>& f) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int kb = 8 * (NQ * hf + q) + 4 * h;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (A_KC) {
                const float4 v = ld4(As + (wm0 + i * 32 + l31) * LDA + kb);
                f.a[q][i][0] = v.x; f.a[q][i][1] = v.y; f.a[q][i][2] = v.z; f.a[q][i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.a[q][i][e] = As[(kb + e) * LDA + wm0 + i * 32 + l31];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (B_KC) {
                const float4 v = ld4(Bs + (wn0 + j * 32 + l31) * LDB + kb);
                f.b[q][j][0] = v.x; f.b[q][j][1] = v.y; f.b[q][j][2] = v.z; f.b[q][j][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.b[q][j][e] = Bs[(kb + e) * LDB + wn0 + j * 32 + l31];
            }
        }
    }
}
// the 8 x TM x TN MFMAs of a half, in the k order of mma_half (results are bit-identical to the KL = 0 loop)
template <int TM, int TN, int NQ>
__device__ __forceinline__ void mma_hfrags(const HalfFrags<TM, TN, NQ>& f, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q][i][e], f.b[q][j][e], acc[i][j], 0, 0, 0);
}
#endif

// one float written through the XCD's L2 (see st4_through below)
__device__ __forceinline__ void st1_through(float* p, float v) {
#if defined(MTTS_EMU)
    *p = v;
#else
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
}

// Fused epilogue shared by the fp32 and the split-bf16 kernels: alpha, bias, accumulate, ReLU, ReLU-mask of a
// saved activation, row mask, output row remap.  (mb, nb) = origin of this wave's tile.
// LNF: the kernel instantiation carries the row-complete LayerNorm epilogue (LnFuse) — only those pay for its stores, barrier and tail code
template <int TM, int TN, bool LNF = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, int z, f32x16 (&acc)[TM][TN], float* C, int ldc, int M, int N,
                                              int mb, int nb, int lane) {
    const float* bias = g.bias ? g.bias + (long long)z * g.bias_gs : nullptr;
    const unsigned char* rowmask = g.rowmask ? g.rowmask + (long long)z * g.rowmask_gs : nullptr;
    const float* relu_ref = g.relu_ref ? g.relu_ref + (long long)z * g.relu_ref_gs : nullptr;
    const int* rowmap = g.c_rowmap ? g.c_rowmap + (long long)z * g.c_rowmap_gs : nullptr;
    bf16_t* Ch = g.Ch ? g.Ch + (C - g.C) : nullptr;   // (C carries the group's offset into g.C; the twin has the same layout)
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nb + j * 32 + l31;
            const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= M || n >= N) continue;
                int mo = m;
                if (rowmap) { mo = rowmap[m]; if (mo < 0) continue; }
                float* p = C + (long long)mo * ldc + n;
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.flags & GEMM_ACCUM) v += *p;  // accumulate first: the masks below act on the sum
                if (g.flags & GEMM_RELU) v = fmaxf(v, 0.f);
                if (g.flags & GEMM_LRELU) v = v > 0.f ? v : g.act_slope * v;
                if (relu_ref && !(relu_ref[(long long)m * g.ld_relu + n] > 0.f)) v = 0.f;
                if (rowmask && !rowmask[m]) v = 0.f;
                if (LNF && g.ln.ctr) st1_through(p, v);   // (read back by the m-tile's last workgroup: LnFuse)
                else *p = v;
                if (Ch) Ch[(long long)mo * ldc + n] = f32_to_bf16(v);
            }
        }
}

// Partial-tile slab stores.  The slab is written THROUGH the XCD's L2 (sc1), so publishing it needs no release fence — a
// fence(release, agent) writes back every dirty line of that L2, which between GEMM epilogues costs far more than the slab itself
// (cdna_hip_programming.md, Guideline 16 form R1: write-through payload, every storing wave drains, one relaxed agent-scope
// counter increment, ONE acquire on the reducing workgroup, plain loads).
__device__ __forceinline__ void st4_through(float* p, float4 v) {
#if defined(MTTS_EMU)
    st4(p, v);
#else
    f32x4 x; x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    // (s_nop: a store of more than 8 bytes must not have its data VGPRs overwritten by the very next VALU instruction — a hazard hipcc
    // pads for its own stores but cannot see inside an asm; without it the slab held garbage whenever the register allocation reused x)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(x) : "memory");
#endif
}

// Rendezvous of the S workgroups that share one output tile: each parks its partial tile in a slab (`base` + split * slab size); the
// last to arrive (`ctr`) reloads all S partials in split order (its own included, so the sum does not depend on the arrival order)
// and returns true to run the epilogue; it also re-arms the counter for the next launch.
template <int TM, int TN, int NTH>
__device__ __forceinline__ bool slab_combine(float* base, int* ctr, int split, int S, f32x16 (&acc)[TM][TN]) {
    constexpr int PART = NTH * 16 * TM * TN;  // floats per partial tile
    const int tid = threadIdx.x;
    const bool act = tid < NTH;   // (workgroups larger than the tile's NTH threads only take part in the barriers)
    float* mine = base + (long long)split * PART;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (!act) continue;
                float4 v;
                v.x = acc[i][j][4 * r4]; v.y = acc[i][j][4 * r4 + 1]; v.z = acc[i][j][4 * r4 + 2]; v.w = acc[i][j][4 * r4 + 3];
                st4_through(mine + (((i * TN + j) * 4 + r4) * NTH + tid) * 4, v);
            }
    __shared__ int s_last;
    MTTS_WAIT_VMEM();      // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        s_last = (MTTS_ATOMIC_INC_AGENT(ctr) == S - 1) ? 1 : 0;
        if (s_last) MTTS_FENCE_ACQUIRE_AGENT();
    }
    __syncthreads();
    if (!s_last) return false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (!act) continue;
                float4 sum = zero4();
                for (int sp = 0; sp < S; ++sp) {
                    const float4 v = ld4(base + (long long)sp * PART + (((i * TN + j) * 4 + r4) * NTH + tid) * 4);
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                }
                acc[i][j][4 * r4] = sum.x; acc[i][j][4 * r4 + 1] = sum.y; acc[i][j][4 * r4 + 2] = sum.z; acc[i][j][4 * r4 + 3] = sum.w;
            }
    if (tid == 0) *ctr = 0;
    return true;
}
// split-K of a plain grid: slab / counter of output tile `tile_lin` of group z
template <int TM, int TN, int NTH>
__device__ __forceinline__ bool splitk_combine(const GemmArgs& g, int z, int tile_lin, int split, int S, f32x16 (&acc)[TM][TN]) {
    const long long slot = (long long)z * g.tiles_pg + tile_lin;
    return slab_combine<TM, TN, NTH>(g.ws + slot * S * (NTH * 16 * TM * TN), g.tile_ctr + slot, split, S, acc);
}

// WGM x WGN waves per workgroup (64 threads each); the wave tile is (BM/WGM) x (BN/WGN) = TM x TN MFMA tiles.
template <int FORM, int BM, int BN, int BK>
struct GemmSmem {
    static constexpr int kLDK = BK + 4;
    static constexpr int A_TILE = (FORM != GEMM_TN) ? BM * kLDK : BK * BM;
    static constexpr int B_TILE = (FORM == GEMM_NT) ? BN * kLDK : BK * BN;
    static constexpr int FLOATS = 2 * (A_TILE + B_TILE);
};

// A problem of a grouped launch resolved for one group z: operand bases, sizes and leading dimensions.
struct GemmProb {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc;
};
__device__ __forceinline__ GemmProb gemm_resolve(const GemmArgs& g, int z) {
    GemmProb pr{g.A, g.B, g.C, g.M, g.N, g.K, g.lda, g.ldb, g.ldc};
    if (g.table) {
        const GemmGroupDesc d = g.table[z];
        pr.A += d.a_off; pr.B += d.b_off; pr.C += d.c_off;
        pr.M = d.M; pr.N = d.N; pr.K = d.K;
        if (d.lda) pr.lda = d.lda;
        if (d.ldb) pr.ldb = d.ldb;
        if (d.ldc) pr.ldc = d.ldc;
    } else {
        pr.A += (long long)z * g.a_gs; pr.B += (long long)z * g.b_gs; pr.C += (long long)z * g.c_gs;
        if (g.dimptr) {
            const int v = g.dimptr[(long long)z * g.dim_stride] * g.dim_mult;
            if (g.dim_sel == 0) pr.M = v; else pr.K = v;
        }
    }
    return pr;
}
// second source of a dual-source problem (GemmArgs::A2 / B2) for group z: everything but the operand bases is the first source's
__device__ __forceinline__ GemmProb gemm_resolve2(const GemmArgs& g, int z, const GemmProb& pr) {
    GemmProb q = pr;
    if (g.table) {
        const GemmGroupDesc d = g.table[z];
        q.A = g.A2 + d.a_off; q.B = g.B2 + d.b_off;
    } else {
        q.A = g.A2 + (long long)z * g.a2_gs; q.B = g.B2 + (long long)z * g.b2_gs;
    }
    return q;
}
// n-tiles of a resolved problem: the tiles of C plus the column-sum tile (GemmArgs::colsum, TN form only)
template <int FORM>
__device__ __forceinline__ bool gemm_has_colsum(const GemmArgs& g) { return (FORM == GEMM_TN) && g.colsum != nullptr && !g.table; }

// K-loop of one output tile (origin m0, n0) of a resolved problem over the K-chunks [c_lo, c_hi): the products are added into acc.
// cs_tile: the tile is the column-sum tile (its B operand is GemmArgs::colsum_w).
// ABL (diagnostic builds only, results are wrong): bit 0 drops the in-loop global loads, bit 1 the LDS stores, bit 2 the
// in-loop fragment reads, bit 3 the barrier — timing the kernel with one stage removed shows what that stage costs.
// KL (K-loop variant of the pipelined 64x64 kernels; measured in profiles/r05_kloop_ab.md): 0 = two whole-slice fragment sets, clustered
// phases (rounds 2-4); 1 = rotating half fragments + loop-invariant pointer steps; 4 = 1 with the slice's LDS / global instructions
// interleaved into the MFMA chain (sched_group_barrier) — the default.  (Arms 2 / 3 of the A/B, s_setprio around the MFMA clusters of
// 1 / 0, measured 3 % SLOWER on this lock-step structure and are gone.)
template <int FORM, int BM, int BN, int BK, bool PIPE, int WGM = 2, int WGN = 2, int ABL = 0, int KL = 0>
__device__ __forceinline__ void gemm_f32_kloop(const GemmArgs& g, const GemmProb& pr, int z, int m0, int n0, bool cs_tile, int c_lo, int c_hi,
                                               float* smem, f32x16 (&acc)[(BM / WGM) / 32][(BN / WGN) / 32]) {
    constexpr int NTH = 64 * WGM * WGN;
    constexpr int kLDK = BK + 4;  // K-contiguous LDS row stride: 80 / 144 bytes, ds_read_b128 conflict-free
    constexpr int KQ = BK / 4;    // float4 per K-contiguous row
    constexpr bool A_KC = (FORM != GEMM_TN);
    constexpr bool B_KC = (FORM == GEMM_NT);
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int LDA_S = A_KC ? kLDK : BM;
    constexpr int LDB_S = B_KC ? kLDK : BN;
    constexpr int A_TILE = A_KC ? BM * kLDK : BK * BM;
    constexpr int B_TILE = B_KC ? BN * kLDK : BK * BN;
    constexpr int A_LD4 = (BM * KQ) / NTH;  // float4 loads per thread (both layouts: BM*BK/4 float4 per tile)
    constexpr int B_LD4 = (BN * KQ) / NTH;
    constexpr int RPP = NTH / KQ;           // K-contiguous rows covered per pass
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

    float4 areg[A_LD4], breg[B_LD4];
    // tap of the K-slice starting at k0 WITHOUT a runtime integer division per slice (k0 only ever grows inside a workgroup, so
    // a running (tap, tap * tap_k) pair is advanced instead; the division cost ~25 scalar/vector instructions per operand per slice)
    int a_tap_i = 0, a_tap_base = 0, b_tap_i = 0, b_tap_base = 0;
    // (the tap geometry in registers: read through `g` inside the K-loop it is re-loaded from the kernel-argument segment every slice, and
    // the s_waitcnt lgkmcnt(0) behind each scalar load also drains the LDS fragment reads in flight)
    const int g_a_tap_k = g.a_tap_k, g_tap_k = g.tap_k, g_taps = g.taps, g_tap_bstride = g.tap_bstride, g_a_tap_rows = g.a_tap_rows;
    auto a_tap_of = [&](int k0) { while (k0 - a_tap_base >= g_a_tap_k) { a_tap_base += g_a_tap_k; ++a_tap_i; } return a_tap_i; };
    auto b_tap_of = [&](int k0) { while (k0 - b_tap_base >= g_tap_k) { b_tap_base += g_tap_k; ++b_tap_i; } return b_tap_i; };

    // Per-thread operand offsets are hoisted out of the K-loop; a slice then costs one uniform 64-bit base per operand
    // (the per-slice 64-bit row * lda multiplies and the exec-mask branches of predicated loads cost ~60 instructions
    // per slice before).  Rows / columns beyond the operand are CLAMPED to the last valid one instead of predicated: their products
    // only reach accumulator entries the epilogue never stores (m >= M or n >= N).  Only a partial LAST K-slice is predicated
    // (zero fill: there both operands would otherwise contribute garbage to valid outputs).
    // Loop-carried per-thread pointers, advanced in place by the uniform distance between consecutive slices (one 64-bit add per
    // pointer per slice, no temporaries: an address temporary that aliases a load destination made hipcc drain vmcnt between the A
    // and the B loads of a slice).
    const float* a_ptr[A_LD4];
    const float* b_ptr[B_LD4];
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) {
        if (A_KC) {
            int gm = m0 + tid / KQ + RPP * i;
            gm = gm < M ? gm : M - 1;
            a_ptr[i] = A + (long long)gm * lda + (tid % KQ) * 4;
        } else {
            const int idx = tid + NTH * i, kk = idx / (BM / 4), c4 = idx % (BM / 4);
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
            const int idx = tid + NTH * i, kk = idx / (BN / 4), c4 = idx % (BN / 4);
            int gc = n0b + c4 * 4;
            gc = gc < N4b ? gc : N4b - 4;
            b_ptr[i] = B + (long long)kk * ldb + gc;
        }
    }
    long long a_koff = 0, b_koff = 0;   // uniform element offset of the slice the pointers currently address
    const long long a_tap_stride = (long long)g_a_tap_rows * lda;
    auto load_a = [&](int k0) {
        const int atap = A_KC ? a_tap_of(k0) : 0;   // 0 unless the operand has dilated taps
        const long long koff = A_KC ? (long long)atap * a_tap_stride + (k0 - a_tap_base) : (long long)k0 * lda;
        const long long delta = koff - a_koff;
        a_koff = koff;
#pragma unroll
        for (int i = 0; i < A_LD4; ++i) a_ptr[i] += delta;
        if (k0 + BK <= K) {
#pragma unroll
            for (int i = 0; i < A_LD4; ++i) areg[i] = ld4(a_ptr[i]);
        } else {
#pragma unroll
            for (int i = 0; i < A_LD4; ++i) {
                const bool ok = A_KC ? (k0 + (tid % KQ) * 4 < K4) : (k0 + (tid + NTH * i) / (BM / 4) < K);
                areg[i] = ok ? ld4(a_ptr[i]) : zero4();
            }
        }
    };
    auto load_b = [&](int k0) {
        const int tap = B_KC ? 0 : b_tap_of(k0);
        const long long koff = B_KC ? (long long)k0 : (long long)(g_taps - 1 - tap) * g_tap_bstride + (long long)(k0 - b_tap_base) * ldb;
        const long long delta = koff - b_koff;
        b_koff = koff;
#pragma unroll
        for (int i = 0; i < B_LD4; ++i) b_ptr[i] += delta;
        if (k0 + BK <= K) {
#pragma unroll
            for (int i = 0; i < B_LD4; ++i) breg[i] = ld4(b_ptr[i]);
        } else {
#pragma unroll
            for (int i = 0; i < B_LD4; ++i) {
                const bool ok = B_KC ? (k0 + (tid % KQ) * 4 < K4) : (k0 + (tid + NTH * i) / (BN / 4) < K);
                breg[i] = ok ? ld4(b_ptr[i]) : zero4();
            }
        }
    };
    auto store_ab = [&](int buf) {
        float* As = smem + buf * (A_TILE + B_TILE);
        float* Bs = As + A_TILE;
#pragma unroll
        for (int i = 0; i < A_LD4; ++i) {
            if (A_KC) st4(As + (tid / KQ + RPP * i) * kLDK + (tid % KQ) * 4, areg[i]);
            else { const int idx = tid + NTH * i; st4(As + (idx / (BM / 4)) * BM + (idx % (BM / 4)) * 4, areg[i]); }
        }
#pragma unroll
        for (int i = 0; i < B_LD4; ++i) {
            if (B_KC) st4(Bs + (tid / KQ + RPP * i) * kLDK + (tid % KQ) * 4, breg[i]);
            else { const int idx = tid + NTH * i; st4(Bs + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4, breg[i]); }
        }
    };

    const int nchunks = c_hi > c_lo ? c_hi - c_lo : 0;  // this workgroup's run of K-chunks
    if (nchunks == 0) return;
    const int kb0 = c_lo * BK;
    load_a(kb0);
    load_b(kb0);
    store_ab(0);
    __syncthreads();
#if !defined(MTTS_EMU)
    if constexpr (PIPE && (BK == 32 || BK == 16) && (KL == 1 || KL == 4)) {
        constexpr int NQ = BK / 16, NMF = 4 * NQ;   // groups of 8 k-values / MFMA steps per half
        // Rotating half fragments.  Per slice c:  [read half 1 of c | store c+1 to LDS | issue the loads of c+2]  MFMA(half 0)  barrier
        // [read half 0 of c+1]  MFMA(half 1) — every LDS read is issued one MFMA cluster (8 x 64 cycles) before its first use.
        // Pointer steps: the K-slices of a workgroup are consecutive, so an operand's pointer advances by one of two loop-invariant
        // distances (inside a conv tap / across a tap boundary: taps are multiples of 32) chosen by one scalar compare — the generic
        // load_a / load_b above re-derive the tap from k0 every slice (~40 scalar instructions and two loops per slice).
        int a_in = 0, b_in = 0;   // offset of the NEXT slice inside its tap
        {
            const int k1 = kb0;   // (position after the prologue's load of slice kb0)
            a_in = A_KC ? k1 - a_tap_base : 0;
            b_in = B_KC ? 0 : k1 - b_tap_base;
        }
        const long long a_s1 = A_KC ? (long long)BK : (long long)BK * lda;
        const long long a_s2 = A_KC ? a_tap_stride - (long long)(g_a_tap_k - BK) : a_s1;
        const long long b_s1 = B_KC ? (long long)BK : (long long)BK * ldb;
        const long long b_s2 = B_KC ? b_s1 : -(long long)g_tap_bstride - (long long)(g_tap_k - BK) * ldb;
        int k_next = kb0;
        // fast: the slice is known to be a full one (no zero-filled tail) — straight-line code, which is what lets the scheduler interleave
        auto load_next = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value == 3;
            k_next += BK;
            a_in += BK; b_in += BK;
            const bool a_x = A_KC && a_in >= g_a_tap_k, b_x = !B_KC && b_in >= g_tap_k;
            const long long da = a_x ? a_s2 : a_s1, db = b_x ? b_s2 : b_s1;
            if (a_x) a_in = 0;
            if (b_x) b_in = 0;
#pragma unroll
            for (int i = 0; i < A_LD4; ++i) a_ptr[i] += da;
#pragma unroll
            for (int i = 0; i < B_LD4; ++i) b_ptr[i] += db;
            if (FAST || k_next + BK <= K) {
#pragma unroll
                for (int i = 0; i < A_LD4; ++i) areg[i] = ld4(a_ptr[i]);
#pragma unroll
                for (int i = 0; i < B_LD4; ++i) breg[i] = ld4(b_ptr[i]);
            } else {
#pragma unroll
                for (int i = 0; i < A_LD4; ++i) {
                    const bool ok = A_KC ? (k_next + (tid % KQ) * 4 < K4) : (k_next + (tid + NTH * i) / (BM / 4) < K);
                    areg[i] = ok ? ld4(a_ptr[i]) : zero4();
                }
#pragma unroll
                for (int i = 0; i < B_LD4; ++i) {
                    const bool ok = B_KC ? (k_next + (tid % KQ) * 4 < K4) : (k_next + (tid + NTH * i) / (BN / 4) < K);
                    breg[i] = ok ? ld4(b_ptr[i]) : zero4();
                }
            }
        };
        HalfFrags<TM, TN, NQ> h0, h1;
        constexpr int NDR = NQ * ((A_KC ? 1 : 4) * TM + (B_KC ? 1 : 4) * TN);   // LDS read instructions of a half (ds_read_b128 / ds_read_b32)
        auto body = [&](int c, auto mode_tag) {
            // MODE 3: slices c + 1 and c + 2 exist and c + 2 is a whole one (steady state); 2: c + 1 exists, nothing left to load; 1: the last
            // slice; 0: decided per slice (the iteration that loads a zero-filled tail).  Modes 1-3 are straight-line code — which is what
            // lets the scheduler interleave (KL = 4).
            constexpr int MODE = decltype(mode_tag)::value;
            const int nb = (c & 1) ^ 1;
            const float* Ac = smem + (c & 1) * (A_TILE + B_TILE);
            read_half<TM, TN, NQ, A_KC, B_KC, LDA_S, LDB_S>(Ac, Ac + A_TILE, wm0, wn0, lane, 1, h1);
            if (MODE >= 2 || (MODE == 0 && c + 1 < nchunks)) store_ab(nb);
            if (MODE == 3) load_next(mode_tag);
            else if (MODE == 0 && c + 2 < nchunks) load_next(mode_tag);
            if (KL != 4) __builtin_amdgcn_sched_barrier(0);
            mma_hfrags<TM, TN, NQ>(h0, acc);
            if (KL == 4 && MODE != 0) {
                // one group of memory instructions behind each MFMA of the cluster: the chain is dependent (an MFMA issues when its
                // predecessor retires, 64 cycles later), so the LDS reads / writes and the global loads ride in its shadow instead of in front of it
                constexpr int RD = NMF / 2, WR = NMF / 4;   // MFMA steps that carry reads / writes (the rest: loads)
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                                                            // MFMA
                    if (q < RD) __builtin_amdgcn_sched_group_barrier(0x100, (NDR + RD - 1) / RD, 0);                                    // DS read (half 1 of this slice)
                    else if (q < RD + WR) { if (MODE >= 2) __builtin_amdgcn_sched_group_barrier(0x200, (A_LD4 + B_LD4 + WR - 1) / WR, 0); }   // DS write (slice c + 1)
                    else if (MODE == 3) __builtin_amdgcn_sched_group_barrier(0x020, (A_LD4 + B_LD4 + WR - 1) / WR, 0);                  // global loads (slice c + 2)
                }
            }
            __syncthreads();
            if (MODE >= 2 || (MODE == 0 && c + 1 < nchunks)) {
                const float* An = smem + nb * (A_TILE + B_TILE);
                read_half<TM, TN, NQ, A_KC, B_KC, LDA_S, LDB_S>(An, An + A_TILE, wm0, wn0, lane, 0, h0);
            }
            if (KL != 4) __builtin_amdgcn_sched_barrier(0);
            mma_hfrags<TM, TN, NQ>(h1, acc);
            if (KL == 4 && MODE >= 2) {
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                    if (q < NMF / 2) __builtin_amdgcn_sched_group_barrier(0x100, (NDR + NMF / 2 - 1) / (NMF / 2), 0);   // half 0 of slice c + 1
                }
            }
        };
        if (nchunks > 1) load_next(KlTag<0>{});
        read_half<TM, TN, NQ, A_KC, B_KC, LDA_S, LDB_S>(smem, smem + A_TILE, wm0, wn0, lane, 0, h0);
        const int full = (K - kb0) / BK;                                        // whole slices from kb0 on
        const int n_fast = nchunks - 2 < full - 2 ? nchunks - 2 : full - 2;     // iterations whose slice c + 2 exists and is whole
        int c = 0;
        for (; c < n_fast; ++c) body(c, KlTag<3>{});
        for (; c < nchunks - 2; ++c) body(c, KlTag<0>{});                       // (at most one: it loads the zero-filled tail slice)
        if (c == nchunks - 2) { body(c, KlTag<2>{}); ++c; }
        body(c, KlTag<1>{});
        return;
    }
#endif
    if (!PIPE) {
        // simple double buffer: fragments read and consumed inside one barrier interval
        Frags<TM, TN, BK> f;
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) { load_a(kb0 + (c + 1) * BK); load_b(kb0 + (c + 1) * BK); }
            const float* As = smem + buf * (A_TILE + B_TILE);
            read_frags<TM, TN, BK, A_KC, B_KC, LDA_S, LDB_S>(As, As + A_TILE, wm0, wn0, lane, f);
            mma_half<TM, TN, BK>(f, 0, acc);
            mma_half<TM, TN, BK>(f, 1, acc);
            if (c + 1 < nchunks) store_ab(buf ^ 1);
            __syncthreads();
        }
    } else {
        // software pipeline: the LDS store of slice c+1, the barrier and the fragment reads of slice
        // c+1 are issued between the two MFMA halves of slice c, whose operands already sit in
        // registers — the wave's MFMA stream never waits on LDS or HBM, only on barrier skew.
        Frags<TM, TN, BK> f0, f1;
        if (nchunks > 1) { load_a(kb0 + BK); load_b(kb0 + BK); }
        read_frags<TM, TN, BK, A_KC, B_KC, LDA_S, LDB_S>(smem, smem + A_TILE, wm0, wn0, lane, f0);
        auto step = [&](int c, const Frags<TM, TN, BK>& fc, Frags<TM, TN, BK>& fn) {
            const int nb = (c & 1) ^ 1;
            if (!(ABL & 2) && c + 1 < nchunks) store_ab(nb);
            if (!(ABL & 1) && c + 2 < nchunks) { load_a(kb0 + (c + 2) * BK); load_b(kb0 + (c + 2) * BK); }
            mma_half<TM, TN, BK>(fc, 0, acc);
            if (!(ABL & 8)) __syncthreads();
            if (!(ABL & 4) && c + 1 < nchunks) {
                const float* As = smem + nb * (A_TILE + B_TILE);
                read_frags<TM, TN, BK, A_KC, B_KC, LDA_S, LDB_S>(As, As + A_TILE, wm0, wn0, lane, fn);
            }
            mma_half<TM, TN, BK>(fc, 1, acc);
        };
        for (int c = 0; c < nchunks; c += 2) {
            step(c, f0, f1);
            if (c + 1 < nchunks) step(c + 1, f1, f0);
        }
    }

}

// LnFuse: the m-tile's last workgroup normalises its BM rows (see the struct).  Same publication protocol as slab_combine: write-through
// payload (gemm_epilogue), every storing wave drains, one relaxed agent-scope increment, ONE acquire on the last arriver, plain loads.
template <int BM, int NTH>
__device__ __forceinline__ void gemm_ln_tail(const GemmArgs& g, const GemmProb& pr, int z, int m0) {
    const LnFuse& f = g.ln;
    const int C = pr.N, tiles_n = (C + 63) / 64, tiles_m = (pr.M + BM - 1) / BM;
    // (counter slot: groups are laid out with the launch's common m-tile bound, which the engine passes through tiles_pg)
    int* ctr = f.ctr + (long long)z * g.tiles_pg + m0 / BM;
    (void)tiles_m;
    __shared__ int s_ln_last;
    MTTS_WAIT_VMEM();
    __syncthreads();
    if (threadIdx.x == 0) {
        s_ln_last = (MTTS_ATOMIC_INC_AGENT(ctr) == tiles_n - 1) ? 1 : 0;
        if (s_ln_last) MTTS_FENCE_ACQUIRE_AGENT();
    }
    __syncthreads();
    if (!s_ln_last) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NTH / 64, RPW = BM / NW;   // rows per wavefront
    const int c = lane * 4;
    const float* gm = f.gamma + (long long)z * f.par_gs;
    const float* bt = f.beta + (long long)z * f.par_gs;
    DropSpec din;
    din.seed = f.drop_seed; din.thr16 = f.drop_thr16; din.scale = f.drop_scale;
    for (int i = 0; i < RPW; ++i) {
        const int row = m0 + wave * RPW + i;
        if (row >= pr.M) break;
        float* pz = pr.C + (long long)row * pr.ldc;
        float4 x = zero4();
        if (c < C) {
            x = ld4(pz + c);
            if (din.thr16) x = drop4(din, z, row, C, c, x);
            if (f.res) { const float4 r4 = ld4(f.res + (long long)z * f.res_gs + (long long)row * C + c); x = make_float4(x.x + r4.x, x.y + r4.y, x.z + r4.z, x.w + r4.w); }
        }
        const float mean = wave_sum(c < C ? (x.x + x.y) + (x.z + x.w) : 0.f) / (float)C;
        const float dx = x.x - mean, dy = x.y - mean, dz = x.z - mean, dw = x.w - mean;
        const float rstd = rsqrtf(wave_sum(c < C ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f) / (float)C + f.eps);
        const bool keep = f.mask ? f.mask[(long long)z * f.mask_gs + row] != 0 : true;
        if (c < C) {
            st4(pz + c, x);
            float4 o = zero4();
            if (keep) {
                const float4 g4 = ld4(gm + c), b4 = ld4(bt + c);
                o = make_float4(dx * rstd * g4.x + b4.x, dy * rstd * g4.y + b4.y, dz * rstd * g4.z + b4.z, dw * rstd * g4.w + b4.w);
            }
            st4(f.y + (long long)z * f.y_gs + (long long)row * C + c, o);
        }
        if (lane == 0) {
            float* st = f.stats + (long long)z * f.st_gs + (long long)row * 2;
            st[0] = mean;
            st[1] = rstd;
        }
    }
    if (tid == 0) *ctr = 0;
}

// What follows the K-loop: the column sums of the extra n-tile, or the fused epilogue.
template <int TM, int TN, int WGM, int WGN, bool LNF = false>
__device__ __forceinline__ void gemm_finish(const GemmArgs& g, const GemmProb& pr, int z, int m0, int n0, bool cs_tile, f32x16 (&acc)[TM][TN]) {
    const int tid = MTTS_OPAQUE_TID(), lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * (32 * TM), wn0 = (wave % WGN) * (32 * TN);
    if (cs_tile) {  // column 0 of the extra n-tile = masked column sums of A
        if (wn0 == 0 && (lane & 31) == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < pr.M) g.colsum[(long long)z * g.colsum_gs + row] = acc[i][0][r];
                }
        }
        return;
    }
    gemm_epilogue<TM, TN, LNF>(g, z, acc, pr.C, pr.ldc, pr.M, pr.N, m0 + wm0, n0 + wn0, lane);
    if constexpr (LNF) { if (g.ln.ctr) gemm_ln_tail<32 * TM * WGM, 64 * WGM * WGN>(g, pr, z, m0); }
}

// One workgroup's share of one problem: output tile `bxs` (times split) of group `z`.
// DUAL: the kernel also runs the second source of dual-source problems (GemmArgs::A2) — a second, separately inlined K-loop into the
// same accumulators.  (A runtime loop over the sources around ONE inlined K-loop costs every kernel ~10 VGPRs of spilled SGPR state
// — the multi-problem BK = 32 kernel drops from 4 to 3 workgroups per CU — so only the kernels that may be handed such problems pay for it.)
template <int FORM, int BM, int BN, int BK, bool PIPE, int WGM = 2, int WGN = 2, int ABL = 0, bool DUAL = false, int KL = 0, bool LNF = false>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, int z, int bxs, float* smem) {
    constexpr int NTH = 64 * WGM * WGN;
    constexpr int TM = (BM / WGM) / 32, TN = (BN / WGN) / 32;
    if (g.wave_prio > 0) MTTS_SETPRIO_HIGH();
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
    const int c_lo = split * cps, c_hi = (c_lo + cps < nch_all) ? c_lo + cps : nch_all;   // (all chunks when S == 1)
    gemm_f32_kloop<FORM, BM, BN, BK, PIPE, WGM, WGN, ABL, KL>(g, pr, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
    if (DUAL) {
        if (g.A2 != nullptr && !cs_tile) {
            __syncthreads();   // every wave is done with the first source's LDS tiles
            const GemmProb p2 = gemm_resolve2(g, z, pr);
            gemm_f32_kloop<FORM, BM, BN, BK, PIPE, WGM, WGN, ABL, KL>(g, p2, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
        }
    }
    if (S > 1 && !splitk_combine<TM, TN, NTH>(g, z, tile_lin, split, S, acc)) return;
    gemm_finish<TM, TN, WGM, WGN, LNF>(g, pr, z, m0, n0, cs_tile, acc);
}

template <int FORM, int BM, int BN, int BK, bool PIPE, int WGM = 2, int WGN = 2, int ABL = 0, int KL = 0, bool LNF = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[GemmSmem<FORM, BM, BN, BK>::FLOATS];
    int z = blockIdx.z;
    int bxs = blockIdx.x;
    if (g.xs.on) { if (!xcd_sched_locate(g.xs, bxs, z, bxs)) return; }   // 1-D grid, task-per-XCD order
    else if (g.swizzle == 2) { if (!xcd_panel_locate(bxs, g.po_tiles_m, (g.N + BN - 1) / BN + (gemm_has_colsum<FORM>(g) ? 1 : 0), g.splitk > 1 ? g.splitk : 1, bxs)) return; }
    else if (g.swizzle) bxs = xcd_group_remap(bxs, (int)gridDim.x, xcd_group_size((g.N + BN - 1) / BN, g.splitk));
    gemm_f32_body<FORM, BM, BN, BK, PIPE, WGM, WGN, ABL, false, KL, LNF>(g, z, bxs, smem);
}

// Several independent problems (any mix of forms, e.g. the dgrad and the wgrad of one layer) in ONE launch: workgroups
// [start[p], start[p+1]) of every z-slice belong to problem p.  The chip then never drains between two under-filled or
// badly quantised grids — the tail of one problem is filled by the next — and it costs no stream / event traffic.
constexpr int kGemmMultiMax = 6;
struct GemmMulti {
    int n = 0;
    int start[kGemmMultiMax + 1] = {0};
    int form[kGemmMultiMax] = {0};
    int groups[kGemmMultiMax] = {0};
    int xcd_group[kGemmMultiMax] = {0};  // > 1: this many consecutive tile slots share an XCD (gemm_multi_locate); -2: GemmArgs::xs; -3: panel order (xcd_panel_locate)
    int tiles_pg[kGemmMultiMax] = {0};   // tile slots per group (start[p + 1] - start[p] may be padded up to a multiple of 8)
    short po_tm[kGemmMultiMax] = {0}, po_tn[kGemmMultiMax] = {0}, po_s[kGemmMultiMax] = {0};   // xcd_group == -3 (panel order): m-tiles, n-tiles, split
    GemmArgs g[kGemmMultiMax];
};

// Problem / group / tile of a workgroup of a multi-problem launch.  Problems are sorted by per-tile cost (longest first) and
// laid out problem-major over a 1-D grid, so the long tiles all start in the first dispatch round and the short ones fill
// the tail; within a problem the tile order is XCD-grouped (xcd_group_remap).
__device__ __forceinline__ bool gemm_multi_locate(const GemmMulti& mp, int& p, int& z, int& bx) {
    p = 0;
    int lin = (int)blockIdx.x;
    while (p + 1 < mp.n && lin >= mp.start[p + 1]) ++p;
    lin -= mp.start[p];
    if (mp.xcd_group[p] == -2) return xcd_sched_locate(mp.g[p].xs, lin, z, bx);   // (starts are multiples of 8 then)
    if (mp.xcd_group[p] == -3) {   // panel order, group after group (every group's slot run is a multiple of 8)
        const int per = xcd_panel_slots(mp.po_tm[p], mp.po_tn[p], mp.po_s[p]);
        z = lin / per;
        if (z >= mp.groups[p]) return false;
        return xcd_panel_locate(lin - z * per, mp.po_tm[p], mp.po_tn[p], mp.po_s[p], bx);
    }
    const int total = mp.tiles_pg[p] * mp.groups[p];
    if (lin >= total) return false;   // padding up to the next multiple of 8
    const int tiles = mp.tiles_pg[p];
    lin = xcd_group_remap(lin, total, mp.xcd_group[p]);
    z = lin / tiles;
    bx = lin - z * tiles;
    return true;
}
template <int BM, int BN, int BK, int KL = 0, bool LNF = false>
__global__ __launch_bounds__(256) void gemm_f32_multi_kernel(GemmMulti mp) {
    constexpr int F0 = GemmSmem<GEMM_NT, BM, BN, BK>::FLOATS, F1 = GemmSmem<GEMM_NN, BM, BN, BK>::FLOATS, F2 = GemmSmem<GEMM_TN, BM, BN, BK>::FLOATS;
    constexpr int FL = F0 > F1 ? (F0 > F2 ? F0 : F2) : (F1 > F2 ? F1 : F2);
    __shared__ __attribute__((aligned(16))) float smem[FL];
    int p, z, bx;
    if (!gemm_multi_locate(mp, p, z, bx)) return;
    const int form = mp.form[p];
    if (form == GEMM_NT) gemm_f32_body<GEMM_NT, BM, BN, BK, true, 2, 2, 0, false, KL, LNF>(mp.g[p], z, bx, smem);
    else if (form == GEMM_NN) gemm_f32_body<GEMM_NN, BM, BN, BK, true, 2, 2, 0, false, KL>(mp.g[p], z, bx, smem);
    else gemm_f32_body<GEMM_TN, BM, BN, BK, true, 2, 2, 0, false, KL>(mp.g[p], z, bx, smem);
}

// the same grid for launches that carry dual-source problems (GemmArgs::A2)
template <int BM, int BN, int BK, int KL = 0>
__global__ __launch_bounds__(256) void gemm_f32_multi_dual_kernel(GemmMulti mp) {
    constexpr int F0 = GemmSmem<GEMM_NT, BM, BN, BK>::FLOATS, F1 = GemmSmem<GEMM_NN, BM, BN, BK>::FLOATS, F2 = GemmSmem<GEMM_TN, BM, BN, BK>::FLOATS;
    constexpr int FL = F0 > F1 ? (F0 > F2 ? F0 : F2) : (F1 > F2 ? F1 : F2);
    __shared__ __attribute__((aligned(16))) float smem[FL];
    int p, z, bx;
    if (!gemm_multi_locate(mp, p, z, bx)) return;
    const int form = mp.form[p];
    if (form == GEMM_NT) gemm_f32_body<GEMM_NT, BM, BN, BK, true, 2, 2, 0, true, KL>(mp.g[p], z, bx, smem);
    else if (form == GEMM_NN) gemm_f32_body<GEMM_NN, BM, BN, BK, true, 2, 2, 0, true, KL>(mp.g[p], z, bx, smem);
    else gemm_f32_body<GEMM_TN, BM, BN, BK, true, 2, 2, 0, true, KL>(mp.g[p], z, bx, smem);
}

// Kernel kinds of the launcher: one per real kernel symbol, so that a profiler line can be matched to a rocprofv3 kernel-trace row.
enum GemmKind {
    GK_F32_64_BK16 = 0,   // + form: gemm_f32_kernel<F, 64, 64, 16, true, 2, 2, 0>
    GK_F32_64_BK32 = 3,   // + form: gemm_f32_kernel<F, 64, 64, 32, true, 2, 2, 0>
    GK_F32_128 = 6,       // + form: gemm_f32_kernel<F, 128, 128, *, *>
    GK_GLDS = 9,          // + form: gemm_glds_kernel<F>
    GK_MULTI16 = 12, GK_MULTI32 = 13, GK_GLDS_MULTI = 14, GK_OTHER = 15,
    GK_MULTI16_DUAL = 16, GK_MULTI32_DUAL = 17, GK_GLDS_MULTI_DUAL = 18,
    GK_BF16_64 = 19,      // + form: gemm_bf16_kernel<F, 64, 64, 32, PF>
    GK_BF16_128 = 22,     // + form: gemm_bf16_kernel<F, 128, 128, 32, PF>
    GK_BF16_MULTI64 = 25, GK_BF16_MULTI128 = 26, GK_BF16_MULTI64_DUAL = 27, GK_BF16_MULTI128_DUAL = 28,
    GK_ATTN_FWD = 29,     // attention.h: the fused QK^T -> softmax -> PV forward (both products' flops)
    GK_BF16_64_H64 = 30, GK_BF16_64_H32 = 31, GK_BF16_128_H = 32,   // gemm_bf16_kernel<GEMM_NT_H, ...>: NT staged from bf16 operand planes
    GK_BF16_MULTI64_PLANES = 33,                                      // gemm_bf16_multi_planes_kernel: a multi-problem launch that carries plane problems
    GK_COUNT = 34
};
inline const char* gemm_kind_name(int k) {
    static const char* names[GK_COUNT] = {
        "gemm_f32_kernel<0, 64, 64, 16, true, 2, 2, 0, 4>", "gemm_f32_kernel<1, 64, 64, 16, true, 2, 2, 0, 4>", "gemm_f32_kernel<2, 64, 64, 16, true, 2, 2, 0, 4>",
        "gemm_f32_kernel<0, 64, 64, 32, true, 2, 2, 0, 4>", "gemm_f32_kernel<1, 64, 64, 32, true, 2, 2, 0, 4>", "gemm_f32_kernel<2, 64, 64, 32, true, 2, 2, 0, 4>",
        "gemm_f32_kernel<0, 128, 128, ...>", "gemm_f32_kernel<1, 128, 128, ...>", "gemm_f32_kernel<2, 128, 128, ...>",
        "gemm_glds_kernel<0>", "gemm_glds_kernel<1>", "gemm_glds_kernel<2>",
        "gemm_f32_multi_kernel<64, 64, 16, 4>", "gemm_f32_multi_kernel<64, 64, 32, 4>", "gemm_glds_multi_kernel",   // (the trailing 4 = the default K-loop variant, MTTS_KLOOP)
        "gemm_f32_kernel<other>", "gemm_f32_multi_dual_kernel<64, 64, 16, 4>", "gemm_f32_multi_dual_kernel<64, 64, 32, 4>", "gemm_glds_multi_dual_kernel",
        "gemm_bf16_kernel<0, 64, 64, 32, 1>", "gemm_bf16_kernel<1, 64, 64, 32, 1>", "gemm_bf16_kernel<2, 64, 64, 32, 1>",
        "gemm_bf16_kernel<0, 128, 128, 32, 1>", "gemm_bf16_kernel<1, 128, 128, 32, 1>", "gemm_bf16_kernel<2, 128, 128, 32, 1>",
        "gemm_bf16_multi_kernel<64, 64, 32, 1, false>", "gemm_bf16_multi_kernel<128, 128, 32, 1, false>",
        "gemm_bf16_multi_kernel<64, 64, 32, 1, true>", "gemm_bf16_multi_kernel<128, 128, 32, 1, true>", "attn_fwd_kernel",
        "gemm_bf16_kernel<3, 64, 64, 64, 1>", "gemm_bf16_kernel<3, 64, 64, 32, 1>", "gemm_bf16_kernel<3, 128, 128, 32, 1>",
        "gemm_bf16_multi_planes_kernel<64, 64, 32, 64, 1>"};
    return (k >= 0 && k < GK_COUNT) ? names[k] : "?";
}

// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg):
// one record per launch, aggregated per kernel kind.
struct GemmProfiler {
    struct Rec { int kernel; double flops; hipEvent_t e0, e1; int form = 0, tile = 0, N = 0, K = 0, groups = 0, splitk = 1; double rows = 0, bytes = 0; int tag = 0; };
    int ctx = 0;   // which of the handle's launch contexts this is: 0 the main stream, 1 the weight-gradient side stream, 2 the run-ahead side stream (CSV column)
    int tag = 0;   // call-site class of the launches being recorded (set by the owner: 0 other, 1 encoder blocks, 2 decoder blocks, 3 PostNet, 4 variance predictors)
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    bool enabled = false;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
    void reset() { recs.clear(); used = 0; }
    void destroy() { for (hipEvent_t e : pool) hipEventDestroy(e); pool.clear(); recs.clear(); used = 0; }
    // out[kind][4] = launches, total ms, total algorithmic flops, total algorithmic bytes
    void report(double out[GK_COUNT][4]) {
        for (int k = 0; k < GK_COUNT; ++k) out[k][0] = out[k][1] = out[k][2] = out[k][3] = 0.0;
        FILE* dump = (getenv("MTTS_GEMM_DUMP") && !recs.empty()) ? fopen(getenv("MTTS_GEMM_DUMP"), "a") : nullptr;  // per-launch CSV (tools/gemm_sites.py); appended: a handle reports its three launch contexts one after the other
        if (dump && ftell(dump) == 0) fprintf(dump, "kind,form,tile,N,K,rows,groups,splitk,us,gflop,site,ctx\n");
        for (auto& r : recs) {
            hipEventSynchronize(r.e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, r.e0, r.e1);
            out[r.kernel][0] += 1.0; out[r.kernel][1] += ms; out[r.kernel][2] += r.flops; out[r.kernel][3] += r.bytes;
            if (dump) fprintf(dump, "%d,%d,%d,%d,%d,%.0f,%d,%d,%.2f,%.4f,%d,%d\n", r.kernel, r.form, r.tile, r.N, r.K, r.rows, r.groups, r.splitk, 1e3 * ms, r.flops * 1e-9, r.tag, ctx);
        }
        if (dump) fclose(dump);
    }
};
constexpr int kGemmXcdSwizzle = 1;     // XCD-grouped tile order of the plain grids (xcd_group_remap)
constexpr int kGemmDefaultBk = 16;     // K-slice of the register-staged kernels (32 for long K-contiguous panels, see gemm_launch)
constexpr bool kGemmDefaultPipe = true;  // software-pipelined K-loop
// K-loop variant of the pipelined 64x64 kernels (gemm_f32_kloop: KL).  MTTS_KLOOP=0 / 1 / 4 picks one for A/B runs
// (profiles/r05_kloop_ab.md); results are bit-identical across variants (same k order in every accumulator chain).
constexpr int kGemmDefaultKloop = 4;
inline int gemm_kloop_variant() {
    static const int v = [] { const char* e = getenv("MTTS_KLOOP"); const int x = e ? atoi(e) : kGemmDefaultKloop; return (x == 0 || x == 1 || x == 4) ? x : kGemmDefaultKloop; }();
    return v;
}
// The KL >= 1 loops step their operand pointers by loop-invariant distances, which needs conv taps that are whole multiples of the K-slice
// (every channel count of the model is a multiple of 16; anything else keeps the generic KL = 0 loop)
inline int gemm_kloop_for(const GemmArgs& g, int bk) {
    const bool ok = (g.taps <= 1 || g.tap_k % bk == 0) && (g.a_tap_k >= 0x40000000 || g.a_tap_k % bk == 0);
    return ok ? gemm_kloop_variant() : 0;
}
inline int gemm_xcd_swizzle() { return kGemmXcdSwizzle; }
inline int gemm_default_bk() { return kGemmDefaultBk; }
inline bool gemm_default_pipe() { return kGemmDefaultPipe; }

// Split-K workspace (partial tiles + tile counters): owned by a GemmCtx, allocated once by its owner's create.
struct GemmWorkspace { float* ws = nullptr; int* ctr = nullptr; };
constexpr long long kSplitWsFloats = 16ll << 20;  // 64 MB of partial tiles
constexpr int kSplitCtrs = 1 << 16;
constexpr int kLnCtrs = 4096;                     // the last kLnCtrs counters: arrival counters of row-complete LayerNorm epilogues (LnFuse)

// Launch batching: between gemm_batch_begin() and gemm_batch_end() every eligible gemm_launch (automatic tile choice) is queued
// instead of launched; gemm_batch_end() issues the queue as ONE launch (gemm_f32_multi_kernel / gemm_glds_multi_kernel).
// The caller guarantees the queued problems are mutually independent and that nothing launched before gemm_batch_end()
// reads their outputs (engine: the wgrad / dgrad pair of a layer, dQ / dK / dV of an attention block).
struct GemmPending { int form; GemmArgs g; int max_M, max_N, groups; double flops, rows, bytes; };
struct GemmBatch { bool open = false; std::vector<GemmPending> q; int force_family = 0; int force_tile = 0; };   // force_family: 16 / 32 (BK of the register-staged multi-problem kernel) or 4064 (LDS-DMA) for the next flush (explicit tile codes of dual-source problems)

// Every piece of MUTABLE launcher state — the launch-batching queue, the per-launch profiler, the split-K workspace —
// lives in a context owned by one handle (Engine / Vocoder / ...), so two handles on two host threads share nothing
// (include/mtts.h conventions).  The workspace is allocated by the owner's create, never lazily; a context without one (the
// handle-less kernel entry points) never splits K.  One context = one stream (the split-K slabs / counters of consecutive
// launches are reused in stream order).
struct GemmCtx {
    GemmBatch batch;
    GemmProfiler prof;
    GemmWorkspace wsp;
    int last_kind = GK_OTHER;  // kernel kind of the last launch (profiler)
    long long plane_problems = 0;   // problems launched on the plane-staged K-loop (bf16 mode; a test / diagnostic counter)
    bool bf16 = false;         // numerics mode of the owner: bf16 operand family (gemm_bf16.h) for every problem it can take
    const char* error = nullptr;   // sticky: a launch the launcher refused (the C ABI entry points turn it into an error return)
    bool flushing = false;     // gemm_batch_end is issuing the queue
    bool prefer_bk16 = false;   // multi-problem launches of this context take the BK = 16 kernels (20 KB of LDS per workgroup instead of 37) whatever their K
    int wave_prio = 0;      // GemmArgs::wave_prio of every launch of this context (the engine sets it on the critical stream's context while side streams carry work)
    bool no_glds = false;   // never pick the LDS-DMA kernels (48 KB of LDS per workgroup: a side-stream launch would leave no LDS for the main stream's)
    int alloc_workspace() {
        if (wsp.ws) return 0;
        if (hipMalloc((void**)&wsp.ws, kSplitWsFloats * sizeof(float)) != hipSuccess || hipMalloc((void**)&wsp.ctr, (kSplitCtrs + 16) * sizeof(int)) != hipSuccess ||
            hipMemset(wsp.ctr, 0, (kSplitCtrs + 16) * sizeof(int)) != hipSuccess) { release(); return -1; }
        return 0;
    }
    void release() {
        if (wsp.ws) hipFree(wsp.ws);
        if (wsp.ctr) hipFree(wsp.ctr);
        wsp.ws = nullptr; wsp.ctr = nullptr;
        prof.destroy();
    }
};
// LnFuse (requested by the caller through g.ln.y): can this launch carry it, and its counter slice.  `used` = counters already handed out
// inside the same launch.  Returns false (and sets cx.error) when the request cannot be honoured — the engine asks gemm_ln_fusable first.
inline bool gemm_ln_fusable(const GemmCtx& cx, int form, const GemmArgs& g, int max_M, int groups) {
    return form == GEMM_NT && !cx.bf16 && cx.wsp.ctr != nullptr && gemm_kloop_variant() == 4 && g.N <= 256 && g.N % 4 == 0 && !g.table && !g.c_rowmap && !g.A2 &&
           !(g.flags & GEMM_ACCUM) && !g.colsum && (long long)groups * ((max_M + 63) / 64) <= kLnCtrs;
}
inline bool gemm_ln_bind(GemmCtx& cx, int form, GemmArgs& g, int max_M, int groups, int tile, int& used) {
    if (!g.ln.y) { g.ln.ctr = nullptr; return true; }
    const int tiles_m = (max_M + 63) / 64;
    if (tile != 64 || !gemm_ln_fusable(cx, form, g, max_M, groups) || used + groups * tiles_m > kLnCtrs) {
        cx.error = "row-complete LayerNorm epilogue requested for a launch that cannot carry it";
        return false;
    }
    g.ln.ctr = cx.wsp.ctr + (kSplitCtrs - kLnCtrs) + used;
    g.tiles_pg = tiles_m;   // (counter stride between groups; the problem is never split, so the split-K meaning of the field is free)
    g.splitk = 1;
    used += groups * tiles_m;
    return true;
}
inline void gemm_batch_begin(GemmCtx& cx) { cx.batch.open = true; }
inline void gemm_batch_end(GemmCtx& cx, hipStream_t stream);

// LDS-DMA kernel family (gemm_glds.h, device builds only)
// LDS-DMA family (gemm_glds.h).  Round 5: with the interleaved K-loop (KL = 4) the register-staged kernels beat it in the latency regime it
// was built for (single-task rank 33.5 -> 32.0 ms first order, 83.9 -> 79.3 ms second order; the 8-task step is indifferent: 158.0 vs
// 157.4 ms, profiles/r05_kloop_ab.md), so the automatic tile choice no longer takes it.  MTTS_GLDS=1 restores the round 3-4 rule (launches
// of <= 768 workgroups) for A/B runs — GemmCtx::no_glds (engine.h: set_regime) can then still veto it per pass; tile code 4064 selects
// the family explicitly (kernel tests, micro-benchmarks).
inline int gemm_glds_mode() {
    static const int m = [] { const char* e = getenv("MTTS_GLDS"); return e ? (atoi(e) != 0 ? 1 : 0) : 0; }();
    return m;
}
inline bool gemm_use_glds() { return gemm_glds_mode() != 0; }
inline bool gemm_glds_ok(const GemmArgs& g) { return !(g.taps > 1 && g.tap_k % 32 != 0); }
// n-tiles of a problem's grid: the tiles of C plus the column-sum tile (GemmArgs::colsum)
inline int gemm_tiles_n(const GemmArgs& g, int max_N, int t) { return (max_N + t - 1) / t + (g.colsum ? 1 : 0); }
// Launches up to this many workgroups take the LDS-DMA kernels (latency regime: few workgroups per CU, where the DMA
// ring's two slices in flight replace the occupancy the register-staged kernel needs; measured +8 % / +14 % on the
// single-task first- / second-order step), larger ones the register-staged kernels (5 vs 3 workgroups per CU resident:
// +3 % on the full 8-task step).
inline long gemm_glds_max_wgs() { return 768L; }
#if !defined(MTTS_EMU)
inline void gemm_glds_launch(int form, const GemmArgs& g, dim3 grid, hipStream_t stream);
inline void gemm_glds_multi_launch(const GemmMulti& mp, dim3 grid, hipStream_t stream, bool dual = false);
#endif

// bf16 operand family (gemm_bf16.h)
inline bool gemm_bf16_ok(const GemmArgs& g);
inline bool gemm_bf16_planes_ok(int form, const GemmArgs& g);
inline int gemm_bf16_launch(int form, const GemmArgs& g, int T, dim3 grid, hipStream_t stream, int pf);      // returns the GemmKind launched
inline int gemm_bf16_multi_launch(const GemmMulti& mp, int T, bool dual, dim3 grid, hipStream_t stream);
// block tile of a bf16 launch: 128x128 once the launch still fills the chip twice over with it (the MFMA rate is 16x the fp32 kernels':
// a 64x64 tile moves 1 byte per 16 flop through the L2 -> LDS path), 64x64 for under-filled launches
inline int gemm_bf16_tile(double rows, int tiles_n128) { return std::ceil(rows / 128.0) * tiles_n128 >= 512.0 ? 128 : 64; }

// Task-per-XCD schedule (XcdSched) of a problem: exactly 8 groups whose sizes the caller knows on the host, whole tiles.  MTTS_XCD_SCHED=0: off.
inline bool gemm_xcd_sched_for(GemmArgs& g, int max_M, int max_N, int groups, int S, int tile) {
    static const bool on = [] { const char* e = getenv("MTTS_XCD_SCHED"); return e ? atoi(e) != 0 : true; }();
    g.xs.on = 0;
    static const int min_groups = [] { const char* e = getenv("MTTS_XCD_SCHED_MIN_GROUPS"); return e ? atoi(e) : 8; }();   // 2 / 4: also the launches of a 4- / 2-rank job's ranks
    if (!on || (groups != 8 && groups != 4 && groups != 2) || groups < min_groups || !g.host_dims || !g.dimptr || g.table || S != 1) return false;
    int dims[8] = {0};
    for (int z = 0; z < groups; ++z) dims[z] = g.host_dims[z] * g.dim_mult;
    const int tn = gemm_tiles_n(g, max_N, tile);
    if (g.dim_sel == 0) xcd_sched_build(g.xs, dims, 1, tn, 0, tile, groups);
    else xcd_sched_build(g.xs, dims, 2, 1, ((max_M + tile - 1) / tile) * tn, 64, groups);
    if (g.xs.maxlen <= 0) g.xs.on = 0;
    static const bool dbg = getenv("MTTS_XCD_SCHED_DEBUG") != nullptr;   // one line per scheduled problem (tests: the schedule really is in use)
    if (dbg && g.xs.on) fprintf(stderr, "xcd_sched cls %d tn %d maxlen %d pool %d\n", g.xs.on, g.xs.tn, g.xs.maxlen, g.xs.P[8]);
    return g.xs.on != 0;
}
inline long gemm_xcd_sched_slots(const XcdSched& s) { return 8L * s.maxlen * (s.on == 1 ? s.tn : 1); }
// Panel order (xcd_panel_locate) for a problem: TASK mode, fewer than 8 groups (the task-per-XCD schedule covers 8), an under-filled
// launch, and a B operand (N x K: the weight image of a forward / input-gradient problem) that is the bigger of the two — otherwise the
// m-tile-major grouping, which keeps an m-tile's A panel in one L2, is already the right one.  Measured (profiles/r04_ab_log.md): the
// k = 9 input gradient of a single-task rank 212 -> 194 us, its step 35.17 -> 34.73 ms.  MTTS_PANEL_ORDER=0: off (A/B runs).
inline bool gemm_panel_order_for(const GemmArgs& g, int form, double rows, int max_M, int max_N, int groups) {
    static const bool on = [] { const char* e = getenv("MTTS_PANEL_ORDER"); return e ? atoi(e) != 0 : true; }();
    if (!on || g.table || groups >= 8 || groups < 1) return false;
    if (std::ceil(rows / 64.0) * gemm_tiles_n(g, max_N, 64) > (double)gemm_glds_max_wgs()) return false;   // chip-filling launches: measured slightly worse (C2: +1 %)
    const double K = (double)gemm_keff(g);
    const double a_bytes = form == GEMM_TN ? rows * max_M : rows / groups * (g.lda > 0 ? g.lda : K);   // unique bytes behind the A operand (per group)
    const double b_bytes = form == GEMM_TN ? rows * max_N : (double)max_N * K;
    (void)max_M;
    return b_bytes > a_bytes;
}

// Host launcher.  max_M / max_N bound the tile grid over all groups.  tile = 0: automatic — the problem goes through the launch
// queue (alone if no batch is open): a plain 64x64 grid (LDS-DMA kernels in the latency regime, a multi-problem grid for a batch).
// An explicit tile code picks one kernel: 64 / 128 (+1000 software pipeline, +2000 BK = 32), 4064 LDS-DMA (kernel tests, micro-benchmarks).
// total_M = sum of the groups' row counts (0: max_M * groups).  alg_flops / alg_bytes: algorithmic (unpadded) work of this launch,
// profiler only.
inline void gemm_launch(GemmCtx& cx, int form, const GemmArgs& g_in, int max_M, int max_N, int groups, hipStream_t stream,
                        int tile = 0, double alg_flops = 0.0, long long total_M = 0, double alg_bytes = 0.0) {
    if (max_M <= 0 || max_N <= 0 || groups <= 0) return;
    GemmArgs g = g_in;
    g.swizzle = gemm_xcd_swizzle();
    g.wave_prio = cx.wave_prio;
    // bf16 planes: kept only where the plane-staged K-loop will run (bf16 mode, an NT problem it can take); the twin of C in bf16 mode only
    if (!(cx.bf16 && gemm_bf16_ok(g) && gemm_bf16_planes_ok(form, g))) {
        if (g.plane_only) { cx.error = "plane-only GEMM problem outside the bf16 plane path"; return; }
        g.Ah = g.Bh = nullptr;
    }
    if (!cx.bf16) g.Ch = nullptr;
    const int user_tile = tile;
    const double rows = total_M > 0 ? (double)total_M : (double)max_M * groups;
    auto ntiles = [&](int t) { return (long)((max_M + t - 1) / t) * gemm_tiles_n(g, max_N, t); };
    if (g.A2 && user_tile != 0 && !cx.flushing && !cx.batch.open) {
        // dual-source problems exist in the multi-problem kernels only: an explicit tile code picks the family (kernel tests)
        const int code = user_tile % 10000;
        cx.batch.force_family = code == 4064 ? 4064 : (code / 2000 ? 32 : 16);
        cx.batch.force_tile = code % 1000 == 128 ? 128 : 64;   // (bf16 mode: the block tile of the forced launch)
        cx.batch.q.push_back(GemmPending{form, g, max_M, max_N, groups, alg_flops, rows, alg_bytes});
        gemm_batch_end(cx, stream);
        cx.batch.force_family = 0; cx.batch.force_tile = 0;
        return;
    }
    if (user_tile == 0) {
        if (!cx.flushing) {
            // through the queue: with the batch's other problems, or alone
            cx.batch.q.push_back(GemmPending{form, g, max_M, max_N, groups, alg_flops, rows, alg_bytes});
            if (!cx.batch.open) {
                gemm_batch_end(cx, stream);
            } else if ((int)cx.batch.q.size() == kGemmMultiMax) { gemm_batch_end(cx, stream); cx.batch.open = true; }
            return;
        }
        tile = 64;
    }
    if (g.A2) {   // (unreachable from the engine: gemm_batch_end never sends a dual-source problem to the stand-alone kernels)
        cx.error = "dual-source GEMM handed to a stand-alone kernel (explicit tile code inside an open batch)";   // reported by the C ABI entry
        return;
    }
    // tile code: 64 / 128 (+1000 software pipeline, +2000 BK=32); plain 64 / 128 take the defaults
    bool pipe = gemm_default_pipe();
    int bk = gemm_default_bk();
    bool glds = false;
    const bool bf16 = cx.bf16 && gemm_bf16_ok(g) && !g.A2;
    if (bf16 && user_tile == 0) tile = gemm_bf16_tile(rows, gemm_tiles_n(g, max_N, 128));
#if !defined(MTTS_EMU)
    if (bf16) { if (tile == 4064) tile = 64; }
    else if (tile == 4064) { glds = gemm_glds_ok(g); tile = 64; }            // explicit request (kernel tests, microbenchmarks)
    else if (user_tile == 0 && tile == 64) {
        const long wgs = (long)std::ceil(rows / 64.0) * gemm_tiles_n(g, max_N, 64);
        glds = gemm_use_glds() && gemm_glds_ok(g) && wgs <= gemm_glds_max_wgs() && !cx.no_glds && !g.ln.y;
    }
#else
    if (tile == 4064) tile = 64;
#endif
    const int ablate = tile / 10000;  // diagnostic stage ablation (NT 64x64 pipelined BK=16 only), see gemm_f32_kloop
    tile %= 10000;
    if (tile >= 1000) { pipe = (tile / 1000) & 1; bk = (tile / 2000) ? 32 : 16; tile %= 1000; }
    if (user_tile == 0 && !g.table && gemm_keff(g) >= 1024 && (form == GEMM_NT || form == GEMM_TN)) bk = 32;  // long K-contiguous panels: full 128-B lines per row
    if (g.taps > 1 && g.tap_k % 32 != 0) bk = 16;  // a K-slice must not straddle two conv taps
    if (bf16) bk = 32;
    // split-K for under-filled grids (single-task ranks, the phoneme-side GEMMs, small wgrads): enough workgroups for
    // ~4 per CU, each still reducing >= 4 K-chunks
    const int S = 1;   // (stand-alone launches never split K: the long chains of under-filled batches are cut in gemm_batch_end)
    { int used = 0; if (!gemm_ln_bind(cx, form, g, max_M, groups, bf16 ? 0 : tile, used)) return; }
    const int nth = 256;
    const long grid_tiles = ntiles(tile);
    dim3 block(nth), grid((unsigned)(grid_tiles * S), 1, (unsigned)groups);
    if (gemm_xcd_sched_for(g, max_M, max_N, groups, S, tile)) grid = dim3((unsigned)gemm_xcd_sched_slots(g.xs), 1, 1);   // 1-D, task-per-XCD order
    else if (gemm_panel_order_for(g, form, rows, max_M, max_N, groups)) {
        g.swizzle = 2; g.po_tiles_m = (max_M + tile - 1) / tile;
        grid = dim3((unsigned)xcd_panel_slots(g.po_tiles_m, gemm_tiles_n(g, max_N, tile), S), 1, (unsigned)groups);
    }
    GemmProfiler& prof = cx.prof;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof.enabled) { e0 = prof.get(); e1 = prof.get(); hipEventRecord(e0, stream); }
    int kind = GK_OTHER;
#define MTTS_GEMM_CASE(F, T)                                                                              \
    if (form == F && tile == T) {                                                                         \
        if (pipe && T == 64) {                                                                            \
            const int kl = gemm_kloop_for(g, bk);                                                         \
            if (F == GEMM_NT && g.ln.ctr) {   /* row-complete LayerNorm epilogue (LnFuse): instantiations of their own */    \
                if (bk == 32) MTTS_LAUNCH((gemm_f32_kernel<GEMM_NT, 64, 64, 32, true, 2, 2, 0, 4, true>), grid, block, stream, g);   \
                else MTTS_LAUNCH((gemm_f32_kernel<GEMM_NT, 64, 64, 16, true, 2, 2, 0, 4, true>), grid, block, stream, g);            \
                kind = (bk == 32 ? GK_F32_64_BK32 : GK_F32_64_BK16) + F;                                  \
            } else if (bk == 32) {                                                                        \
                if (kl == 4) MTTS_LAUNCH((gemm_f32_kernel<F, 64, 64, 32, true, 2, 2, 0, 4>), grid, block, stream, g);        \
                else if (kl == 1) MTTS_LAUNCH((gemm_f32_kernel<F, 64, 64, 32, true, 2, 2, 0, 1>), grid, block, stream, g);   \
                else MTTS_LAUNCH((gemm_f32_kernel<F, 64, 64, 32, true, 2, 2, 0, 0>), grid, block, stream, g);                \
                kind = GK_F32_64_BK32 + F;                                                                \
            } else {                                                                                      \
                if (kl == 4) MTTS_LAUNCH((gemm_f32_kernel<F, 64, 64, 16, true, 2, 2, 0, 4>), grid, block, stream, g);        \
                else MTTS_LAUNCH((gemm_f32_kernel<F, 64, 64, 16, true, 2, 2, 0, 0>), grid, block, stream, g);                \
                kind = GK_F32_64_BK16 + F;                                                                \
            }                                                                                             \
        }                                                                                                 \
        else if (bk == 32 && pipe) { MTTS_LAUNCH((gemm_f32_kernel<F, T, T, 32, true>), grid, block, stream, g); kind = (T == 64 ? GK_F32_64_BK32 : GK_F32_128) + F; }   \
        else if (bk == 32) { MTTS_LAUNCH((gemm_f32_kernel<F, T, T, 32, false>), grid, block, stream, g); kind = T == 64 ? GK_OTHER : GK_F32_128 + F; }     \
        else if (pipe) { MTTS_LAUNCH((gemm_f32_kernel<F, T, T, 16, true>), grid, block, stream, g); kind = (T == 64 ? GK_F32_64_BK16 : GK_F32_128) + F; }          \
        else { MTTS_LAUNCH((gemm_f32_kernel<F, T, T, 16, false>), grid, block, stream, g); kind = T == 64 ? GK_OTHER : GK_F32_128 + F; }                   \
    }
    if (bf16) {
        if (g.Ah) ++cx.plane_problems;
        kind = gemm_bf16_launch(form, g, tile, grid, stream, (user_tile % 10000) / 1000);   // (bf16 tile codes: T + 1000 * slices in flight, 0 = default)
    } else
#if !defined(MTTS_EMU)
    if (glds) {
        gemm_glds_launch(form, g, grid, stream);
        kind = GK_GLDS + form;
    } else
#endif
    {
#if defined(MTTS_GEMM_ABLATION)
#define MTTS_ABL(A) if (ablate == A && form == GEMM_NT) { MTTS_LAUNCH((gemm_f32_kernel<GEMM_NT, 64, 64, 16, true, 2, 2, A>), grid, block, stream, g); return; }
        MTTS_ABL(1) MTTS_ABL(2) MTTS_ABL(3) MTTS_ABL(4) MTTS_ABL(7) MTTS_ABL(8) MTTS_ABL(15) MTTS_ABL(6) MTTS_ABL(14)
#undef MTTS_ABL
#else
        (void)ablate;
#endif
        MTTS_GEMM_CASE(GEMM_NT, 128) MTTS_GEMM_CASE(GEMM_NT, 64)
        MTTS_GEMM_CASE(GEMM_NN, 128) MTTS_GEMM_CASE(GEMM_NN, 64)
        MTTS_GEMM_CASE(GEMM_TN, 128) MTTS_GEMM_CASE(GEMM_TN, 64)
    }
#undef MTTS_GEMM_CASE
    cx.last_kind = kind;
    if (prof.enabled) {
        hipEventRecord(e1, stream);
        GemmProfiler::Rec rec{kind, alg_flops, e0, e1};
        rec.form = form; rec.tile = bf16 ? 16000 + tile : (glds ? 4064 : tile); rec.N = max_N; rec.K = gemm_keff(g); rec.groups = groups; rec.splitk = S;
        rec.rows = rows;
        rec.bytes = alg_bytes; rec.tag = prof.tag;
        prof.recs.push_back(rec);
    }
}

inline bool batch_full_regime(const std::vector<GemmPending>& q) {  // more workgroups than the latency regime's limit
    double wgs = 0.0;
    for (const GemmPending& p : q) wgs += std::ceil(p.rows / 64.0) * gemm_tiles_n(p.g, p.max_N, 64);
    return wgs > (double)gemm_glds_max_wgs();
}
inline void gemm_batch_end(GemmCtx& cx, hipStream_t stream) {
    GemmBatch& b = cx.batch;
    b.open = false;
    if (b.q.empty()) return;
    struct Flush { GemmCtx& c; Flush(GemmCtx& x) : c(x) { c.flushing = true; } ~Flush() { c.flushing = false; } } guard(cx);
    std::stable_sort(b.q.begin(), b.q.end(), [](const GemmPending& x, const GemmPending& y) { return gemm_keff(x.g) > gemm_keff(y.g); });
    GemmProfiler& prof = cx.prof;
    // a single queued problem normally takes the stand-alone launcher; in the latency regime it stays here, where the long-chain
    // split-K rule applies (the k=9 dgrad of a single-task rank is 124 tiles x 288 slices: alone it would run at one tile per CU)
    static const int single_multi = [] { const char* e = getenv("MTTS_SINGLE_MULTI"); return e ? atoi(e) : 1; }();
    bool solo = b.q.size() == 1 && (!single_multi || batch_full_regime(b.q));
    bool any_dual = false;   // dual-source problems (GemmArgs::A2) exist in the multi-problem kernels only
    for (const GemmPending& p : b.q) any_dual = any_dual || p.g.A2 != nullptr;
    if (any_dual) solo = false;
    if (solo) {
        const std::vector<GemmPending> q = b.q;
        b.q.clear();
        for (const GemmPending& p : q) gemm_launch(cx, p.form, p.g, p.max_M, p.max_N, p.groups, stream, 0, p.flops, (long long)p.rows, p.bytes);
        return;
    }
    GemmMulti mp;
    mp.n = (int)b.q.size();
    int max_groups = 0;
    double flops = 0.0, rows = 0.0, bytes = 0.0;
    int max_split = 1;   // (profiler record: the largest split factor among the launch's problems)
    // split-K for the long chains of an under-filled batch: a tile whose K-loop is longer than two thirds (swept: 1 / 1.25 .. 1 / 1.5)
    // of the whole batch's per-CU work would finish last on its own (single-task ranks: the k=9 dgrad tile, 576 slices, beside a
    // batch that is worth ~590 slices per CU), so it is cut into S workgroups (rendezvous in splitk_combine)
    double work = 0.0;
    for (const GemmPending& p : b.q) work += std::ceil(p.rows / 64.0) * gemm_tiles_n(p.g, p.max_N, 64) * std::max(1, (gemm_keff(p.g) + 15) / 16);
    const double per_cu = work / 256.0;
    double batch_wgs = 0.0, batch_wgs128 = 0.0;
    for (const GemmPending& p : b.q) {
        batch_wgs += std::ceil(p.rows / 64.0) * gemm_tiles_n(p.g, p.max_N, 64);
        batch_wgs128 += std::ceil(p.rows / 128.0) * gemm_tiles_n(p.g, p.max_N, 128);
    }
    const bool small_batch = batch_wgs <= (double)gemm_glds_max_wgs();  // latency regime: split-K and the LDS-DMA kernels apply
    // bf16 numerics mode: the whole launch takes the bf16 operand family when every problem qualifies (gemm_bf16_ok); block tile T
    bool bf16 = cx.bf16;
    for (const GemmPending& p : b.q) bf16 = bf16 && gemm_bf16_ok(p.g);
    if (!bf16) {
        // A batch that mixes plane-only problems (bf16 mode: the input-gradient conv over the weight's transposed shadow has no fp32 B operand)
        // with problems the bf16 kernels cannot take (a dual-source tangent product whose tap length is not a multiple of 32: the
        // Hessian-vector pass through the PostNet's output layer) goes out as two launches: the plane-only problems on the bf16 family, the
        // rest on the fp32 family.  The problems of a batch are independent, so the split changes nothing but the launch count.
        std::vector<GemmPending> planes, rest;
        for (const GemmPending& p : b.q) (p.g.plane_only ? planes : rest).push_back(p);
        if (!planes.empty()) {
            bool ok = cx.bf16 && !rest.empty();
            for (const GemmPending& p : planes) ok = ok && gemm_bf16_ok(p.g);
            if (!ok) { cx.error = "plane-only GEMM problem queued outside the bf16 plane path"; b.q.clear(); return; }
            const int ff = b.force_family, ft = b.force_tile;
            b.q = planes; b.force_family = 0;
            gemm_batch_end(cx, stream);
            b.q = rest; b.force_family = ff; b.force_tile = ft;
            gemm_batch_end(cx, stream);
            return;
        }
    }
    const int T = !bf16 ? 64 : (b.force_tile ? b.force_tile : (batch_wgs128 >= 512.0 ? 128 : 64));
    GemmWorkspace* wsp = nullptr;
    long long ws_off = 0, ctr_off = 0;
    int ln_used = 0;
    for (int i = 0; i < mp.n; ++i) {
        const GemmPending& p = b.q[i];
        mp.form[i] = p.form; mp.groups[i] = p.groups; mp.g[i] = p.g;
        mp.g[i].swizzle = 0; mp.g[i].splitk = 1; mp.g[i].wave_prio = cx.wave_prio;
        const int tiles = ((p.max_M + T - 1) / T) * gemm_tiles_n(p.g, p.max_N, T);
        int S = 1;
        const int nch = (gemm_keff(p.g) + 15) / 16;
        constexpr double ratio = 1.5;
        static const int split_on = [] { const char* e = getenv("MTTS_BATCH_SPLITK"); return e ? atoi(e) : 1; }();   // 0: never cut K (an arm of tools/so_tolerance_bisect.py)
        if (split_on && small_batch && T == 64 && !p.g.table && !p.g.colsum && !p.g.ln.y && nch > per_cu / ratio) {
            S = (int)std::min<double>(std::min<double>(std::ceil(nch / std::max(per_cu / ratio, 1.0)), nch / 16), 8);
            const long long slots = (long long)tiles * p.groups;
            if (S >= 2 && ((ws_off + slots * S * 4096) > kSplitWsFloats || ctr_off + slots > kSplitCtrs - kLnCtrs)) S = 1;
            if (S >= 2) {
                if (!wsp) wsp = &cx.wsp;
                if (wsp->ws) {
                    mp.g[i].splitk = S; mp.g[i].tiles_pg = tiles; mp.g[i].ws = wsp->ws + ws_off; mp.g[i].tile_ctr = wsp->ctr + ctr_off;
                    max_split = std::max(max_split, S);
                    ws_off += slots * S * 4096; ctr_off += slots;
                } else S = 1;
            } else S = 1;
        }
        if (!gemm_ln_bind(cx, p.form, mp.g[i], p.max_M, p.groups, bf16 ? 0 : T, ln_used)) { b.q.clear(); return; }
        mp.xcd_group[i] = gemm_xcd_swizzle() ? std::min(gemm_tiles_n(p.g, p.max_N, T) * S, 64) : 0;
        mp.tiles_pg[i] = tiles * S;
        long slots = (long)tiles * S * p.groups;
        if (gemm_xcd_sched_for(mp.g[i], p.max_M, p.max_N, p.groups, S, T)) { mp.xcd_group[i] = -2; slots = gemm_xcd_sched_slots(mp.g[i].xs); }
        else if (gemm_panel_order_for(p.g, p.form, p.rows, p.max_M, p.max_N, p.groups)) {
            mp.xcd_group[i] = -3;
            mp.po_tm[i] = (short)((p.max_M + T - 1) / T); mp.po_tn[i] = (short)gemm_tiles_n(p.g, p.max_N, T); mp.po_s[i] = (short)S;
            slots = (long)xcd_panel_slots(mp.po_tm[i], mp.po_tn[i], S) * p.groups;
        }
        mp.start[i + 1] = mp.start[i] + (int)((slots + 7) & ~7L);   // (every problem starts on a multiple of 8: workgroup slot % 8 = XCD)
        max_groups = std::max(max_groups, p.groups);
        flops += p.flops; rows += p.rows; bytes += p.bytes;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof.enabled) { e0 = prof.get(); e1 = prof.get(); hipEventRecord(e0, stream); }
    dim3 block(256), grid((unsigned)mp.start[mp.n], 1, 1);
    // BK = 32 stages K-contiguous operands in full 128-byte lines (half the load instructions, TA transactions and barriers
    // per flop); needs every tapped problem's tap length to be a multiple of 32 and pays off only for long K.  Default since round 3
    // (profiles/r03_sk_queue.md: the multi-problem launches with K >= 1024 run at 0.676 instead of 0.646 of the fp32 matrix peak, the
    // whole step is unchanged, a single-task rank gains 1.7 %).
    bool bk32 = true;
    int maxK = 0;
    for (int i = 0; i < mp.n; ++i) {
        if (mp.g[i].taps > 1 && mp.g[i].tap_k % 32 != 0) bk32 = false;
        maxK = std::max(maxK, gemm_keff(mp.g[i]));
    }
    bool glds = gemm_use_glds() && small_batch && !cx.no_glds && ln_used == 0;
    if (b.force_family) { glds = b.force_family == 4064; bk32 = bk32 && b.force_family == 32; if (bk32) maxK = std::max(maxK, 1024); }
    for (int i = 0; i < mp.n; ++i) glds = glds && gemm_glds_ok(mp.g[i]);
    int kind = GK_MULTI16;
    if (bf16) {
        for (int i = 0; i < mp.n; ++i) if (mp.g[i].Ah) ++cx.plane_problems;
        kind = gemm_bf16_multi_launch(mp, T, any_dual, grid, stream);
        glds = false;
    } else
#if !defined(MTTS_EMU)
    if (glds) { gemm_glds_multi_launch(mp, grid, stream, any_dual); kind = any_dual ? GK_GLDS_MULTI_DUAL : GK_GLDS_MULTI; }
    else
#endif
    if (bk32 && maxK >= 1024 && !cx.prefer_bk16) {
        int kl = gemm_kloop_variant();
        for (int i = 0; i < mp.n; ++i) kl = std::min(kl, gemm_kloop_for(mp.g[i], 32));
        if (any_dual) {
            if (kl == 4) MTTS_LAUNCH((gemm_f32_multi_dual_kernel<64, 64, 32, 4>), grid, block, stream, mp);
            else if (kl == 1) MTTS_LAUNCH((gemm_f32_multi_dual_kernel<64, 64, 32, 1>), grid, block, stream, mp);
            else MTTS_LAUNCH((gemm_f32_multi_dual_kernel<64, 64, 32, 0>), grid, block, stream, mp);
            kind = GK_MULTI32_DUAL;
        } else {
            if (ln_used > 0) MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 32, 4, true>), grid, block, stream, mp);
            else if (kl == 4) MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 32, 4>), grid, block, stream, mp);
            else if (kl == 1) MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 32, 1>), grid, block, stream, mp);
            else MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 32, 0>), grid, block, stream, mp);
            kind = GK_MULTI32;
        }
    } else {
        int kl = gemm_kloop_variant() == 4 ? 4 : 0;
        for (int i = 0; i < mp.n; ++i) kl = std::min(kl, gemm_kloop_for(mp.g[i], 16) == 4 ? 4 : 0);
        if (any_dual) {
            if (kl == 4) MTTS_LAUNCH((gemm_f32_multi_dual_kernel<64, 64, 16, 4>), grid, block, stream, mp);
            else MTTS_LAUNCH((gemm_f32_multi_dual_kernel<64, 64, 16, 0>), grid, block, stream, mp);
            kind = GK_MULTI16_DUAL;
        } else {
            if (ln_used > 0) MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 16, 4, true>), grid, block, stream, mp);
            else if (kl == 4) MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 16, 4>), grid, block, stream, mp);
            else MTTS_LAUNCH((gemm_f32_multi_kernel<64, 64, 16, 0>), grid, block, stream, mp);
        }
    }
    cx.last_kind = kind;
    if (prof.enabled) {
        hipEventRecord(e1, stream);
        GemmProfiler::Rec rec{kind, flops, e0, e1};
        rec.form = 3; rec.tile = bf16 ? 16000 + T : (glds ? 4064 : 64); rec.N = mp.n; rec.K = maxK; rec.groups = mp.start[mp.n]; rec.splitk = max_split; rec.rows = rows; rec.bytes = bytes; rec.tag = prof.tag;  // multi: N = problems, K = longest K, groups = workgroups
        prof.recs.push_back(rec);
    }
    b.q.clear();
}

}  // namespace mtts
