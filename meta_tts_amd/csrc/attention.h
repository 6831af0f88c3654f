// Fused scaled-dot-product attention forward: softmax(Q K^T / sqrt(dk)) V for one (task, sequence, head) and one tile of 32 query
// rows per workgroup — Q K^T on the matrix cores into an LDS-resident score tile, the row softmax on wavefront (DPP) reductions, P V on
// the matrix cores from that same LDS tile.  The probabilities are written to HBM exactly once (the hand-derived backward reads them:
// dV = P^T dO, dS = P o (dP - rowsum(dP o P)), engine.h: fft_bwd); the (L, L) score matrix never makes the HBM round trip
// QK^T-GEMM -> softmax kernel -> PV-GEMM it made as three launches.
//
// Reference: transformer/Modules.py:14-25 (ScaledDotProductAttention: bmm, / temperature, masked_fill(-inf) over padded keys,
// softmax(dim=2), bmm) called from transformer/SubLayers.py:42-52 per head.  Padded keys / query rows are not part of a sequence's rows
// here (packed frame space, engine.h), so no mask tensor exists: the key loop simply ends at L.
//
// Per workgroup (256 threads = 4 wavefronts, one per SIMD; two workgroups per CU for L <= 608):
//   LDS  S[32][LCAP + 4]   scores, then probabilities, of the 32 query rows against ALL keys of the sequence — the ONLY LDS object
//                          (LCAP = 128 / 352 / 608 / 1024: 17 / 46 / 78 / 132 KB; row stride = 4 x odd floats: ds_read_b128 conflict-free)
//   A    every wavefront holds the whole Q tile as MFMA A fragments in registers (dk / 2 VGPRs, read once from L2) and computes the
//        32 x 32 score blocks of key chunks w, w + 4, ...: B fragments = K rows straight from L2 in 16-byte pieces, the NEXT chunk's
//        fragments in flight behind the current chunk's 64 x v_mfma_f32_32x32x2_f32 (every K row is read by exactly one wavefront)
//   B    wavefront w normalises rows 8w .. 8w + 7: max / sum over the row with the DPP wavefront reductions, exp, one coalesced
//        store of the row of P to HBM, the probabilities stay in LDS
//   C    wavefront w owns output columns 32w .. 32w + 31 of O[32][dk]: A fragments (P) from LDS, B fragments = its 32 columns of V
//        straight from L2 (128-byte row segments), the next 32-key block's values in flight behind the current block's MFMAs
// Three barriers in all (none inside a loop).  fp32 throughout (exact fp32 MFMA chain): equal to the three-launch form up to summation order.
#pragma once
#include "gemm.h"
#include "rowops.h"

namespace mtts {

struct AttnFwdArgs {
    const AttnSeq* seqs;             // per group: offset of its P matrix, L, ldS
    const GemmGroupDesc* tab_qk;     // per group: a_off = Q rows, b_off = K rows (TAB_QK of the plan)
    const GemmGroupDesc* tab_pv;     // per group: b_off = V rows, c_off = O rows (TAB_PV)
    const float* Q; const float* K; const float* V;   // bases the offsets are added to (the fused qkv buffer: all three equal)
    int ld_q, ld_k, ld_v;
    float* P;
    float* O; int ld_o;
    float scale;
    int dk;                          // head width: multiple of 8, <= 128
    int prio;                        // > 0: the wavefronts raise their issue priority (the critical stream's launches beside side-stream work; gemm.h: GemmArgs::wave_prio)
    int rot;                         // 1: the key chunks of phase A start at wavefront (blockIdx.x + blockIdx.z) & 3 instead of 0 — with nkc % 4 != 0 the first
                                     // wavefronts carry one chunk more, and wavefront w of every workgroup sits on SIMD w: unrotated, SIMDs 0-1 of every CU carry the kernel
#if defined(MTTS_ATTN_DIAG)
    int diag;                        // diagnostic builds only (tools/attn_phases.sh; WRONG results, timing of the phases): bit 0 / 1 / 2 = skip phase A / B / C,
                                     // bit 3 = every lane of phase A reads its chunk's FIRST key row (one cache line per load instruction), bit 4 = no score stores
#endif
};

constexpr int kAttnQ = 32;           // query rows per workgroup
inline int attn_rot_default() { static const int r = [] { const char* e = getenv("MTTS_ATTN_ROT"); return e ? atoi(e) : 0; }(); return r; }

// NJ = dk / 8: a compile-time head width keeps every fragment load unconditional (a load behind a run-time "j < dk / 8" test, or a prefetch
// behind "is there a next chunk", makes hipcc drain vmcnt(0) in front of the MFMAs that follow: the whole L2 latency exposed per chunk —
// 61 us instead of the three-launch form's 47 on a single-task rank); the prefetch past the last chunk re-reads the sequence's last row.
template <int LCAP, int NJ>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnFwdArgs a) {
    constexpr int LD = LCAP + 4;
    __shared__ __attribute__((aligned(16))) float Ss[kAttnQ * LD];
    const int z = blockIdx.z;
    const AttnSeq sq = a.seqs[z];
    const int L = sq.L, ldS = sq.ldS;
    constexpr int dk = 8 * NJ;
    const int q0 = blockIdx.x * kAttnQ;
    if (q0 >= L) return;
    if (a.prio > 0) MTTS_SETPRIO_HIGH();
    const GemmGroupDesc dq = a.tab_qk[z], dv = a.tab_pv[z];
    const float* Qg = a.Q + dq.a_off;
    const float* Kg = a.K + dq.b_off;
    const float* Vg = a.V + dv.b_off;
    float* Og = a.O + dv.c_off;
    float* Pg = a.P + sq.s_off;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int nkc = (L + 31) / 32;                 // 32-key chunks
    const int wva = a.rot ? ((wave + (int)blockIdx.x + (int)blockIdx.z) & 3) : wave;   // this wavefront's first chunk of phase A

    // ---- phase A: S = scale * Q K^T
#if defined(MTTS_ATTN_DIAG)
    if (!(a.diag & 1))
#endif
    {
        // the Q tile as A fragments: lane (row l31, half h) holds channels 8j + 4h .. + 3 of every step j (rows beyond the sequence repeat
        // its last row: their results are never stored)
        const float* qp = Qg + (long long)(q0 + l31 < L ? q0 + l31 : L - 1) * a.ld_q + 4 * h;
        float4 qf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) qf[j] = ld4(qp + 8 * j);
#if defined(MTTS_ATTN_DIAG)
        const int kdiag = (a.diag & 8) ? 0 : 1;
        auto kptr = [&](int c) { return Kg + (long long)(c * 32 + kdiag * l31 < L ? c * 32 + kdiag * l31 : L - 1) * a.ld_k + 4 * h; };
#else
        auto kptr = [&](int c) { return Kg + (long long)(c * 32 + l31 < L ? c * 32 + l31 : L - 1) * a.ld_k + 4 * h; };
#endif
#if defined(MTTS_EMU)
        for (int c = wave; c < nkc; c += 4) {
            const float* kp = kptr(c) - 4 * h;
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float* qr = Qg + (long long)(q0 + row < L ? q0 + row : L - 1) * a.ld_q;
                float s = 0.f;
                for (int k = 0; k < dk; ++k) s = fmaf(qr[k], kp[k], s);
                Ss[row * LD + c * 32 + l31] = s * a.scale;
            }
        }
        (void)qf;
#else
        float4 kf[2][NJ];
        {
            const float* kp = kptr(wva);   // (clamped rows when this wavefront has no chunk at all)
#pragma unroll
            for (int j = 0; j < NJ; ++j) kf[0][j] = ld4(kp + 8 * j);
        }
        auto chunk = [&](int c, const float4 (&kc)[NJ], float4 (&kn)[NJ]) {
            {   // the next chunk's fragments go in flight behind this chunk's MFMAs (past the end: clamped rows, never used)
                const float* kp = kptr(c + 4);
#pragma unroll
                for (int j = 0; j < NJ; ++j) kn[j] = ld4(kp + 8 * j);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch AHEAD of the MFMAs (hipcc otherwise sinks each load to just before its use)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j].x, kc[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j].y, kc[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j].z, kc[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j].w, kc[j].w, acc, 0, 0, 0);
            }
            const int col = c * 32 + l31;
#if defined(MTTS_ATTN_DIAG)
            if (a.diag & 16) { if (acc[0] == 12345.678f) Ss[col] = acc[0]; return; }
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) Ss[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + col] = acc[r] * a.scale;
        };
        for (int c = wva; c < nkc; c += 8) {
            chunk(c, kf[0], kf[1]);
            if (c + 4 < nkc) chunk(c + 4, kf[1], kf[0]);
        }
#endif
    }
    __syncthreads();

    // ---- phase B: row softmax over the L valid keys; P -> HBM once, and kept in LDS (columns L .. 32 * nkc zeroed for phase C).
    // A wavefront owns 8 rows and walks them 4 at a time (four independent reduction chains per pass over the LDS row).
#if defined(MTTS_ATTN_DIAG)
    if (!(a.diag & 2))
#endif
    for (int rg = 0; rg < kAttnQ / 4; rg += 4) {
        const int row0 = wave * (kAttnQ / 4) + rg;
        if (q0 + row0 >= L) break;   // (wave-uniform)
        float* s0 = Ss + row0 * LD;
        float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        for (int c = lane; c < L; c += 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mx[q] = fmaxf(mx[q], s0[q * LD + c]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) mx[q] = wave_max(mx[q]);
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane; c < L; c += 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float e = expf(s0[q * LD + c] - mx[q]); s0[q * LD + c] = e; sum[q] += e; }
        }
        float inv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) inv[q] = 1.f / wave_sum(sum[q]);
        for (int c = lane; c < nkc * 32; c += 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = c < L ? s0[q * LD + c] * inv[q] : 0.f;
                s0[q * LD + c] = v;
                if (c < ldS && q0 + row0 + q < L) Pg[(long long)(q0 + row0 + q) * ldS + c] = v;
            }
        }
    }
    __syncthreads();

    // ---- phase C: O = P V; this wavefront's 32 output columns
#if defined(MTTS_ATTN_DIAG)
    if (a.diag & 4) return;
#endif
    if (32 * wave >= dk) return;
    const int ocol = 32 * wave + l31 < dk ? 32 * wave + l31 : dk - 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#if defined(MTTS_EMU)
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float s = 0.f;
        for (int k = 0; k < L; ++k) s = fmaf(Ss[row * LD + k], Vg[(long long)k * a.ld_v + ocol], s);
        acc[r] = s;
    }
#else
    // B fragments: step (j, e) of a 32-key block needs V[8j + 4h + e][ocol] (keys beyond the sequence: its last row, P is zero there)
    auto vload = [&](int kb, float (&v)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kb * 32 + 8 * j + 4 * h + e;
                v[4 * j + e] = Vg[(long long)(key < L ? key : L - 1) * a.ld_v + ocol];
            }
    };
    float vf[2][16];
    vload(0, vf[0]);
    auto block = [&](int kb, const float (&vc)[16], float (&vn)[16]) {
        vload(kb + 1, vn);   // (unconditional: past the last block the keys clamp to the sequence's last row, the values are never used)
        __builtin_amdgcn_sched_barrier(0);   // the next block's loads stay in front of this block's MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a4 = ld4(Ss + l31 * LD + kb * 32 + 8 * j + 4 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, vc[4 * j + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, vc[4 * j + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, vc[4 * j + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, vc[4 * j + 3], acc, 0, 0, 0);
        }
    };
    for (int kb = 0; kb < nkc; kb += 2) {
        block(kb, vf[0], vf[1]);
        if (kb + 1 < nkc) block(kb + 1, vf[1], vf[0]);
    }
#endif
    if (32 * wave + l31 < dk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (qrow < L) Og[(long long)qrow * a.ld_o + 32 * wave + l31] = acc[r];
        }
    }
}

// the fused kernel serves sequences of up to 1024 keys and heads of 16 / 32 / 64 / 128 channels (everything base.yaml produces in
// training; eval-mode synthesis beyond max_seq_len and other head widths keep the three-launch path)
inline bool attn_fused_ok(int max_L, int dk) { return max_L >= 1 && max_L <= 1024 && (dk == 16 || dk == 32 || dk == 64 || dk == 128); }
template <int NJ>
inline void attn_fwd_launch_nj(const AttnFwdArgs& a, int max_L, dim3 grid, hipStream_t stream) {
    dim3 block(256);
    if (max_L <= 128) MTTS_LAUNCH((attn_fwd_kernel<128, NJ>), grid, block, stream, a);
    else if (max_L <= 352) MTTS_LAUNCH((attn_fwd_kernel<352, NJ>), grid, block, stream, a);
    else if (max_L <= 608) MTTS_LAUNCH((attn_fwd_kernel<608, NJ>), grid, block, stream, a);
    else MTTS_LAUNCH((attn_fwd_kernel<1024, NJ>), grid, block, stream, a);
}
inline void attn_fwd_launch(const AttnFwdArgs& a, int max_L, int groups, hipStream_t stream) {
    dim3 grid((unsigned)((max_L + kAttnQ - 1) / kAttnQ), 1, (unsigned)groups);
    if (a.dk == 128) attn_fwd_launch_nj<16>(a, max_L, grid, stream);
    else if (a.dk == 64) attn_fwd_launch_nj<8>(a, max_L, grid, stream);
    else if (a.dk == 32) attn_fwd_launch_nj<4>(a, max_L, grid, stream);
    else attn_fwd_launch_nj<2>(a, max_L, grid, stream);
}

}  // namespace mtts
