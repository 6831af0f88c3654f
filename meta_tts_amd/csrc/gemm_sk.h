// Persistent multi-problem GEMM with a work queue and a split tail ("stream-K tail").
//
// Why: every large contraction of the FastSpeech2 step is a few thousand 64x64 tiles on ~1300 resident workgroup slots, i.e.
// 2-4 dispatch "waves" — the last, partly filled wave cost 10-25 % of the launch (conv1 forward: 4272 tiles = 3.34 waves, timed
// as 4), and on a single-task rank (one task per GPU) a launch is 100-800 tiles whose K-loops differ by 4x between the problems
// it carries.  The kernels of gemm.h reach 0.75-0.78 of the fp32 matrix peak on shapes that fill whole waves; what was missing
// is the schedule.
//
// What: ONE grid of G persistent workgroups (G <= the resident capacity of the chip) pulls ITEMS from a device queue (one
// returning atomic per item, fetched one item ahead so its latency hides behind the K-loop).  The item list is every output
// tile of every (problem, group) of the launch in problem-major order (the launcher sorts problems longest K-loop first), so
// equal-length tiles run in lock-step and keep sharing their operand panels through the L2 exactly as a plain grid does.
// Only tiles that would START too late to finish with the rest are cut: a round of G tiles of c K-chunks each that starts with R
// chunks of work left in the whole launch (R counted over all workgroups) finishes on time iff c * G <= R (+ a tolerance); otherwise
// its tiles are cut into S = ceil(c * G / R) pieces (S <= s_max, every piece >= min_chunks chunks), each piece an item of its own.  The
// pieces of a tile park their partial accumulators in a slab each; the last piece to arrive sums the slabs in piece order and
// runs the fused epilogue (same release / counter / acquire hand-off as gemm.h's split-K), so results do not depend on the
// arrival order (run-to-run identical for a given launch configuration).
//
// The schedule is computed INSIDE the kernel from the launch's own descriptors (per-group row counts live on the device: the
// batch plan is built there without a host round trip): one thread per (problem, group) entry, two block scans.
#pragma once
#include "gemm.h"

namespace mtts {

constexpr int kSkMaxEntries = 192;   // (problem, group) pairs of one launch (C3: 8 tasks x 5 utterances x 2 heads x 2 problems = 160)

// inclusive scan of a[1..n] (a[0] = 0 is the caller's), n <= 256, all 256 threads call
__device__ __forceinline__ void sk_scan2(int* a, int* b, int n) {
#if defined(MTTS_EMU)
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i <= n; ++i) { a[i] += a[i - 1]; if (b) b[i] += b[i - 1]; }
    __syncthreads();
#else
    __shared__ int s_wa[4], s_wb[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    int va = tid < n ? a[1 + tid] : 0, vb = (b && tid < n) ? b[1 + tid] : 0;
    // in-row Hillis-Steele on the DPP row_shr path (out-of-row sources read 0), then the three row totals by v_readlane
    auto shr = [](int v, auto ctrl) { return __builtin_amdgcn_update_dpp(0, v, decltype(ctrl)::value, 0xF, 0xF, true); };
    va += shr(va, std::integral_constant<int, 0x111>()); vb += shr(vb, std::integral_constant<int, 0x111>());
    va += shr(va, std::integral_constant<int, 0x112>()); vb += shr(vb, std::integral_constant<int, 0x112>());
    va += shr(va, std::integral_constant<int, 0x114>()); vb += shr(vb, std::integral_constant<int, 0x114>());
    va += shr(va, std::integral_constant<int, 0x118>()); vb += shr(vb, std::integral_constant<int, 0x118>());
    const int row = lane >> 4;
    const int a0 = __builtin_amdgcn_readlane(va, 15), a1 = __builtin_amdgcn_readlane(va, 31), a2 = __builtin_amdgcn_readlane(va, 47);
    const int b0 = __builtin_amdgcn_readlane(vb, 15), b1 = __builtin_amdgcn_readlane(vb, 31), b2 = __builtin_amdgcn_readlane(vb, 47);
    va += (row >= 1 ? a0 : 0) + (row >= 2 ? a1 : 0) + (row >= 3 ? a2 : 0);
    vb += (row >= 1 ? b0 : 0) + (row >= 2 ? b1 : 0) + (row >= 3 ? b2 : 0);
    if (lane == 63) { s_wa[wave] = va; s_wb[wave] = vb; }
    __syncthreads();
    for (int w = 0; w < wave; ++w) { va += s_wa[w]; vb += s_wb[w]; }
    if (tid < n) { a[1 + tid] = va; if (b) b[1 + tid] = vb; }
    __syncthreads();
#endif
}

// Split level of the tiles of one entry (T tiles of c chunks each, P chunks of work before the entry, W in the launch, G workgroups).
// Model: list scheduling on G workgroups.  The entry's tiles start in rounds of G ("waves"): wave w starts when P + w * G * c chunks
// of work are done, i.e. with R_w = W - P - w * G * c chunks left in the launch; whole tiles started then end c chunk-times later,
// on time iff c * G <= R_w (+ tol).  Otherwise the wave's tiles are cut into ceil(c * G / R_w) pieces each.  Returns the number of
// tiles (a prefix of the entry's tile list, whole waves) whose level is <= s.
__device__ __forceinline__ int sk_cnt_le(int s, int T, int c, long long P, long long W, long long tol, int G) {
    const long long cg = (long long)c * G;
    const long long need = (cg + s - 1) / s;                   // R_w >= need  <=>  level <= s
    const long long room = W + tol - P - need;                 // w * c * G <= room
    if (room < 0) return 0;
    const long long t = (room / cg + 1) * G;
    return t < T ? (int)t : T;
}

// The pieces of one split tile: `S` slabs starting at `base`, arrival counter `ctr`.  Same hand-off as splitk_combine (gemm.h).
template <int NTH>
__device__ __forceinline__ bool sk_combine(float* base, int* ctr, int split, int S, f32x16 (&acc)[1][1]) {
    constexpr int PART = NTH * 16;
    const int tid = threadIdx.x;
    float* mine = base + (long long)split * PART;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        float4 v;
        v.x = acc[0][0][4 * r4]; v.y = acc[0][0][4 * r4 + 1]; v.z = acc[0][0][4 * r4 + 2]; v.w = acc[0][0][4 * r4 + 3];
        st4(mine + (r4 * NTH + tid) * 4, v);
    }
    __shared__ int s_last;
    MTTS_WAIT_VMEM();
    __syncthreads();
    if (tid == 0) {
        MTTS_FENCE_RELEASE_AGENT();
        MTTS_WAIT_VMEM();
        s_last = (MTTS_ATOMIC_INC_AGENT(ctr) == S - 1) ? 1 : 0;
        if (s_last) MTTS_FENCE_ACQUIRE_AGENT();
    }
    __syncthreads();
    if (!s_last) return false;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        float4 sum = zero4();
        for (int sp = 0; sp < S; ++sp) {
            const float4 v = ld4(base + (long long)sp * PART + (r4 * NTH + tid) * 4);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        acc[0][0][4 * r4] = sum.x; acc[0][0][4 * r4 + 1] = sum.y; acc[0][0][4 * r4 + 2] = sum.z; acc[0][0][4 * r4 + 3] = sum.w;
    }
    if (tid == 0) *ctr = 0;   // re-armed for the next launch
    return true;
}

template <int FORM, int BK>
__device__ __forceinline__ void sk_tile(const GemmMulti& mp, int p, int z, int tile, int tn, int spt, int split, int S, int slab, float* smem) {
    const GemmArgs& g = mp.g[p];
    const GemmProb pr = gemm_resolve(g, z);
    const bool has_cs = gemm_has_colsum<FORM>(g);
    const int tiles_nc = (pr.N + 63) / 64;
    const int m0 = (tile / tn) * 64, n0 = (tile % tn) * 64;
    const bool cs_tile = has_cs && n0 == tiles_nc * 64;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int cps = (spt + S - 1) / S;
    const int c_lo = split * cps, c_hi = (c_lo + cps < spt) ? c_lo + cps : spt;
    gemm_f32_kloop<FORM, 64, 64, BK, true>(g, pr, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
    if (S > 1 && !sk_combine<256>(mp.sk.ws + (long long)slab * 4096, mp.sk.ctr + slab, split, S, acc)) return;
    gemm_finish<1, 1, 2, 2>(g, pr, z, m0, n0, cs_tile, acc);
}

// WPE: waves per SIMD the register allocation must leave room for (= resident workgroups per CU)
template <int BK, int WPE>
__global__ __launch_bounds__(256) MTTS_WAVES_PER_EU(WPE) void gemm_sk_kernel(GemmMulti mp) {
    constexpr int F0 = GemmSmem<GEMM_NT, 64, 64, BK>::FLOATS, F1 = GemmSmem<GEMM_NN, 64, 64, BK>::FLOATS, F2 = GemmSmem<GEMM_TN, 64, 64, BK>::FLOATS;
    constexpr int FL = F0 > F1 ? (F0 > F2 ? F0 : F2) : (F1 > F2 ? F1 : F2);
    __shared__ __attribute__((aligned(16))) float smem[FL];
    __shared__ int s_upref[kSkMaxEntries + 1];   // K-chunks of work before entry e
    __shared__ int s_ipref[kSkMaxEntries + 1];   // items before entry e
    __shared__ int s_spref[kSkMaxEntries + 1];   // pieces of split tiles before entry e (slab index)
    __shared__ int s_tn[kSkMaxEntries], s_spt[kSkMaxEntries], s_tiles[kSkMaxEntries];
    __shared__ int s_item, s_state[4];
    const int tid = threadIdx.x;
    const GemmSk& sk = mp.sk;
    const int E = sk.ent_start[mp.n], G = (int)gridDim.x;

    // ---- schedule: one entry per thread ----
    int my_tiles = 0, my_spt = 1;
    for (int p = 0; p < mp.n; ++p) {
        if (tid < sk.ent_start[p] || tid >= sk.ent_start[p + 1]) continue;
        const GemmArgs& g = mp.g[p];
        const int z = tid - sk.ent_start[p];
        int M = g.M, N = g.N, K = g.K;
        if (g.table) { const GemmGroupDesc d = g.table[z]; M = d.M; N = d.N; K = d.K; }
        else if (g.dimptr) {
            const int v = g.dimptr[(long long)z * g.dim_stride] * g.dim_mult;
            if (g.dim_sel == 0) M = v; else K = v;
        }
        const bool cs = mp.form[p] == GEMM_TN && g.colsum != nullptr && !g.table;
        const int tn = (N + 63) / 64 + (cs ? 1 : 0);
        my_spt = (K + BK - 1) / BK;
        my_tiles = (M > 0 && N > 0 && K > 0) ? ((M + 63) / 64) * tn : 0;
        s_tn[tid] = tn; s_spt[tid] = my_spt; s_tiles[tid] = my_tiles;
        s_upref[tid + 1] = my_tiles * my_spt;
    }
    if (tid == 0) { s_upref[0] = 0; s_ipref[0] = 0; s_spref[0] = 0; }
    sk_scan2(s_upref, nullptr, E);
    const long long W = MTTS_UNIFORM(s_upref[E]);
    if (W == 0) return;
    const long long tol = W / sk.tol_div;
    const int s_cap = (my_spt / sk.min_chunks < sk.s_max) ? (my_spt / sk.min_chunks > 1 ? my_spt / sk.min_chunks : 1) : sk.s_max;
    if (tid < E) {
        int items = my_tiles, pieces = 0;
        if (s_cap > 1 && my_tiles > 0) {
            const long long P = s_upref[tid];
            int prev = sk_cnt_le(1, my_tiles, my_spt, P, W, tol, G);
            items = prev;
            for (int s = 2; s < s_cap; ++s) {
                const int le = sk_cnt_le(s, my_tiles, my_spt, P, W, tol, G);
                items += (le - prev) * s;
                prev = le;
            }
            items += (my_tiles - prev) * s_cap;
            pieces = items - sk_cnt_le(1, my_tiles, my_spt, P, W, tol, G);
        }
        s_ipref[tid + 1] = items; s_spref[tid + 1] = pieces;
    }
    sk_scan2(s_ipref, s_spref, E);
    bool nosplit = false;
    if (MTTS_UNIFORM(s_spref[E]) > sk.slabs) {   // more pieces than slabs (never at the model's sizes): plain tiles, one item each
        nosplit = true;
        __syncthreads();
        if (tid < E) s_ipref[tid + 1] = my_tiles;
        sk_scan2(s_ipref, nullptr, E);
    }
    // ---- queue ----
    // Loop state lives in LDS (s_state), not in registers: the K-loop below is register-tight (occupancy), and everything the
    // scheduler needs between two items is re-read after the item's closing barrier.
    if (tid == 0) {
        s_state[0] = s_ipref[E];                          // items of this launch
        s_state[1] = 0;                                   // entry of the previous item (the search resumes there)
        s_state[2] = nosplit ? 1 : 0;
        if (blockIdx.x == 0) *sk.head_next = 0;           // the next launch of this context uses the other head
        s_item = MTTS_ATOMIC_INC_AGENT(sk.head);
    }
    __syncthreads();
    for (;;) {
        const int item = MTTS_UNIFORM(s_item);
        if (item >= MTTS_UNIFORM(s_state[0])) break;
        int nxt = 0;
        if (tid == 0) nxt = MTTS_ATOMIC_INC_AGENT(sk.head);   // one item ahead: the round trip hides behind this item's K-loop
        int e = MTTS_UNIFORM(s_state[1]);
        while (MTTS_UNIFORM(s_ipref[e + 1]) <= item) ++e;
        const int j = item - MTTS_UNIFORM(s_ipref[e]), c = MTTS_UNIFORM(s_spt[e]), T = MTTS_UNIFORM(s_tiles[e]), tn = MTTS_UNIFORM(s_tn[e]);
        int tile = j, split = 0, S = 1, slab = 0;
        if (!MTTS_UNIFORM(s_state[2])) {
            const int cap = (c / sk.min_chunks < sk.s_max) ? (c / sk.min_chunks > 1 ? c / sk.min_chunks : 1) : sk.s_max;
            if (cap > 1) {
                const long long Wl = MTTS_UNIFORM(s_upref[E]), tl = Wl / sk.tol_div;
                const long long P = MTTS_UNIFORM(s_upref[e]);
                const int le1 = sk_cnt_le(1, T, c, P, Wl, tl, G);
                if (j >= le1) {
                    int jj = j - le1, prev = le1;
                    S = cap;
                    for (int s = 2; s < cap; ++s) {
                        const int le = sk_cnt_le(s, T, c, P, Wl, tl, G);
                        if (jj < (le - prev) * s) { S = s; break; }
                        jj -= (le - prev) * s; prev = le;
                    }
                    tile = prev + jj / S; split = jj - (jj / S) * S;
                    slab = MTTS_UNIFORM(s_spref[e]) + (j - le1) - split;   // slab of the tile's first piece
                }
            }
        }
        // (the 64-bit divisions above run on the vector ALU: tell the compiler their results are wave-uniform, or the whole K-loop
        // below is compiled with exec-masked control flow)
        tile = MTTS_UNIFORM(tile); split = MTTS_UNIFORM(split); S = MTTS_UNIFORM(S); slab = MTTS_UNIFORM(slab);
        if (tid == 0) s_state[1] = e;
        int p = 0;
        while (p + 1 < mp.n && e >= sk.ent_start[p + 1]) ++p;
        p = MTTS_UNIFORM(p);
        const int z = MTTS_UNIFORM(e - sk.ent_start[p]);
        const int form = mp.form[p];
        if (form == GEMM_NT) sk_tile<GEMM_NT, BK>(mp, p, z, tile, tn, c, split, S, slab, smem);
        else if (form == GEMM_NN) sk_tile<GEMM_NN, BK>(mp, p, z, tile, tn, c, split, S, slab, smem);
        else sk_tile<GEMM_TN, BK>(mp, p, z, tile, tn, c, split, S, slab, smem);
        __syncthreads();          // every wave is done with the LDS tiles (and with s_item)
        if (tid == 0) s_item = nxt;
        __syncthreads();
    }
}

// ---- host side ----
inline int& gemm_sk_enabled() {   // MTTS_SK=0: the plain grids of gemm.h everywhere (A/B runs)
    static int v = [] { const char* e = getenv("MTTS_SK"); return e ? atoi(e) : 1; }();
    return v;
}
inline int gemm_sk_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// resident capacity of the chip for the persistent kernel, in workgroups
// kernel variants: BK = 16 compiled for 5 (default) or 4 resident workgroups per CU, BK = 32 for 4 or 3 (MTTS_SK_WPE)
inline int gemm_sk_wpe(int bk) {
    static const int e = gemm_sk_env("MTTS_SK_WPE", 0);
    if (bk == 32) return e == 3 ? 3 : 4;
    return e == 4 ? 4 : 5;
}
inline int gemm_sk_capacity(int bk) {
#if defined(MTTS_EMU)
    (void)bk;
    static const int cap = gemm_sk_env("MTTS_SK_WGS", 8);
    return cap;
#else
    static int cap[2] = {0, 0};
    int& c = cap[bk == 32];
    if (c == 0) {
        int dev = 0, cus = 256, per_cu = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int wpe = gemm_sk_wpe(bk);
        if (bk == 32) { if (wpe == 3) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<32, 3>, 256, 0); else hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<32, 4>, 256, 0); }
        else { if (wpe == 4) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<16, 4>, 256, 0); else hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<16, 5>, 256, 0); }
        // never more workgroups than are resident at once (a queued workgroup would start when the others are done): the API's answer,
        // capped by what the variant was compiled for
        per_cu = std::max(1, std::min(per_cu, gemm_sk_env("MTTS_SK_OCC", wpe)));
        c = gemm_sk_env("MTTS_SK_WGS", cus * per_cu);
    }
    return c;
#endif
}

// Launch the queued problems (already sorted longest K-loop first) as one persistent grid.  Returns false when the launch does
// not qualify (too many entries / too little work per workgroup): the caller then takes the plain grids.
inline bool gemm_sk_launch(GemmCtx& cx, GemmMulti& mp, const std::vector<GemmPending>& q, hipStream_t stream, bool force) {
    if (!cx.wsp.ws || !cx.sk_heads) return false;
    static const int bk_env = gemm_sk_env("MTTS_SK_BK", 16);
    int bk = bk_env == 32 ? 32 : 16;
    int entries = 0;
    double work = 0.0, tiles = 0.0;   // host estimate (exact per-group sizes live on the device)
    int max_chunks = 0;
    for (int i = 0; i < mp.n; ++i) {
        const GemmPending& p = q[i];
        if (p.g.taps > 1 && p.g.tap_k % 32 != 0) bk = 16;
        mp.sk.ent_start[i] = entries;
        entries += p.groups;
    }
    mp.sk.ent_start[mp.n] = entries;
    if (entries > kSkMaxEntries) return false;
    for (int i = 0; i < mp.n; ++i) {
        const GemmPending& p = q[i];
        const double t = std::ceil(p.rows / 64.0) * gemm_tiles_n(p.g, p.max_N, 64);
        const int ch = std::max(1, (p.g.K + bk - 1) / bk);
        tiles += t; work += t * ch; max_chunks = std::max(max_chunks, ch);
    }
    const int cap = gemm_sk_capacity(bk);
#if defined(MTTS_EMU)
    constexpr int kMinUnits = 0, kMinTile = 0;   // the emulator build sends every queued launch through the work-queue kernel (CPU tests of its logic)
#else
    constexpr int kMinUnits = 64, kMinTile = 32;
#endif
    static const int min_units = gemm_sk_env("MTTS_SK_MIN_UNITS", kMinUnits);   // K-chunks (of 16) per workgroup below which the queue is not worth its prologue
    static const int min_tile = gemm_sk_env("MTTS_SK_MIN_TILE", kMinTile);      // ... and the launch's longest K-loop, in chunks of 16
    const double unit = bk / 16.0;
    if (!force && (work * unit < (double)min_units * cap || max_chunks * unit < min_tile)) return false;
    int G = (int)std::min<double>(cap, std::max(1.0, tiles));
    mp.sk.s_max = gemm_sk_env("MTTS_SK_SMAX", 4);
    mp.sk.min_chunks = std::max(1, gemm_sk_env("MTTS_SK_MINCH", 16) * 16 / bk);
    mp.sk.tol_div = std::max(1, gemm_sk_env("MTTS_SK_TOL", 16));
    mp.sk.slabs = (int)std::min<long long>(kSplitWsFloats / 4096, kSplitCtrs);
    mp.sk.ws = cx.wsp.ws; mp.sk.ctr = cx.wsp.ctr;
    mp.sk.head = cx.sk_heads + (cx.sk_parity & 1);
    mp.sk.head_next = cx.sk_heads + ((cx.sk_parity + 1) & 1);
    cx.sk_parity ^= 1;
    for (int i = 0; i < mp.n; ++i) { mp.g[i].splitk = 1; mp.g[i].swizzle = 0; }
    dim3 block(256), grid((unsigned)G, 1, 1);
    const int wpe = gemm_sk_wpe(bk);
    if (bk == 32) {
        if (wpe == 3) { MTTS_LAUNCH((gemm_sk_kernel<32, 3>), grid, block, stream, mp); }
        else { MTTS_LAUNCH((gemm_sk_kernel<32, 4>), grid, block, stream, mp); }
    } else {
        if (wpe == 4) { MTTS_LAUNCH((gemm_sk_kernel<16, 4>), grid, block, stream, mp); }
        else { MTTS_LAUNCH((gemm_sk_kernel<16, 5>), grid, block, stream, mp); }
    }
    cx.last_kind = bk == 32 ? GK_SK32 : GK_SK16;
    return true;
}

}  // namespace mtts
