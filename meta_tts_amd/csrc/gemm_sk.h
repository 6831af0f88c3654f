// Persistent multi-problem GEMM with a work queue and a split tail ("stream-K tail").
//
// Why: every large contraction of the FastSpeech2 step is a few thousand 64x64 tiles on ~1300 resident workgroup slots, i.e.
// 2-4 dispatch "waves" — the last, partly filled wave cost 10-25 % of the launch (conv1 forward: 4272 tiles = 3.34 waves, timed
// as 4), and on a single-task rank (one task per GPU) a launch is 100-800 tiles whose K-loops differ by 4x between the problems
// it carries.  The kernels of gemm.h reach 0.75-0.78 of the fp32 matrix peak on shapes that fill whole waves; what was missing
// is the schedule.
//
// What: ONE grid of G persistent workgroups (G <= the resident capacity of the chip) pulls ITEMS from a device queue (one
// returning atomic per item, issued between the item's K-loop and its epilogue so that its round trip hides behind the epilogue).  The item list is every output
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

constexpr int kSkMaxEntries = 160;   // (problem, group) pairs of one launch (C3: 8 tasks x 5 utterances x 2 heads x 2 problems = 160)

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
// tiles (a prefix of the entry's tile list, whole waves) whose level is <= s.  32-bit arithmetic: the launcher keeps W below 2^30.
__device__ __forceinline__ int sk_cnt_le(int s, int T, int c, int P, int W, int tol, int G) {
    const unsigned cg = (unsigned)c * (unsigned)G;
    const unsigned need = (cg + (unsigned)s - 1u) / (unsigned)s;   // R_w >= need  <=>  level <= s
    const int room = W + tol - P - (int)need;                     // w * c * G <= room
    if (room < 0) return 0;
    const unsigned waves = (unsigned)room / cg + 1u;
    const unsigned full = ((unsigned)T + (unsigned)G - 1u) / (unsigned)G;
    return waves >= full ? T : (int)(waves * (unsigned)G);
}

// One item.  The NEXT item is claimed here, between the K-loop and the epilogue: early enough for the atomic's round trip to hide behind
// the epilogue, late enough that a workgroup never sits on a second item while another workgroup has none (claiming it before the
// K-loop doubled the time of every launch with about one item per workgroup).
template <int FORM, int BK>
__device__ __forceinline__ void sk_tile(const GemmMulti& mp, int p, int z, int tile, int tn, int spt, int split, int S, int slab, float* smem,
                                        int* head, int& nxt) {
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
    const int nsrc = (g.A2 != nullptr && !cs_tile) ? 2 : 1;   // dual-source problems: both K-loops over the piece's chunk range
    GemmProb src = pr;
#pragma nounroll
    for (int s = 0; s < nsrc; ++s) {
        if (s) { __syncthreads(); src = gemm_resolve2(g, z, pr); }
        gemm_f32_kloop<FORM, 64, 64, BK, true>(g, src, z, m0, n0, cs_tile, c_lo, c_hi, smem, acc);
    }
    if (threadIdx.x == 0) nxt = MTTS_ATOMIC_INC_AGENT(head);
    if (S > 1 && !slab_combine<1, 1, 256>(mp.sk.ws + (long long)slab * 4096, mp.sk.ctr + slab, split, S, acc, (g.flags & GEMM_SLAB_FENCE) != 0)) return;
    gemm_finish<1, 1, 2, 2>(g, pr, z, m0, n0, cs_tile, acc);
}

// Item queues: eight heads, one per XCD (workgroup b runs on XCD b % 8 — an observation used for speed only: any placement gives
// the same results).  The global item list is dealt to the queues in runs of `run` consecutive items (the n-tiles of an m-tile, which
// read the same operand panel, then meet in one L2), round-robin — queue x holds the runs x, x + 8, x + 16, ...  A workgroup
// drains the queue of its XCD, then helps the others.
__device__ __forceinline__ int sk_item_of(int j, int q, int run) { return ((j / run) * 8 + q) * run + (j % run); }

template <int BK>
__global__ __launch_bounds__(256) void gemm_sk_kernel(GemmMulti mp) {
    constexpr int F0 = GemmSmem<GEMM_NT, 64, 64, BK>::FLOATS, F1 = GemmSmem<GEMM_NN, 64, 64, BK>::FLOATS, F2 = GemmSmem<GEMM_TN, 64, 64, BK>::FLOATS;
    constexpr int FL = F0 > F1 ? (F0 > F2 ? F0 : F2) : (F1 > F2 ? F1 : F2);
    __shared__ __attribute__((aligned(16))) float smem[FL];
    __shared__ int s_upref[kSkMaxEntries + 1];   // K-chunks of work before entry e
    __shared__ int s_ipref[kSkMaxEntries + 1];   // items before entry e
    __shared__ int s_spref[kSkMaxEntries + 1];   // pieces of split tiles before entry e (slab index)
    __shared__ int s_tn[kSkMaxEntries], s_spt[kSkMaxEntries], s_tiles[kSkMaxEntries];
    __shared__ int s_le[3][kSkMaxEntries];       // tiles of entry e with split level <= 1, 2, 3
    __shared__ int s_item, s_state[8];
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
    const int W = MTTS_UNIFORM(s_upref[E]);
    if (W == 0) return;
    const int tol = W / sk.tol_div;
    if (tid < E) {
        const int cap = (my_spt / sk.min_chunks < sk.s_max) ? (my_spt / sk.min_chunks > 1 ? my_spt / sk.min_chunks : 1) : sk.s_max;
        int le[3] = {my_tiles, my_tiles, my_tiles};
        int items = my_tiles;
        if (cap > 1 && my_tiles > 0) {
            const int P = s_upref[tid];
            int prev = 0;
            items = 0;
            for (int s = 1; s < cap; ++s) {     // cap <= 4
                le[s - 1] = sk_cnt_le(s, my_tiles, my_spt, P, W, tol, G);
                items += (le[s - 1] - prev) * s;
                prev = le[s - 1];
            }
            for (int s = cap; s <= 3; ++s) le[s - 1] = prev;
            items += (my_tiles - prev) * cap;
        }
        s_le[0][tid] = le[0]; s_le[1][tid] = le[1]; s_le[2][tid] = le[2];
        s_ipref[tid + 1] = items; s_spref[tid + 1] = items - le[0];
    }
    sk_scan2(s_ipref, s_spref, E);
    bool nosplit = false;
    if (MTTS_UNIFORM(s_spref[E]) > sk.slabs) {   // more pieces than slabs (never at the model's sizes): plain tiles, one item each
        nosplit = true;
        __syncthreads();
        if (tid < E) s_ipref[tid + 1] = my_tiles;
        sk_scan2(s_ipref, nullptr, E);
    }

    // ---- queues ----
    // Loop state lives in LDS (s_state), not in registers: the K-loop below is register-tight (occupancy), and everything the
    // scheduler needs between two items is re-read after the item's closing barrier.
    // s_state: 0 items of this launch, 1 entry of the previous item (the search resumes there), 2 no-split flag, 3 queue
    if (tid == 0) {
        s_state[0] = s_ipref[E];
        s_state[1] = 0;
        s_state[2] = nosplit ? 1 : 0;
        s_state[3] = (int)(blockIdx.x & 7);
        s_item = MTTS_ATOMIC_INC_AGENT(sk.head + (blockIdx.x & 7));
    }
    if (blockIdx.x == 0 && tid < 8) sk.head_next[tid] = 0;   // the next launch of this context uses the other set of heads
    __syncthreads();
    for (;;) {
        const int total = MTTS_UNIFORM(s_state[0]);
        int q = MTTS_UNIFORM(s_state[3]);
        int item = sk_item_of(MTTS_UNIFORM(s_item), q, sk.run);
        if (item >= total) {
            // this queue is drained: help the one with the most items left (one look at the eight heads, one returning atomic)
            __syncthreads();
            if (tid == 0) {
                const int run = sk.run, runs = (total + run - 1) / run, deficit = runs * run - total;
                int best = -1, best_rem = 0;
                for (int k = 1; k < 8; ++k) {
                    const int qq = (q + k) & 7;
                    int len = runs > qq ? ((runs - qq + 7) / 8) * run : 0;
                    if (len > 0 && ((runs - 1) & 7) == qq) len -= deficit;
                    const int rem = len - MTTS_ATOMIC_LOAD_AGENT(sk.head + qq);
                    if (rem > best_rem) { best = qq; best_rem = rem; }
                }
                s_state[3] = best;
                if (best >= 0) { s_state[1] = 0; s_item = MTTS_ATOMIC_INC_AGENT(sk.head + best); }
            }
            __syncthreads();
            if (MTTS_UNIFORM(s_state[3]) < 0) break;
            continue;
        }
        int nxt = 0;
        int e = MTTS_UNIFORM(s_state[1]);
        while (MTTS_UNIFORM(s_ipref[e + 1]) <= item) ++e;
        const int j = item - MTTS_UNIFORM(s_ipref[e]), c = MTTS_UNIFORM(s_spt[e]), tn = MTTS_UNIFORM(s_tn[e]);
        int tile = j, split = 0, S = 1, slab = 0;
        const int le1 = MTTS_UNIFORM(s_le[0][e]);
        if (!MTTS_UNIFORM(s_state[2]) && j >= le1) {
            const int cap = (c / sk.min_chunks < sk.s_max) ? (c / sk.min_chunks > 1 ? c / sk.min_chunks : 1) : sk.s_max;
            int jj = j - le1, prev = le1;
            S = cap;
            for (int s = 2; s < cap; ++s) {
                const int le = MTTS_UNIFORM(s_le[s - 1][e]);
                if (jj < (le - prev) * s) { S = s; break; }
                jj -= (le - prev) * s; prev = le;
            }
            tile = prev + jj / S; split = jj - (jj / S) * S;
            slab = MTTS_UNIFORM(s_spref[e]) + (j - le1) - split;   // slab of the tile's first piece
        }
        if (tid == 0) s_state[1] = e;
        int p = 0;
        while (p + 1 < mp.n && e >= sk.ent_start[p + 1]) ++p;
        p = MTTS_UNIFORM(p);
        const int z = MTTS_UNIFORM(e - sk.ent_start[p]);
        const int form = mp.form[p];
        if (form == GEMM_NT) sk_tile<GEMM_NT, BK>(mp, p, z, tile, tn, c, split, S, slab, smem, sk.head + q, nxt);
        else if (form == GEMM_NN) sk_tile<GEMM_NN, BK>(mp, p, z, tile, tn, c, split, S, slab, smem, sk.head + q, nxt);
        else sk_tile<GEMM_TN, BK>(mp, p, z, tile, tn, c, split, S, slab, smem, sk.head + q, nxt);
        __syncthreads();          // every wave is done with the LDS tiles (and with s_item)
        if (tid == 0) s_item = nxt;
        __syncthreads();
    }
}

// ---- host side ----
// OPT-IN (MTTS_SK=1; the emulator build defaults to on so that the CPU tests exercise it).  Measured on MI355X (profiles/r03_sk_queue.md):
// on single launches the queue recovers the tail (conv1 forward BK16: 90 -> 100 TFLOP/s) but its tiles run ~10 % slower than the same
// K-loop in a plain grid (PostNet 64x64 BK16: 113 -> 99-104 TFLOP/s; VGPR- and AGPR-form accumulators, 4-6 workgroups per CU alike),
// and the whole 8-task meta-step is 6-9 % slower (171 -> 183-190 ms); the plain grids stay the default.
inline int& gemm_sk_enabled() {
#if defined(MTTS_EMU)
    static int v = [] { const char* e = getenv("MTTS_SK"); return e ? atoi(e) : 1; }();
#else
    static int v = [] { const char* e = getenv("MTTS_SK"); return e ? atoi(e) : 0; }();
#endif
    return v;
}
inline int gemm_sk_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// resident capacity of the chip for the persistent kernel, in workgroups
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
        if (bk == 32) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<32>, 256, 0);
        else hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_sk_kernel<16>, 256, 0);
        // never more workgroups than are resident at once (a queued workgroup would only start when the others are done): the API's
        // answer, capped at 5 (it over-reports by one for SGPR-heavy kernels, MI355X_MICROARCH.md: residency)
        per_cu = std::max(1, std::min(per_cu, gemm_sk_env("MTTS_SK_OCC", 5)));
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
        const int ch = std::max(1, (gemm_keff(p.g) + bk - 1) / bk);
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
    if (work > 1.0e9) return false;   // the in-kernel schedule is 32-bit
    int G = (int)std::min<double>(cap, std::max(1.0, tiles));
    mp.sk.s_max = gemm_sk_env("MTTS_SK_SMAX", 4);
    mp.sk.min_chunks = std::max(1, gemm_sk_env("MTTS_SK_MINCH", 16) * 16 / bk);
    mp.sk.tol_div = std::max(1, gemm_sk_env("MTTS_SK_TOL", 16));
    mp.sk.slabs = (int)std::min<long long>(kSplitWsFloats / 4096, kSplitCtrs);
    mp.sk.ws = cx.wsp.ws; mp.sk.ctr = cx.wsp.ctr;
    mp.sk.head = cx.sk_heads + 8 * (cx.sk_parity & 1);
    mp.sk.head_next = cx.sk_heads + 8 * ((cx.sk_parity + 1) & 1);
    cx.sk_parity ^= 1;
    // a run = the n-tiles of an m-tile of the first (longest) problem, at least 4 and at most 32 items
    mp.sk.run = std::max(4, std::min(32, gemm_tiles_n(q[0].g, q[0].max_N, 64)));
    static const int run_env = gemm_sk_env("MTTS_SK_RUN", 0);
    if (run_env > 0) mp.sk.run = run_env;
    for (int i = 0; i < mp.n; ++i) { mp.g[i].splitk = 1; mp.g[i].swizzle = 0; }
    dim3 block(256), grid((unsigned)G, 1, 1);
    if (bk == 32) { MTTS_LAUNCH((gemm_sk_kernel<32>), grid, block, stream, mp); }
    else { MTTS_LAUNCH((gemm_sk_kernel<16>), grid, block, stream, mp); }
    cx.last_kind = bk == 32 ? GK_SK32 : GK_SK16;
    return true;
}

}  // namespace mtts
