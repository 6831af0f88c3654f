// HBM-bound row / column kernels of the FastSpeech2 hot path (everything that is not a
// contraction): embedding + sinusoid add, residual + LayerNorm (+ row mask) forward/backward,
// softmax forward/backward over per-sequence score matrices, variance-adaptor glue (speaker
// add, bucketize + embedding add, 256->1 projections), length regulator gather / segment-sum,
// PostNet BatchNorm(+tanh), the 5-term loss and its gradient, deterministic embedding-table
// gradients, MAML SGD update and the fused clip + Adam outer update.
//
// Conventions: activations are row matrices [rows][C] (C % 4 == 0) in a "row space" (see
// engine.h); blockIdx.z is the task of the meta-batch; every pointer comes with a per-task
// stride (`*_ts`, elements).  Row kernels use one 64-lane wavefront per row (float4 per lane,
// DPP/shuffle reductions); column reductions use one workgroup per 32-column stripe so results
// are deterministic (no float atomics anywhere).
#pragma once
#include "compat.h"

namespace mtts {

struct TaskMeta {  // one per task, device-visible (int fields only; indexed as int[8])
    int B, Smax, Tcap, Mp, Mf, Mr, nP, nF;
};
enum { META_B = 0, META_SMAX, META_TCAP, META_MP, META_MF, META_MR, META_NP, META_NF, META_STRIDE = 8 };

#define ROW_PROLOGUE(Mfield)                                              \
    const int z = blockIdx.z;                                             \
    const int M_ = meta[z * META_STRIDE + (Mfield)];                      \
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6);             \
    const int lane = (int)threadIdx.x & 63;                               \
    if (row >= M_) return;

inline dim3 row_grid(int max_rows, int tasks) { return dim3((unsigned)((max_rows + 3) / 4), 1, (unsigned)tasks); }

// ------------------------------------------------------------------------------------------
// encoder input: word embedding (padding row 0) + sinusoid positions (transformer/Models.py:89-91)
// ------------------------------------------------------------------------------------------
__global__ void embed_pos_kernel(const int* meta, int mfield, float* out, long long out_ts, const float* emb,
                                 long long emb_ts, const float* pos, const int* tok, const int* row_t,
                                 const unsigned char* valid, long long row_ts, int C) {
    ROW_PROLOGUE(mfield)
    const long long r = (long long)z * row_ts + row;
    float* o = out + (long long)z * out_ts + (long long)row * C;
    const bool v = valid[r] != 0;
    const float* e = emb + (long long)z * emb_ts + (long long)(v ? tok[r] : 0) * C;
    const float* p = pos + (long long)(v ? row_t[r] : 0) * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 a = zero4();
        if (v) { const float4 x = ld4(e + c), y = ld4(p + c); a = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
        st4(o + c, a);
    }
}

// ------------------------------------------------------------------------------------------
// y = mask ? LayerNorm(a (+ res)) : 0   (SubLayers.py:55,91 post-LN; modules.py:222,234)
// z_out (optional) receives a + res; stats = (mean, rstd) per row.  C <= 1024.
// ------------------------------------------------------------------------------------------
// (the counter-based dropout mask — splitmix64 / DropSpec / drop4 — lives in compat.h: the GEMM's row-complete LayerNorm epilogue applies it too)

// Two rows per wavefront (8 per workgroup): all loads of both rows are in flight before the first reduction — these
// launches are only a couple of wave "rounds" long, so what matters is the number of serialised memory round trips per
// wave (was: mask -> rows -> stats), not bandwidth.
#define ROW2_PROLOGUE(Mfield)                                                                   \
    const int z = blockIdx.z;                                                                   \
    const int M_ = meta[z * META_STRIDE + (Mfield)];                                            \
    const int lane = (int)threadIdx.x & 63;                                                     \
    const int row0 = blockIdx.x * 8 + ((int)threadIdx.x >> 6) * 2;                              \
    if (row0 >= M_) return;                                                                     \
    const int rows_[2] = {row0, row0 + 1 < M_ ? row0 + 1 : row0};                               \
    const bool live1 = row0 + 1 < M_;

// LayerNorm-family launch: the kernel instantiation that holds a row of C channels in registers (NV = 1 or 4 float4 per lane)
#define MTTS_LAUNCH_LN(kernel, C_, grid, block, stream, ...)                                           \
    do {                                                                                               \
        if ((C_) <= 256) { MTTS_LAUNCH((kernel<1>), grid, block, stream, __VA_ARGS__); }               \
        else { MTTS_LAUNCH((kernel<4>), grid, block, stream, __VA_ARGS__); }                           \
    } while (0)

// four values of a row and their bf16 twin (the operand plane a GEMM of the bf16 numerics mode reads instead: gemm_bf16.h)
__device__ __forceinline__ void st4_bf16(bf16_t* p, float4 v) {
    const unsigned lo = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16), hi = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<unsigned long long*>(p) = (unsigned long long)lo | ((unsigned long long)hi << 32);
}

inline dim3 row2_grid(int max_rows, int tasks) { return dim3((unsigned)((max_rows + 7) / 8), 1, (unsigned)tasks); }

// NV = float4 a lane holds per row (1: C <= 256, 4: C <= 1024): a compile-time bound keeps the row in VGPRs (with a runtime
// trip count the arrays below are indexed dynamically and live in scratch memory)
template <int NV>
__global__ void layernorm_fwd_kernel(const int* meta, int mfield, const float* a, long long a_ts, const float* res,
                                     long long res_ts, const float* gamma, const float* beta, long long par_ts,
                                     const unsigned char* mask, long long mask_ts, float* z_out, long long z_ts,
                                     float* y, long long y_ts, float* stats, long long st_ts, int C, float eps,
                                     DropSpec din, DropSpec dout, bf16_t* yh) {
    // yh (optional): bf16 twin of y, same layout
    // din: dropout applied to `a` before the residual add (self.dropout(sublayer(x)) + residual, SubLayers.py:54-55,90-91);
    // dout: dropout applied to the normalised output (LayerNorm -> Dropout of the variance predictors, modules.py:222-235)
    ROW2_PROLOGUE(mfield)
    float4 v[2][NV];
    float s[2] = {0.f, 0.f};
    bool keep[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = rows_[q];
        const float* pa = a + (long long)z * a_ts + (long long)row * C;
        const float* pr = res ? res + (long long)z * res_ts + (long long)row * C : nullptr;
        keep[q] = mask ? (mask[(long long)z * mask_ts + row] != 0) : true;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = lane * 4 + 256 * n;
            v[q][n] = zero4();
            if (c < C) {
                float4 x = ld4(pa + c);
                if (din.thr16) x = drop4(din, z, row, C, c, x);
                if (pr) { const float4 r4 = ld4(pr + c); x = make_float4(x.x + r4.x, x.y + r4.y, x.z + r4.z, x.w + r4.w); }
                v[q][n] = x;
                s[q] += (x.x + x.y) + (x.z + x.w);
            }
        }
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        mean[q] = wave_sum(s[q]) / (float)C;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane * 4 + 256 * i < C) {
                const float dx = v[q][i].x - mean[q], dy = v[q][i].y - mean[q], dz = v[q][i].z - mean[q], dw = v[q][i].w - mean[q];
                t += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        rstd[q] = rsqrtf(wave_sum(t) / (float)C + eps);
    }
    const float* g = gamma + (long long)z * par_ts;
    const float* b = beta + (long long)z * par_ts;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q == 1 && !live1) break;
        const int row = rows_[q];
        float* py = y + (long long)z * y_ts + (long long)row * C;
        float* pz = z_out ? z_out + (long long)z * z_ts + (long long)row * C : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c >= C) continue;
            if (pz) st4(pz + c, v[q][i]);
            float4 o = zero4();
            if (keep[q]) {
                const float4 g4 = ld4(g + c), b4 = ld4(b + c);
                o = make_float4((v[q][i].x - mean[q]) * rstd[q] * g4.x + b4.x, (v[q][i].y - mean[q]) * rstd[q] * g4.y + b4.y,
                                (v[q][i].z - mean[q]) * rstd[q] * g4.z + b4.z, (v[q][i].w - mean[q]) * rstd[q] * g4.w + b4.w);
                if (dout.thr16) o = drop4(dout, z, row, C, c, o);
            }
            st4(py + c, o);
            if (yh) st4_bf16(yh + (long long)z * y_ts + (long long)row * C + c, o);
        }
        if (lane == 0) {
            float* st = stats + (long long)z * st_ts + (long long)row * 2;
            st[0] = mean[q];
            st[1] = rstd[q];
        }
    }
}

// dz = mask ? rstd * (g - mean(g) - xhat * mean(g * xhat)) : 0, g = dy * gamma
// (relu_on_z additionally multiplies by [z > 0]: z = ReLU(conv) in the variance predictors)
// The kernel also does STAGE 1 of the gamma / beta gradient reduction (dgamma = sum_rows dy * xhat, dbeta = sum_rows dy over the unmasked
// rows): every workgroup folds its 8 rows in registers and LDS and writes one partial row pair — partial[task][blockIdx.x][0 | 1][C], the
// layout colfinal_kernel folds with rows_per_chunk = kLnRows — so the separate colpart launch over the same dy / z is gone.
// din: dropout applied to the incoming gradient on load (the variance predictors' F.dropout sits behind the LayerNorm, modules.py:222-235:
// its backward used to be a dropout launch of its own in front of this kernel).
constexpr int kLnRows = 8;   // rows per workgroup of the LayerNorm kernels = rows per partial chunk
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const int* meta, int mfield, const float* dy, long long dy_ts, const float* zin,
                                     long long z_ts, const float* stats, long long st_ts, const float* gamma,
                                     long long par_ts, const unsigned char* mask, long long mask_ts, float* dz,
                                     long long dz_ts, int C, int relu_on_z, float* dz_drop, long long dzd_ts, DropSpec dd,
                                     DropSpec din, float* partial, int max_chunks, bf16_t* twin, int twin_sel) {
    // twin / twin_sel: bf16 twin of dz (1) or of dz_drop (2), same layout as its fp32 buffer; 0: none
    // dz_drop (optional): dropout(dz) with the mask of the forward site — the gradient entering the dropped branch, while dz
    // itself continues along the residual path
    __shared__ __attribute__((aligned(16))) float red[4][2][256 * NV];
    const int z = blockIdx.z;
    const int M_ = meta[z * META_STRIDE + mfield];
    if ((int)blockIdx.x * kLnRows >= M_) return;                       // (whole workgroup)
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int row0 = blockIdx.x * kLnRows + wave * 2;
    const bool live0 = row0 < M_, live1 = row0 + 1 < M_;
    const int rows_[2] = {live0 ? row0 : M_ - 1, live1 ? row0 + 1 : (live0 ? row0 : M_ - 1)};
    const float* g = gamma + (long long)z * par_ts;
    float4 gv[2][NV], xh[2][NV], pg[NV], pb[NV];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rstd[2];
    bool keep[2];
#pragma unroll
    for (int n = 0; n < NV; ++n) { pg[n] = zero4(); pb[n] = zero4(); }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = rows_[q];
        keep[q] = mask ? (mask[(long long)z * mask_ts + row] != 0) : true;
        const float wq = (keep[q] && (q == 0 ? live0 : live1)) ? 1.f : 0.f;   // this row counts in the parameter gradients
        const float* pdy = dy + (long long)z * dy_ts + (long long)row * C;
        const float* pz = zin + (long long)z * z_ts + (long long)row * C;
        const float* st = stats + (long long)z * st_ts + (long long)row * 2;
        const float mean = st[0];
        rstd[q] = st[1];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = lane * 4 + 256 * n;
            gv[q][n] = zero4(); xh[q][n] = zero4();
            if (c >= C) continue;
            float4 d = ld4(pdy + c);
            const float4 x = ld4(pz + c), g4 = ld4(g + c);
            if (din.thr16) d = drop4(din, z, row, C, c, d);
            gv[q][n] = make_float4(d.x * g4.x, d.y * g4.y, d.z * g4.z, d.w * g4.w);
            xh[q][n] = make_float4((x.x - mean) * rstd[q], (x.y - mean) * rstd[q], (x.z - mean) * rstd[q], (x.w - mean) * rstd[q]);
            s1[q] += (gv[q][n].x + gv[q][n].y) + (gv[q][n].z + gv[q][n].w);
            s2[q] += (gv[q][n].x * xh[q][n].x + gv[q][n].y * xh[q][n].y) + (gv[q][n].z * xh[q][n].z + gv[q][n].w * xh[q][n].w);
            pg[n].x += wq * d.x * xh[q][n].x; pg[n].y += wq * d.y * xh[q][n].y; pg[n].z += wq * d.z * xh[q][n].z; pg[n].w += wq * d.w * xh[q][n].w;
            pb[n].x += wq * d.x; pb[n].y += wq * d.y; pb[n].z += wq * d.z; pb[n].w += wq * d.w;
        }
    }
    if (partial) {
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = lane * 4 + 256 * n;
            if (c < C) { st4(&red[wave][0][c], pg[n]); st4(&red[wave][1][c], pb[n]); }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float m1 = wave_sum(s1[q]) / (float)C, m2 = wave_sum(s2[q]) / (float)C;
        if (!(q == 0 ? live0 : live1)) continue;
        const int row = rows_[q];
        float* pd = dz + (long long)z * dz_ts + (long long)row * C;
        const float* pz = zin + (long long)z * z_ts + (long long)row * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c >= C) continue;
            float4 o = make_float4(rstd[q] * (gv[q][i].x - m1 - xh[q][i].x * m2), rstd[q] * (gv[q][i].y - m1 - xh[q][i].y * m2),
                                   rstd[q] * (gv[q][i].z - m1 - xh[q][i].z * m2), rstd[q] * (gv[q][i].w - m1 - xh[q][i].w * m2));
            if (relu_on_z) {  // z is a ReLU output: pass the gradient only where the pre-activation was > 0
                const float4 x = ld4(pz + c);
                o = make_float4(x.x > 0.f ? o.x : 0.f, x.y > 0.f ? o.y : 0.f, x.z > 0.f ? o.z : 0.f, x.w > 0.f ? o.w : 0.f);
            }
            if (!keep[q]) o = zero4();
            st4(pd + c, o);
            if (twin_sel == 1) st4_bf16(twin + (long long)z * dz_ts + (long long)row * C + c, o);
            if (dz_drop) {
                const float4 od = drop4(dd, z, row, C, c, o);
                st4(dz_drop + (long long)z * dzd_ts + (long long)row * C + c, od);
                if (twin_sel == 2) st4_bf16(twin + (long long)z * dzd_ts + (long long)row * C + c, od);
            }
        }
    }
    if (!partial) return;
    __syncthreads();
    float* out = partial + ((long long)z * max_chunks + blockIdx.x) * 3 * C;   // (wave order: fixed => run-to-run identical)
    for (int idx = (int)threadIdx.x; idx < 2 * C; idx += 256) {
        const int k = idx >= C ? 1 : 0, c = idx - k * C;
        out[(long long)k * C + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// ------------------------------------------------------------------------------------------
// Column reductions over rows, two deterministic stages (no float atomics):
//   stage 1  colpart_kernel: grid (C/128, row chunks of 128, tasks); 8 row lanes x 32 float4 column
//            groups per workgroup -> partial[task][chunk][k][C]
//   stage 2  colfinal_kernel: one thread per column folds the chunks in order.
// modes: 0  out0[c] = sum_m w(m) X[m][c]                                   (bias / 256->1 weight grads)
//        1  out0 = sum dy*xhat (dgamma), out1 = sum dy (dbeta), xhat from per-row LayerNorm stats
//        2  BatchNorm batch statistics over masked rows: per-chunk (count, mean, M2) merged with
//           Chan's formula -> out0 = [mean | rstd | unbiased var]
//        3  BatchNorm backward sums: out0 = sum dpre*xhat, out1 = sum dpre, dpre = dy*(1-y^2) if tanh,
//           xhat from per-column stats [mean | rstd]
// ------------------------------------------------------------------------------------------
constexpr int kRC = 32;   // rows per chunk (small chunks: 4 row iterations per thread, ~1000+ workgroups per launch)

struct ColArgs {
    const float* X = nullptr; long long x_ts = 0;      // primary operand [M][C] (dy for modes 1, 3)
    const float* Z = nullptr; long long z_ts = 0;      // mode 1: LN input; mode 3: pre-BN conv output
    const float* Y = nullptr; long long y_ts = 0;      // mode 3: post-tanh activation
    const float* stats = nullptr; long long st_ts = 0; // mode 1: per-row (mean, rstd); mode 3: per-column [mean|rstd]
    const unsigned char* mask = nullptr; long long mask_ts = 0;
    const float* roww = nullptr; long long roww_ts = 0;
    // tangent modes (second-order MAML, tangent.h):
    //   5  LayerNorm: out0 = sum (X xhat + X2 t_xhat), out1 = sum X     X = tg_y, X2 = dy, Z2 = tz, stats2 = per-row (m1, m2)
    //   6  BatchNorm: out0 = sum (tg xhat + g t_xhat), out1 = sum tg    X = tg_dy, X2 = dy, Y2 = ta, Z2 = tc, stats2 = per-column [S1 | S0]
    const float* X2 = nullptr; long long x2_ts = 0;
    const float* Z2 = nullptr; long long z2_ts = 0;
    const float* Y2 = nullptr; long long y2_ts = 0;
    const float* stats2 = nullptr; long long st2_ts = 0;
    float yscale = 1.f;  // Y holds dropout(tanh(.)): y = Y * yscale with yscale = 1 - p (modes 3, 6)
    DropSpec xdrop;      // mode 3: dropout applied to X (= dY) on load — the backward of the dropout behind the BatchNorm layer; modes 5 / 6: to X and X2 (both incoming gradients of a LayerNorm / BatchNorm tangent backward)
    int C = 0, mode = 0, do_tanh = 0, mfield = 0, accumulate = 0;
};

// CG column groups of 4 floats x RL row lanes per 256-thread workgroup.  STRIPE = false: rows [chunk * kRC, +kRC) of a 128-column
// group -> partial sums (stage 1 of the two-stage reduction).  STRIPE = true: ALL rows of a 32-column stripe in one workgroup,
// final values written directly (one launch instead of two, no cross-workgroup hand-off): the per-task row counts of this workload
// (<= ~5 k) keep a stripe at a few hundred KB, and a task's column count / 32 x tasks workgroups still spread over the chip.
template <int CG, int RL, bool STRIPE>
__device__ __forceinline__ void col_body(const int* meta, const ColArgs& a, float* partial, int max_chunks, float* out0, float* out1,
                                         long long out_ts, float eps) {
    static_assert(CG * RL == 256, "256 threads");
    __shared__ __attribute__((aligned(16))) float red[3][RL][CG * 4 + 4];
    const int z = blockIdx.z, chunk = blockIdx.y;
    const int M_ = meta[z * META_STRIDE + a.mfield];
    const int r0 = STRIPE ? 0 : chunk * kRC;
    if (r0 >= M_) return;
    const int r1 = STRIPE ? M_ : ((r0 + kRC < M_) ? r0 + kRC : M_);
    const int cg = (int)threadIdx.x % CG, ry = (int)threadIdx.x / CG;
    const int c = blockIdx.x * (CG * 4) + cg * 4;
    const bool cin = c < a.C;
    const int C = a.C;
    const float* px = a.X + (long long)z * a.x_ts;
    const float* pz = a.Z ? a.Z + (long long)z * a.z_ts : nullptr;
    const float* py = a.Y ? a.Y + (long long)z * a.y_ts : nullptr;
    const float* ps = a.stats ? a.stats + (long long)z * a.st_ts : nullptr;
    const unsigned char* pm = a.mask ? a.mask + (long long)z * a.mask_ts : nullptr;
    const float* pw = a.roww ? a.roww + (long long)z * a.roww_ts : nullptr;
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    // C may be < 4-aligned only for C == 1 (scalar bias of the 256->1 projections)
    const bool vec = (C & 3) == 0;
    auto ldv = [&](const float* base, int m, float (&v)[4]) {
        if (vec) { const float4 t = ld4(base + (long long)m * C + c); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else { for (int k = 0; k < 4; ++k) v[k] = (c + k < C) ? base[(long long)m * C + c + k] : 0.f; }
    };
    if (cin) {
        if (a.mode == 2) {
            float mean[4] = {0.f, 0.f, 0.f, 0.f};
            for (int m = r0 + ry; m < r1; m += RL) if (!pm || pm[m]) { float v[4]; ldv(px, m, v); for (int k = 0; k < 4; ++k) acc0[k] += v[k]; cnt += 1.f; }
            (void)mean;
        } else {
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f}, q1[4] = {0.f, 0.f, 0.f, 0.f}, q0[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.mode == 3 || a.mode == 6) { ldv(ps, 0, mu); float t[4]; const float* p2 = ps + C; ldv(p2, 0, t); for (int k = 0; k < 4; ++k) rs[k] = t[k]; }
            const float* px2 = a.X2 ? a.X2 + (long long)z * a.x2_ts : nullptr;
            const float* pz2 = a.Z2 ? a.Z2 + (long long)z * a.z2_ts : nullptr;
            const float* py2 = a.Y2 ? a.Y2 + (long long)z * a.y2_ts : nullptr;
            const float* ps2 = a.stats2 ? a.stats2 + (long long)z * a.st2_ts : nullptr;
            if (a.mode == 6) { ldv(ps2, 0, q1); const float* p3 = ps2 + C; ldv(p3, 0, q0); }
            const float inv_n6 = 1.f / (float)(meta[z * META_STRIDE + META_B] * meta[z * META_STRIDE + META_TCAP]);
            for (int m = r0 + ry; m < r1; m += RL) {
                if (pm && !pm[m]) continue;
                float x[4]; ldv(px, m, x);
                if (a.mode == 0) {
                    const float w = pw ? pw[m] : 1.f;
                    for (int k = 0; k < 4; ++k) acc0[k] += w * x[k];
                } else if (a.mode == 1) {
                    float zz[4]; ldv(pz, m, zz);
                    const float mean = ps[2 * m], rstd = ps[2 * m + 1];
                    for (int k = 0; k < 4; ++k) { acc0[k] += x[k] * (zz[k] - mean) * rstd; acc1[k] += x[k]; }
                } else if (a.mode == 3) {
                    float zz[4]; ldv(pz, m, zz);
                    if (a.xdrop.thr16) { const float4 t = drop4(a.xdrop, z, m, C, c, make_float4(x[0], x[1], x[2], x[3])); x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
                    if (a.do_tanh) { float y[4]; ldv(py, m, y); for (int k = 0; k < 4; ++k) { const float yy = y[k] * a.yscale; x[k] *= (1.f - yy * yy); } }
                    for (int k = 0; k < 4; ++k) { acc0[k] += x[k] * (zz[k] - mu[k]) * rs[k]; acc1[k] += x[k]; }
                } else if (a.mode == 5) {
                    float zz[4], dy[4], tz[4]; ldv(pz, m, zz); ldv(px2, m, dy); ldv(pz2, m, tz);
                    if (a.xdrop.thr16) {   // the dropout behind the LayerNorm (variance predictors): its backward masks both incoming gradients on load
                        const float4 t0 = drop4(a.xdrop, z, m, C, c, make_float4(x[0], x[1], x[2], x[3])), t1 = drop4(a.xdrop, z, m, C, c, make_float4(dy[0], dy[1], dy[2], dy[3]));
                        x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; dy[0] = t1.x; dy[1] = t1.y; dy[2] = t1.z; dy[3] = t1.w;
                    }
                    const float mean = ps[2 * m], rstd = ps[2 * m + 1], m1 = ps2[2 * m], m2 = ps2[2 * m + 1];
                    for (int k = 0; k < 4; ++k) {
                        const float xh = (zz[k] - mean) * rstd, txh = rstd * (tz[k] - m1 - xh * m2);
                        acc0[k] += x[k] * xh + dy[k] * txh;
                        acc1[k] += x[k];
                    }
                } else {  // mode 6
                    float zz[4], dy[4], tc[4]; ldv(pz, m, zz); ldv(px2, m, dy); ldv(pz2, m, tc);
                    if (a.xdrop.thr16) {   // the dropout behind the BatchNorm layer: its backward masks both incoming gradients on load (as mode 3 does for X)
                        const float4 t0 = drop4(a.xdrop, z, m, C, c, make_float4(x[0], x[1], x[2], x[3])), t1 = drop4(a.xdrop, z, m, C, c, make_float4(dy[0], dy[1], dy[2], dy[3]));
                        x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; dy[0] = t1.x; dy[1] = t1.y; dy[2] = t1.z; dy[3] = t1.w;
                    }
                    float g[4], tg[4];
                    for (int k = 0; k < 4; ++k) { g[k] = dy[k]; tg[k] = x[k]; }
                    if (a.do_tanh) {
                        float y[4], ty[4]; ldv(py, m, y); ldv(py2, m, ty);
                        for (int k = 0; k < 4; ++k) { const float yy = y[k] * a.yscale, tyy = ty[k] * a.yscale; const float sq = 1.f - yy * yy; tg[k] = x[k] * sq - 2.f * yy * tyy * dy[k]; g[k] = dy[k] * sq; }
                    }
                    for (int k = 0; k < 4; ++k) {
                        const float xh = (zz[k] - mu[k]) * rs[k];
                        const float txh = rs[k] * (tc[k] - q0[k] * inv_n6 - xh * q1[k] * inv_n6);
                        acc0[k] += tg[k] * xh + g[k] * txh;
                        acc1[k] += tg[k];
                    }
                }
            }
        }
    }
    for (int k = 0; k < 4; ++k) { red[0][ry][cg * 4 + k] = acc0[k]; red[1][ry][cg * 4 + k] = acc1[k]; }
    if (cg == 0) red[2][ry][0] = cnt;
    __syncthreads();
    float* out = STRIPE ? nullptr : partial + ((long long)z * max_chunks + chunk) * 3 * C;
    if (a.mode != 2) {
        if (ry < 2) {  // row lane 0 folds acc0, row lane 1 folds acc1 (row-lane order: fixed, run-to-run identical)
            for (int k = 0; k < 4; ++k) {
                if (c + k >= C) break;
                float s0 = 0.f;
                for (int i = 0; i < RL; ++i) s0 += red[ry][i][cg * 4 + k];
                if (!STRIPE) { out[(long long)ry * C + c + k] = s0; continue; }
                float* o = ry == 0 ? out0 : out1;
                if (!o) continue;
                if (a.accumulate) s0 += o[(long long)z * out_ts + c + k];
                o[(long long)z * out_ts + c + k] = s0;
            }
        }
        return;
    }
    // mode 2: chunk mean, then chunk M2 (second pass over the same rows)
    float n = 0.f;
    for (int i = 0; i < RL; ++i) n += red[2][i][0];
    float mean[4];
    for (int k = 0; k < 4; ++k) { float s0 = 0.f; for (int i = 0; i < RL; ++i) s0 += red[0][i][cg * 4 + k]; mean[k] = n > 0.f ? s0 / n : 0.f; }
    float m2[4] = {0.f, 0.f, 0.f, 0.f};
    if (cin)
        for (int m = r0 + ry; m < r1; m += RL) if (!pm || pm[m]) { float v[4]; ldv(px, m, v); for (int k = 0; k < 4; ++k) { const float d = v[k] - mean[k]; m2[k] += d * d; } }
    __syncthreads();
    for (int k = 0; k < 4; ++k) red[1][ry][cg * 4 + k] = m2[k];
    __syncthreads();
    if (ry == 0 && cin)
        for (int k = 0; k < 4; ++k) {
            if (c + k >= C) break;
            float s = 0.f;
            for (int i = 0; i < RL; ++i) s += red[1][i][cg * 4 + k];
            if (!STRIPE) { out[c + k] = n; out[(long long)C + c + k] = mean[k]; out[2LL * C + c + k] = s; continue; }
            float* so = out0 + (long long)z * out_ts;   // BatchNorm statistics: [mean | rstd | unbiased var]
            so[c + k] = mean[k];
            so[C + c + k] = rsqrtf(s / n + eps);
            so[2 * C + c + k] = s / fmaxf(n - 1.f, 1.f);
        }
}
__device__ __forceinline__ void colpart_body(const int* meta, const ColArgs& a, float* partial, int max_chunks) {
    col_body<32, 8, false>(meta, a, partial, max_chunks, nullptr, nullptr, 0, 0.f);
}

// 16 columns x 16 chunk lanes per workgroup (256 threads): lane q folds chunks q, q + 16, ... of its column — four loads in flight per
// step, added in chunk order — and the sixteen lane results are merged in lane order through LDS (fixed order => run-to-run identical).
// (The first version ran 64 columns x 4 lanes: a single-task LayerNorm fold is 245 chunks deep, 61 dependent L2 round trips per lane and
// 4-16 workgroups per launch — 14-18 us a launch, 8 % of a single-task rank's kernel time.)
constexpr int kCfCols = 16, kCfLanes = 16;
inline int colfinal_blocks(int C) { return (C + kCfCols - 1) / kCfCols; }
__device__ __forceinline__ void colfinal_fold(const int* meta, int mfield, int mode, const float* partial, int max_chunks, int C,
                                              float* out0, float* out1, long long out_ts, float eps, int accumulate, int c_base, int rows_per_chunk = kRC) {
    __shared__ float red[3][kCfLanes][kCfCols];
    const int z = blockIdx.z, cl = (int)threadIdx.x % kCfCols, q = (int)threadIdx.x / kCfCols;
    const int c = c_base + cl;
    const bool cin = c < C;
    const int M_ = meta[z * META_STRIDE + mfield];
    const int nch = (M_ + rows_per_chunk - 1) / rows_per_chunk;
    const float* p = partial + (long long)z * max_chunks * 3 * C + (cin ? c : 0);
    const long long cs = 3LL * C;   // floats per chunk
    if (mode != 2) {
        float s0 = 0.f, s1 = 0.f;
        if (cin) {
            int i = q;
            for (; i + 3 * kCfLanes < nch; i += 4 * kCfLanes) {
                const float* r = p + i * cs;
                const float a0 = r[0], a1 = r[kCfLanes * cs], a2 = r[2 * kCfLanes * cs], a3 = r[3 * kCfLanes * cs];
                const float b0 = r[C], b1 = r[kCfLanes * cs + C], b2 = r[2 * kCfLanes * cs + C], b3 = r[3 * kCfLanes * cs + C];
                s0 = (((s0 + a0) + a1) + a2) + a3;
                s1 = (((s1 + b0) + b1) + b2) + b3;
            }
            for (; i < nch; i += kCfLanes) { s0 += p[i * cs]; s1 += p[i * cs + C]; }
        }
        red[0][q][cl] = s0; red[1][q][cl] = s1;
        __syncthreads();
        if (q != 0 || !cin) return;
        s0 = red[0][0][cl]; s1 = red[1][0][cl];
#pragma unroll
        for (int i = 1; i < kCfLanes; ++i) { s0 += red[0][i][cl]; s1 += red[1][i][cl]; }
        if (accumulate) { s0 += out0[(long long)z * out_ts + c]; if (out1) s1 += out1[(long long)z * out_ts + c]; }
        out0[(long long)z * out_ts + c] = s0;
        if (out1) out1[(long long)z * out_ts + c] = s1;
        return;
    }
    float n = 0.f, mean = 0.f, m2 = 0.f;  // Chan et al. pairwise merge
    auto merge = [&](float nb, float mb, float sb) {
        if (nb <= 0.f) return;
        const float nn = n + nb, d = mb - mean;
        mean += d * nb / nn;
        m2 += sb + d * d * n * nb / nn;
        n = nn;
    };
    if (cin) {
        int i = q;
        for (; i + kCfLanes < nch; i += 2 * kCfLanes) {
            const float* r = p + i * cs;
            const float n0 = r[0], m0 = r[C], q0 = r[2 * C], n1 = r[kCfLanes * cs], m1 = r[kCfLanes * cs + C], q1 = r[kCfLanes * cs + 2 * C];
            merge(n0, m0, q0);
            merge(n1, m1, q1);
        }
        for (; i < nch; i += kCfLanes) merge(p[i * cs], p[i * cs + C], p[i * cs + 2 * C]);
    }
    red[0][q][cl] = n; red[1][q][cl] = mean; red[2][q][cl] = m2;
    __syncthreads();
    if (q != 0 || !cin) return;
    n = 0.f; mean = 0.f; m2 = 0.f;
    for (int i = 0; i < kCfLanes; ++i) merge(red[0][i][cl], red[1][i][cl], red[2][i][cl]);
    float* so = out0 + (long long)z * out_ts;
    so[c] = mean;
    so[C + c] = rsqrtf(m2 / n + eps);
    so[2 * C + c] = m2 / fmaxf(n - 1.f, 1.f);
}

__global__ void colpart_kernel(const int* meta, ColArgs a, float* partial, int max_chunks) { colpart_body(meta, a, partial, max_chunks); }

// rows_per_chunk: kRC for colpart_kernel's partials, kLnRows for the ones layernorm_bwd_kernel writes in passing
__global__ void colfinal_kernel(const int* meta, int mfield, int mode, const float* partial, int max_chunks, int C,
                                float* out0, float* out1, long long out_ts, float eps, int accumulate, int rows_per_chunk) {
    colfinal_fold(meta, mfield, mode, partial, max_chunks, C, out0, out1, out_ts, eps, accumulate, blockIdx.x * kCfCols, rows_per_chunk);
}

// ------------------------------------------------------------------------------------------
// attention softmax over per-(task, sequence, head) score matrices (Modules.py:16-22).  Only the
// L valid query rows / key columns are ever computed (padded keys get -inf in the reference,
// padded query rows are zeroed by FFTBlock's masked_fill), so no mask tensor exists here.
// ------------------------------------------------------------------------------------------
struct AttnSeq { long long s_off; int L, ldS; };

// One wavefront per row, the row lives in registers (L <= 1024: 16 values per lane) between the single read and the single
// write; longer rows (eval mode beyond max_seq_len) take the streaming path.
__global__ void softmax_fwd_kernel(const AttnSeq* seqs, float* S) {
    const AttnSeq q = seqs[blockIdx.z];
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= q.L) return;
    float* p = S + q.s_off + (long long)row * q.ldS;
    if (q.L <= 1024) {
        float v[16];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; v[k] = c < q.L ? p[c] : -3.0e38f; mx = fmaxf(mx, v[k]); }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; v[k] = c < q.L ? expf(v[k] - mx) : 0.f; s += v[k]; }
        s = wave_sum(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; if (c < q.ldS) p[c] = v[k] * inv; }
        return;
    }
    float mx = -3.0e38f;
    for (int c = lane; c < q.L; c += 64) mx = fmaxf(mx, p[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < q.L; c += 64) { const float e = expf(p[c] - mx); p[c] = e; s += e; }
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < q.ldS; c += 64) p[c] = (c < q.L) ? p[c] * inv : 0.f;
}

// in place on dP: dS = alpha * P * (dP - sum_j dP_j P_j); pad columns zeroed
__global__ void softmax_bwd_kernel(const AttnSeq* seqs, const float* P, float* dP, float alpha) {
    const AttnSeq q = seqs[blockIdx.z];
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= q.L) return;
    const float* p = P + q.s_off + (long long)row * q.ldS;
    float* d = dP + q.s_off + (long long)row * q.ldS;
    if (q.L <= 1024) {
        float pv[16], dv[16];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = lane + 64 * k;
            pv[k] = c < q.L ? p[c] : 0.f;
            dv[k] = c < q.L ? d[c] : 0.f;
            s += dv[k] * pv[k];
        }
        s = wave_sum(s);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; if (c < q.ldS) d[c] = alpha * pv[k] * (dv[k] - s); }
        return;
    }
    float s = 0.f;
    for (int c = lane; c < q.L; c += 64) s += d[c] * p[c];
    s = wave_sum(s);
    for (int c = lane; c < q.ldS; c += 64) d[c] = (c < q.L) ? alpha * p[c] * (d[c] - s) : 0.f;
}

// ------------------------------------------------------------------------------------------
// speaker vectors (speaker_encoder.py:62-65; base_adaptor.py:64-70 mean over the support ids)
// ------------------------------------------------------------------------------------------
__global__ void speaker_vec_kernel(const int* meta, const float* table, long long table_ts, const int* ids,
                                   long long ids_ts, int n_ids_max, int average, float* spk, long long spk_ts,
                                   int C) {
    const int z = blockIdx.z, b = blockIdx.x, B = meta[z * META_STRIDE + META_B];
    if (b >= B) return;
    const float* t = table + (long long)z * table_ts;
    const int* id = ids + (long long)z * ids_ts;
    float* o = spk + (long long)z * spk_ts + (long long)b * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v;
        if (average) {
            const int n = id[n_ids_max];  // count stored after the ids
            v = 0.f;
            for (int j = 0; j < n; ++j) v += t[(long long)id[j] * C + c];
            v /= (float)n;
        } else {
            v = t[(long long)id[b] * C + c];
        }
        o[c] = v;
    }
}

// out[row] = inrect ? x[row] + vec[row_b[row]] : 0      (fastspeech2.py:65-68)
__global__ void add_rowvec_kernel(const int* meta, int mfield, const float* x, long long x_ts, const float* vec,
                                  long long vec_ts, const int* row_b, const unsigned char* inrect, long long row_ts,
                                  float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const long long r = (long long)z * row_ts + row;
    const bool in = inrect[r] != 0;
    const float* px = x + (long long)z * x_ts + (long long)row * C;
    const float* pv = vec + (long long)z * vec_ts + (long long)(in ? row_b[r] : 0) * C;
    float* po = out + (long long)z * out_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 o = zero4();
        if (in) { const float4 a = ld4(px + c), b = ld4(pv + c); o = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
        st4(po + c, o);
    }
}

// torch.bucketize(v, bins) (right=False): number of boundaries strictly below v (modules.py:83,94)
__device__ __forceinline__ int bucketize(float v, const float* bins, int nb) {
    int lo = 0, hi = nb;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// x_out = x_in + table[bucketize(val * control)] on in-rect rows; idx saved for backward
__global__ void bucket_embed_add_kernel(const int* meta, int mfield, const float* x, long long x_ts,
                                        const float* val, long long val_ts, float control, const float* bins, int nb,
                                        const float* table, long long table_ts, const unsigned char* inrect,
                                        long long row_ts, int* idx_out, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const long long r = (long long)z * row_ts + row;
    const bool in = inrect[r] != 0;
    const int idx = in ? bucketize(val[(long long)z * val_ts + row] * control, bins, nb) : 0;
    if (lane == 0) idx_out[r] = in ? idx : -1;
    const float* px = x + (long long)z * x_ts + (long long)row * C;
    const float* pt = table + (long long)z * table_ts + (long long)idx * C;
    float* po = out + (long long)z * out_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 o = zero4();
        if (in) { const float4 a = ld4(px + c), b = ld4(pt + c); o = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
        st4(po + c, o);
    }
}

// out[row] = valid ? dot(x[row], w) + b : 0       (modules.py:244-248 linear_layer + masked_fill)
__global__ void rowdot_kernel(const int* meta, int mfield, const float* x, long long x_ts, const float* w,
                              const float* b, long long par_ts, const unsigned char* valid, long long row_ts,
                              float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const float* px = x + (long long)z * x_ts + (long long)row * C;
    const float* pw = w + (long long)z * par_ts;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = ld4(px + c), ww = ld4(pw + c);
        s += (a.x * ww.x + a.y * ww.y) + (a.z * ww.z + a.w * ww.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[(long long)z * out_ts + row] = valid[(long long)z * row_ts + row] ? s + b[(long long)z * par_ts] : 0.f;
}

// dx[row] = dout[row] * w
__global__ void rowdot_bwd_kernel(const int* meta, int mfield, const float* dout, long long dout_ts, const float* w,
                                  long long par_ts, float* dx, long long dx_ts, int C) {
    ROW_PROLOGUE(mfield)
    const float d = dout[(long long)z * dout_ts + row];
    const float* pw = w + (long long)z * par_ts;
    float* pd = dx + (long long)z * dx_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 ww = ld4(pw + c);
        st4(pd + c, make_float4(d * ww.x, d * ww.y, d * ww.z, d * ww.w));
    }
}

// ------------------------------------------------------------------------------------------
// length regulator (modules.py:167-190): frame row r copies phoneme row src[r]; the speaker
// vector (fastspeech2.py:91-94) and the decoder's sinusoid row (Models.py:157-159) are added in
// the same pass.  Backward is a contiguous segment sum per phoneme (no atomics).
// ------------------------------------------------------------------------------------------
__global__ void length_regulate_fwd_kernel(const int* meta, const float* x, long long x_ts, const int* src,
                                           const int* row_b, const int* row_t, long long row_ts, const float* spk,
                                           long long spk_ts, const float* pos, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(META_MF)
    const long long r = (long long)z * row_ts + row;
    const int s = src[r];
    float* po = out + (long long)z * out_ts + (long long)row * C;
    if (s < 0) { for (int c = lane * 4; c < C; c += 256) st4(po + c, zero4()); return; }
    const float* px = x + (long long)z * x_ts + (long long)s * C;
    const float* pv = spk + (long long)z * spk_ts + (long long)row_b[r] * C;
    const float* pp = pos ? pos + (long long)row_t[r] * C : nullptr;  // null: tangent pass (positions carry no tangent)
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = ld4(px + c), b = ld4(pv + c), d = pp ? ld4(pp + c) : zero4();
        st4(po + c, make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w));
    }
}

// frame-level features: the length regulator's output on the zero-padded frame RECTANGLE (modules.py:128-137 + pad), before the
// frame-level pitch / energy predictors: out[rr] = x[f_src[r2f[rr]]] on valid frames, 0 elsewhere
__global__ void length_regulate_rect_kernel(const int* meta, const float* x, long long x_ts, const int* f_src, long long f_ts, const int* r2f,
                                            long long r_ts, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(META_MR)
    const int fr = r2f[(long long)z * r_ts + row];
    const int s = fr >= 0 ? f_src[(long long)z * f_ts + fr] : -1;
    float* po = out + (long long)z * out_ts + (long long)row * C;
    const float* px = s >= 0 ? x + (long long)z * x_ts + (long long)s * C : nullptr;
    for (int c = lane * 4; c < C; c += 256) st4(po + c, px ? ld4(px + c) : zero4());
}

// dx[p] (+)= sum_{r in [first[p], first[p]+count[p])} dout[r]   (rows of the phoneme space)
__global__ void length_regulate_bwd_kernel(const int* meta, const float* dout, long long dout_ts, const int* first,
                                           const int* count, long long row_ts, float* dx, long long dx_ts, int C,
                                           int accumulate) {
    ROW_PROLOGUE(META_MP)
    const long long r = (long long)z * row_ts + row;
    const int f = first[r], n = count[r];
    const float* pd = dout + (long long)z * dout_ts;
    float* po = dx + (long long)z * dx_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 s = accumulate ? ld4(po + c) : zero4();
        for (int i = 0; i < n; ++i) {
            const float4 a = ld4(pd + (long long)(f + i) * C + c);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        st4(po + c, s);
    }
}

// out[b][c] (+)= sum over rows [start[b], start[b]+len[b]) of X   (speaker-vector gradient)
__global__ void segsum_rows_kernel(const int* meta, const float* X, long long x_ts, const int* start, const int* len,
                                   long long seg_ts, float* out, long long out_ts, int C, int accumulate) {
    // 256 threads = 64 columns x 4 row lanes; each lane walks every 4th row with two independent partial sums, the
    // four lanes are folded in lane order through LDS (fixed order)
    __shared__ float red[4][64];
    const int z = blockIdx.z, b = blockIdx.y, B = meta[z * META_STRIDE + META_B];
    if (b >= B) return;
    const int cl = (int)threadIdx.x & 63, q = (int)threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cin = c < C;
    const int s0 = start[(long long)z * seg_ts + b], n = len[(long long)z * seg_ts + b];
    const float* px = X + (long long)z * x_ts + (long long)s0 * C + (cin ? c : 0);
    float a0 = 0.f, a1 = 0.f;
    if (cin) {
        int i = q;
        for (; i + 4 < n; i += 8) { a0 += px[(long long)i * C]; a1 += px[(long long)(i + 4) * C]; }
        if (i < n) a0 += px[(long long)i * C];
    }
    red[q][cl] = a0 + a1;
    __syncthreads();
    if (q != 0 || !cin) return;
    float s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (accumulate) s += out[(long long)z * out_ts + (long long)b * C + c];
    out[(long long)z * out_ts + (long long)b * C + c] = s;
}

// Deterministic embedding-table gradient: one workgroup per table row scans the index list in
// order.  dtable[v] = sum_{m : idx[m] == v} dx[m]; rows never referenced (and `skip_row`, the
// padding_idx of src_word_emb, Models.py:56-58) are written as zero, so no memset is needed.
__global__ void table_grad_kernel(const int* meta, int mfield, const float* dx, long long dx_ts, const int* idx,
                                  long long idx_ts, int skip_row, float* dtable, long long dt_ts, int C) {
    // 64 threads: the index list is matched 64 entries at a time (one compare per lane); the 64 match bits are gathered
    // with four exact 16-bit wavefront sums and the hits are then accumulated in ascending row order — the summation
    // order of a serial scan, with a loop per HIT instead of per entry
    // blockDim.x = 64 * W wavefronts: wavefront w takes every W-th block of 64 entries (the table row that collects the padded
    // positions has ~100 hits per task and was the kernel's tail with one wavefront); the W partial rows are folded in wavefront order
    const int z = blockIdx.z, v = blockIdx.x, lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, W = (int)blockDim.x >> 6;
    const int M_ = meta[z * META_STRIDE + mfield];
    const int* pi = idx + (long long)z * idx_ts;
    const float* pd = dx + (long long)z * dx_ts;
    float* po = dtable + (long long)z * dt_ts + (long long)v * C;
    float s[16];  // C <= 1024
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0.f;
    if (v != skip_row)
        for (int base = wv * 64; base < M_; base += 64 * W) {
            const int m = base + lane;
            const bool hit = m < M_ && pi[m] == v;
            unsigned long long bits = 0;
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const float w = (hit && (lane >> 4) == part) ? (float)(1u << (lane & 15)) : 0.f;
                bits |= (unsigned long long)(unsigned)wave_sum(w) << (16 * part);
            }
            // hits four at a time: the loads of four rows are in flight together, the adds keep the ascending row order
            // (a padding bucket collects ~100 rows per task; one dependent load per hit made that workgroup the kernel's tail)
            while (bits) {
                const float* row[4];
                int nh = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    row[q] = pd;
                    if (bits) { row[q] = pd + (long long)(base + __builtin_ctzll(bits)) * C; bits &= bits - 1; nh = q + 1; }
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int c = lane + 64 * k;
                    if (c < C) {
                        const float v0 = row[0][c], v1 = row[1][c], v2 = row[2][c], v3 = row[3][c];
                        s[k] += v0;
                        if (nh > 1) s[k] += v1;
                        if (nh > 2) s[k] += v2;
                        if (nh > 3) s[k] += v3;
                    }
                }
            }
        }
    if (W > 1) {
        __shared__ float tg_red[3][1024];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; if (c < C && wv > 0 && wv < 4) tg_red[wv - 1][c] = s[k]; }
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = lane + 64 * k;
            if (c < C) for (int w = 1; w < W && w < 4; ++w) s[k] += tg_red[w - 1][c];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int c = lane + 64 * k; if (c < C) po[c] = s[k]; }
}

// speaker table gradient from per-utterance vector grads (table path; averaged path divides by
// the number of support ids and gives every support id the summed gradient)
__global__ void speaker_table_grad_kernel(const int* meta, const float* dspk, long long dspk_ts, const int* ids,
                                          long long ids_ts, int n_ids_max, int average, float* dtable,
                                          long long dt_ts, int C) {
    const int z = blockIdx.z, v = blockIdx.x, B = meta[z * META_STRIDE + META_B];
    const int* id = ids + (long long)z * ids_ts;
    const float* pd = dspk + (long long)z * dspk_ts;
    float* po = dtable + (long long)z * dt_ts + (long long)v * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        if (average) {
            const int n = id[n_ids_max];
            int hits = 0;
            for (int j = 0; j < n; ++j) hits += (id[j] == v);
            if (hits) {
                float t = 0.f;
                for (int b = 0; b < B; ++b) t += pd[(long long)b * C + c];
                s = t * (float)hits / (float)n;
            }
        } else {
            for (int b = 0; b < B; ++b)
                if (id[b] == v) s += pd[(long long)b * C + c];
        }
        po[c] = s;
    }
}

// ------------------------------------------------------------------------------------------
// PostNet BatchNorm1d (+tanh) on the (B, T') rectangle incl. padded frames (Layers.py:129-137)
// ------------------------------------------------------------------------------------------
// running stats: momentum update applied task after task (deterministic order), one launch
__global__ void bn_running_update_kernel(const float* stats, long long st_ts, int tasks, float* running_mean,
                                         float* running_var, int C, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    for (int t = 0; t < tasks; ++t) {
        const float* s = stats + (long long)t * st_ts;
        rm = (1.f - momentum) * rm + momentum * s[c];
        rv = (1.f - momentum) * rv + momentum * s[2 * C + c];
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}

// ... every BatchNorm layer of the PostNet in one launch (blockIdx.y = layer)
constexpr int kBnRunMax = 8;
struct BnRunArgs { const float* stats[kBnRunMax]; long long st_ts[kBnRunMax]; float* rm[kBnRunMax]; float* rv[kBnRunMax]; int C[kBnRunMax]; };
__global__ void bn_running_update_multi_kernel(BnRunArgs a, int tasks, float momentum) {
    const int l = blockIdx.y, C = a.C[l], c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = a.rm[l][c], rv = a.rv[l][c];
    for (int t = 0; t < tasks; ++t) {
        const float* s = a.stats[l] + (long long)t * a.st_ts[l];
        rm = (1.f - momentum) * rm + momentum * s[c];
        rv = (1.f - momentum) * rv + momentum * s[2 * C + c];
    }
    a.rm[l][c] = rm;
    a.rv[l][c] = rv;
}

// stats for eval mode from the running buffers: [mean | rstd]
__global__ void bn_eval_stats_kernel(const float* running_mean, const float* running_var, float* stats_out,
                                     long long st_ts, int tasks, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int t = 0; t < tasks; ++t) {
        float* so = stats_out + (long long)t * st_ts;
        so[c] = running_mean[c];
        so[C + c] = rsqrtf(running_var[c] + eps);
        so[2 * C + c] = running_var[c];
    }
}

// y = dropout(act((x - mean) * rstd * gamma + beta)) on in-rect rows, 0 on guard rows
// dout: the F.dropout behind every PostNet layer (Layers.py:133-134) applied in passing (it used to be a launch of its own)
__global__ void bn_apply_kernel(const int* meta, const float* X, long long x_ts, const float* stats, long long st_ts,
                                const float* gamma, const float* beta, long long par_ts, const unsigned char* inrect,
                                long long row_ts, int do_tanh, float* Y, long long y_ts, int C, DropSpec dout, bf16_t* Yh) {
    ROW_PROLOGUE(META_MR)
    const bool in = inrect[(long long)z * row_ts + row] != 0;
    const float* px = X + (long long)z * x_ts + (long long)row * C;
    float* py = Y + (long long)z * y_ts + (long long)row * C;
    const float* st = stats + (long long)z * st_ts;
    const float* g = gamma + (long long)z * par_ts;
    const float* b = beta + (long long)z * par_ts;
    for (int c = lane * 4; c < C; c += 256) {
        float4 o = zero4();
        if (in) {
            const float4 x = ld4(px + c), mu = ld4(st + c), rs = ld4(st + C + c), g4 = ld4(g + c), b4 = ld4(b + c);
            o = make_float4((x.x - mu.x) * rs.x * g4.x + b4.x, (x.y - mu.y) * rs.y * g4.y + b4.y,
                            (x.z - mu.z) * rs.z * g4.z + b4.z, (x.w - mu.w) * rs.w * g4.w + b4.w);
            if (do_tanh) o = make_float4(tanhf(o.x), tanhf(o.y), tanhf(o.z), tanhf(o.w));
            if (dout.thr16) o = drop4(dout, z, row, C, c, o);
        }
        st4(py + c, o);
        if (Yh) st4_bf16(Yh + (long long)z * y_ts + (long long)row * C + c, o);
    }
}

// backward pass 2: dx = gamma * rstd * (dpre - dbeta/n - xhat * dgamma/n) on in-rect rows
__global__ void bn_bwd_apply_kernel(const int* meta, const float* dY, long long dy_ts, const float* Yact,
                                    long long ya_ts, const float* X, long long x_ts, const float* stats,
                                    long long st_ts, const float* gamma, long long par_ts, const float* dgamma,
                                    const float* dbeta, long long dg_ts, const unsigned char* inrect,
                                    long long row_ts, int do_tanh, float* dX, long long dx_ts, int C, float yscale, DropSpec din, bf16_t* dXh) {
    // din: the backward of the dropout behind the layer, applied to dY on load (same mask as the forward's); dXh: bf16 twin of dX
    ROW_PROLOGUE(META_MR)
    float* pdx = dX + (long long)z * dx_ts + (long long)row * C;
    bf16_t* pdh = dXh ? dXh + (long long)z * dx_ts + (long long)row * C : nullptr;
    if (!inrect[(long long)z * row_ts + row]) {
        for (int c = lane * 4; c < C; c += 256) { st4(pdx + c, zero4()); if (pdh) st4_bf16(pdh + c, zero4()); }
        return;
    }
    const float inv_n = 1.f / (float)(meta[z * META_STRIDE + META_B] * meta[z * META_STRIDE + META_TCAP]);
    const float* pdy = dY + (long long)z * dy_ts + (long long)row * C;
    const float* pya = Yact + (long long)z * ya_ts + (long long)row * C;
    const float* px = X + (long long)z * x_ts + (long long)row * C;
    const float* st = stats + (long long)z * st_ts;
    const float* g = gamma + (long long)z * par_ts;
    const float* dg = dgamma + (long long)z * dg_ts;
    const float* db = dbeta + (long long)z * dg_ts;
    for (int c = lane * 4; c < C; c += 256) {
        float4 d4 = ld4(pdy + c);
        if (din.thr16) d4 = drop4(din, z, row, C, c, d4);
        const float4 x4 = ld4(px + c), mu = ld4(st + c), rs = ld4(st + C + c), g4 = ld4(g + c),
                     dg4 = ld4(dg + c), db4 = ld4(db + c);
        float dd[4] = {d4.x, d4.y, d4.z, d4.w};
        if (do_tanh) {
            const float4 y4 = ld4(pya + c);
            const float y0 = y4.x * yscale, y1 = y4.y * yscale, y2 = y4.z * yscale, y3 = y4.w * yscale;
            dd[0] *= (1.f - y0 * y0); dd[1] *= (1.f - y1 * y1);
            dd[2] *= (1.f - y2 * y2); dd[3] *= (1.f - y3 * y3);
        }
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ms[4] = {mu.x, mu.y, mu.z, mu.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
        const float gs[4] = {g4.x, g4.y, g4.z, g4.w}, dgs[4] = {dg4.x, dg4.y, dg4.z, dg4.w}, dbs[4] = {db4.x, db4.y, db4.z, db4.w};
        float o[4];
        for (int i = 0; i < 4; ++i) {
            const float xh = (xs[i] - ms[i]) * rr[i];
            o[i] = gs[i] * rr[i] * (dd[i] - dbs[i] * inv_n - xh * dgs[i] * inv_n);
        }
        st4(pdx + c, make_float4(o[0], o[1], o[2], o[3]));
        if (pdh) st4_bf16(pdh + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// mel rectangle: padded in-rect rows carry mel_linear.bias (decoder output is zero there);
// guard rows are re-zeroed (the previous batch plan may have had frames there)
__global__ void fill_padded_rows_kernel(const int* meta, float* X, long long x_ts, const float* bias, long long par_ts,
                                        const unsigned char* inrect, const unsigned char* valid, long long row_ts, int C) {
    ROW_PROLOGUE(META_MR)
    const long long r = (long long)z * row_ts + row;
    if (valid[r]) return;
    const bool in = inrect[r] != 0;
    float* px = X + (long long)z * x_ts + (long long)row * C;
    const float* b = bias + (long long)z * par_ts;
    for (int c = lane * 4; c < C; c += 256) st4(px + c, in ? ld4(b + c) : zero4());
}

// out[r] = rowmap[r] >= 0 ? src[rowmap[r]] : 0   (row gather between row spaces)
__global__ void gather_rows_kernel(const int* meta, int mfield, const float* src, long long src_ts, const int* rowmap,
                                   long long map_ts, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const int s = rowmap[(long long)z * map_ts + row];
    float* po = out + (long long)z * out_ts + (long long)row * C;
    const float* ps = src + (long long)z * src_ts + (long long)(s < 0 ? 0 : s) * C;
    for (int c = lane * 4; c < C; c += 256) st4(po + c, s < 0 ? zero4() : ld4(ps + c));
}

// ------------------------------------------------------------------------------------------
// Dropout (nn.Dropout / F.dropout of SubLayers.py:54,90, modules.py:223,235, Layers.py:133-134) as a
// counter-based mask: keep(seed, task, row, col) is a pure function (splitmix64 of the element id), so
// forward, backward and the second-order replay regenerate the same mask without storing it.
// dst = keep ? src / (1 - p) : 0 on rows < M; dst may alias src.  thr16 = round(p * 65536).
// ------------------------------------------------------------------------------------------
__global__ void dropout_kernel(const int* meta, int mfield, const float* src, long long src_ts, float* dst, long long dst_ts,
                               int C, unsigned seed, unsigned thr16, float scale) {
    ROW_PROLOGUE(mfield)
    const float* ps = src + (long long)z * src_ts + (long long)row * C;
    float* pd = dst + (long long)z * dst_ts + (long long)row * C;
    const unsigned long long base = ((unsigned long long)seed << 32) ^ ((unsigned long long)z << 24);
    for (int c = lane * 4; c < C; c += 256) {
        const unsigned long long h = splitmix64(base + ((unsigned long long)row * (unsigned)C + (unsigned)c) / 4ull);
        const float4 v = ld4(ps + c);
        st4(pd + c, make_float4(((h) & 0xFFFFu) >= thr16 ? v.x * scale : 0.f, ((h >> 16) & 0xFFFFu) >= thr16 ? v.y * scale : 0.f,
                                ((h >> 32) & 0xFFFFu) >= thr16 ? v.z * scale : 0.f, ((h >> 48) & 0xFFFFu) >= thr16 ? v.w * scale : 0.f));
    }
}

// ------------------------------------------------------------------------------------------
// elementwise helpers over [tasks][n] float arrays (n % 4 == 0)
// ------------------------------------------------------------------------------------------
__global__ void add2_kernel(const float* a, const float* b, float* out, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ld4(a + i * 4), y = ld4(b + i * 4);
        st4(out + i * 4, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
    }
}

// out = (a + b) + c, in that order (the order in which three accumulating GEMM epilogues would have added them)
__global__ void add3_kernel(const float* a, const float* b, const float* c, float* out, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ld4(a + i * 4), y = ld4(b + i * 4), z = ld4(c + i * 4);
        st4(out + i * 4, make_float4((x.x + y.x) + z.x, (x.y + y.y) + z.y, (x.z + y.z) + z.z, (x.w + y.w) + z.w));
    }
}

// ------------------------------------------------------------------------------------------
// bf16 operand planes (bf16 numerics mode, gemm_bf16.h: GemmArgs::Ah / Bh)
// ------------------------------------------------------------------------------------------
// dst[i] = bf16(src[i]) over a flat range (n8 groups of 8): the plane of an activation slab whose producer does not write one in passing
__global__ void to_bf16_kernel(const float* src, bf16_t* dst, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = ld4(src + i * 8), b = ld4(src + i * 8 + 4);
        struct alignas(16) { bf16_t v[8]; } o;
        o.v[0] = f32_to_bf16(a.x); o.v[1] = f32_to_bf16(a.y); o.v[2] = f32_to_bf16(a.z); o.v[3] = f32_to_bf16(a.w);
        o.v[4] = f32_to_bf16(b.x); o.v[5] = f32_to_bf16(b.y); o.v[6] = f32_to_bf16(b.z); o.v[7] = f32_to_bf16(b.w);
        *reinterpret_cast<decltype(o)*>(dst + i * 8) = o;
    }
}

// The bf16 shadows of the Conv1d / Linear weights, refreshed once per pass from the fp32 masters: `fwd` keeps the [Cout][k][Cin] layout
// (the forward conv's B operand), `tr` is the image the input-gradient conv reads as ITS K-contiguous B operand —
// tr[ci][(k - 1 - t) * Cout + co] = w[co][t * Cin + ci] (dX = conv(dY, flipped transposed W): an NT problem like the forward).
// One workgroup = one 32 x 32 (co, ci) tile of one tap, transposed through LDS; blockIdx.z = task (per-task fast weights).
struct ShadowEnt { long long off, dst; int cout, k, cin, tile0; };   // off / dst: element offset of the weight in the source vector / of its shadow in the shadow vectors (a multiple of 8); tile0: first tile
__global__ void weight_shadow_kernel(const ShadowEnt* ents, int n_ents, const float* src, long long src_ts, bf16_t* fwd, bf16_t* tr, long long dst_ts) {
    __shared__ float tile[32][33];
    int e = 0;
    while (e + 1 < n_ents && (int)blockIdx.x >= ents[e + 1].tile0) ++e;
    const ShadowEnt en = ents[e];
    const int tci = (en.cin + 31) / 32, tco = (en.cout + 31) / 32;
    int t = (int)blockIdx.x - en.tile0;
    const int tap = t / (tci * tco); t -= tap * tci * tco;
    const int co0 = (t / tci) * 32, ci0 = (t % tci) * 32;
    const float* w = src + (long long)blockIdx.z * src_ts + en.off;
    bf16_t* f = fwd + (long long)blockIdx.z * dst_ts + en.dst;
    bf16_t* r = tr + (long long)blockIdx.z * dst_ts + en.dst;
    const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;   // 256 threads: 32 x 8
    const long long ldw = (long long)en.k * en.cin, ldt = (long long)en.k * en.cout;
    for (int j = ty; j < 32; j += 8) {
        const int co = co0 + j, ci = ci0 + tx;
        float v = 0.f;
        if (co < en.cout && ci < en.cin) {
            v = w[co * ldw + (long long)tap * en.cin + ci];
            f[co * ldw + (long long)tap * en.cin + ci] = f32_to_bf16(v);
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int ci = ci0 + j, co = co0 + tx;
        if (co < en.cout && ci < en.cin) r[ci * ldt + (long long)(en.k - 1 - tap) * en.cout + co] = f32_to_bf16(tile[tx][j]);
    }
}

// dst[t][i] = src[i]  (clone the adapted parameters into every task's fast weights)
__global__ void broadcast_kernel(const float* src, float* dst, long long n4, long long dst_ts) {
    float* d = dst + (long long)blockIdx.z * dst_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        st4(d + i * 4, ld4(src + i * 4));
}

// MAML inner update  theta' = theta - lr * g   (learn2learn maml_update via systems/utils.py:39-47)
__global__ void sgd_update_kernel(float* w, const float* g, long long n4, float lr, long long w_ts, long long g_ts) {
    float* pw = w + (long long)blockIdx.z * w_ts;
    const float* pg = g + (long long)blockIdx.z * g_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 x = ld4(pw + i * 4);
        const float4 d = ld4(pg + i * 4);
        x.x -= lr * d.x; x.y -= lr * d.y; x.z -= lr * d.z; x.w -= lr * d.w;
        st4(pw + i * 4, x);
    }
}

// out[i] (+)= scale * sum_t g[t][i]   (fixed task order; accumulate: gradient accumulation over several batches, main.py:62)
__global__ void sum_tasks_kernel(const float* g, long long g_ts, int tasks, float scale, float* out, long long n4, int accumulate) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 s = zero4();
        for (int t = 0; t < tasks; ++t) {
            const float4 x = ld4(g + (long long)t * g_ts + i * 4);
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        float4 o = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
        if (accumulate) { const float4 p = ld4(out + i * 4); o = make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w); }
        st4(out + i * 4, o);
    }
}

// The exchange step's tail (engine.h: sync_pack / sync_unpack): what DDP moves between ranks besides the gradient — the 6 loss scalars
// of `self.log_dict(..., sync_dist=True)` (meta.py:78-79, baseline.py:35) and the PostNet BatchNorm running buffers (DDP
// `broadcast_buffers`, main.py:32) — rides behind the flat outer gradient in ONE all-reduce.
struct SyncBn { float* rm[8]; float* rv[8]; int c[8]; int off[8]; int n; };
// blockIdx.x == 0: tail[k] = scale * sum_t losses[t][k]; blocks 1 .. n: BatchNorm layer (blockIdx.x - 1): tail_bn = w * (running mean | var)
__global__ void sync_pack_kernel(const float* losses, int tasks, float scale, float* tail, SyncBn bn, float w) {
    const int b = blockIdx.x;
    if (b == 0) {
        if (threadIdx.x < 8) {
            float s = 0.f;
            if (threadIdx.x < 6) for (int t = 0; t < tasks; ++t) s += losses[t * 6 + threadIdx.x];
            tail[threadIdx.x] = s * scale;
        }
        return;
    }
    const int l = b - 1;
    float* dst = tail + 8 + bn.off[l];
    for (int j = threadIdx.x; j < bn.c[l]; j += blockDim.x) { dst[j] = w * bn.rm[l][j]; dst[bn.c[l] + j] = w * bn.rv[l][j]; }
}
__global__ void sync_unpack_kernel(const float* tail, SyncBn bn) {
    const int l = blockIdx.x;
    const float* src = tail + 8 + bn.off[l];
    for (int j = threadIdx.x; j < bn.c[l]; j += blockDim.x) { bn.rm[l][j] = src[j]; bn.rv[l][j] = src[bn.c[l] + j]; }
}

// global L2 norm, stage 1: per-block partial sums of squares (double accumulation in the
// second stage keeps it deterministic and accurate)
__global__ void sumsq_partial_kernel(const float* g, long long n4, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ld4(g + i * 4);
        s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    s = wave_sum(s);
    if (((int)threadIdx.x & 63) == 0) red[(int)threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// extra (optional): a device scalar holding the squared gradient norm of parameters that live outside this buffer (a co-trained
// speaker encoder) — clip_grad_norm_ takes the norm over ALL parameters of the model (main.py:61)
__global__ void sumsq_final_kernel(const float* partial, int n, float* out_norm, const float* extra) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = extra ? (double)extra[0] : 0.0;
        for (int i = 0; i < n; ++i) s += (double)partial[i];
        out_norm[0] = (float)sqrt(s);
    }
}

// fused clip_grad_norm_(max_norm) + Adam (optimizer.py:9-15, main.py:61); norm read on device
__global__ void adam_clip_kernel(float* w, const float* g, float* m, float* v, long long n4, const float* norm,
                                 float max_norm, float lr, float b1, float b2, float eps, float bc1, float bc2,
                                 float weight_decay) {
    float coef = 1.f;
    if (max_norm > 0.f) { coef = max_norm / (norm[0] + 1e-6f); coef = coef > 1.f ? 1.f : coef; }
    const float step = lr / bc1, isq = 1.f / sqrtf(bc2);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 W = ld4(w + i * 4), G = ld4(g + i * 4), Mo = ld4(m + i * 4), Vo = ld4(v + i * 4);
        float ws[4] = {W.x, W.y, W.z, W.w}, gs[4] = {G.x, G.y, G.z, G.w}, ms[4] = {Mo.x, Mo.y, Mo.z, Mo.w}, vs[4] = {Vo.x, Vo.y, Vo.z, Vo.w};
        for (int k = 0; k < 4; ++k) {
            float gg = gs[k] * coef;
            if (weight_decay != 0.f) gg += weight_decay * ws[k];
            ms[k] = b1 * ms[k] + (1.f - b1) * gg;
            vs[k] = b2 * vs[k] + (1.f - b2) * gg * gg;
            const float denom = sqrtf(vs[k]) * isq + eps;
            ws[k] -= step * ms[k] / denom;
        }
        st4(w + i * 4, make_float4(ws[0], ws[1], ws[2], ws[3]));
        st4(m + i * 4, make_float4(ms[0], ms[1], ms[2], ms[3]));
        st4(v + i * 4, make_float4(vs[0], vs[1], vs[2], vs[3]));
    }
}

// ------------------------------------------------------------------------------------------
// loss (loss.py:19-92) on the mel rectangle and the phoneme rectangle
// ------------------------------------------------------------------------------------------
struct LossArgs {
    const float *mel, *mel_post, *mel_tgt;       // [Mr][n_mel]
    const unsigned char* rvalid;                 // mel rows
    const float *pp, *ep, *logd, *p_tgt, *e_tgt; // [Mp]
    const int* dur;                              // [Mp] duration targets
    const unsigned char* pvalid;                 // phoneme rows
    long long mel_ts, rrow_ts, prow_ts, pred_ts; // strides
    int n_mel;
    // frame-level pitch / energy (preprocess `feature: frame_level`; loss.py:54-63): predictions and targets live on the mel rows
    int pitch_frame = 0, energy_frame = 0;
    const float *pp_r = nullptr, *ep_r = nullptr;      // [Mr], stride pred_r_ts
    const float *p_tgt_r = nullptr, *e_tgt_r = nullptr; // [Mr], stride rrow_ts
    long long pred_r_ts = 0;
};

constexpr int kLossBlocks = 64;

// partial[task][block][5]: sum|mel - t|, sum|post - t|, sum (pp-pt)^2, sum (ep-et)^2, sum (logd - log(d+1))^2
__global__ void loss_partial_kernel(const int* meta, LossArgs a, float* partial) {
    __shared__ float red[4][5];
    const int z = blockIdx.z;
    const int Mr = meta[z * META_STRIDE + META_MR], Mp = meta[z * META_STRIDE + META_MP];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int n4 = a.n_mel / 4;
    const long long tot = (long long)Mr * n4;
    const float* mel = a.mel + (long long)z * a.mel_ts;
    const float* post = a.mel_post + (long long)z * a.mel_ts;
    const float* tgt = a.mel_tgt + (long long)z * a.mel_ts;
    const unsigned char* rv = a.rvalid + (long long)z * a.rrow_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / n4);
        if (!rv[r]) continue;
        const float4 m = ld4(mel + i * 4), p = ld4(post + i * 4), t = ld4(tgt + i * 4);
        s[0] += (fabsf(m.x - t.x) + fabsf(m.y - t.y)) + (fabsf(m.z - t.z) + fabsf(m.w - t.w));
        s[1] += (fabsf(p.x - t.x) + fabsf(p.y - t.y)) + (fabsf(p.z - t.z) + fabsf(p.w - t.w));
    }
    const unsigned char* pv = a.pvalid + (long long)z * a.prow_ts;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mp; r += gridDim.x * blockDim.x) {
        if (!pv[r]) continue;
        const long long q = (long long)z * a.pred_ts + r, qt = (long long)z * a.prow_ts + r;
        const float dp = a.pitch_frame ? 0.f : a.pp[q] - a.p_tgt[qt], de = a.energy_frame ? 0.f : a.ep[q] - a.e_tgt[qt];
        const float dd = a.logd[q] - logf((float)a.dur[qt] + 1.f);
        s[2] += dp * dp; s[3] += de * de; s[4] += dd * dd;
    }
    if (a.pitch_frame || a.energy_frame)
        for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mr; r += gridDim.x * blockDim.x) {
            if (!rv[r]) continue;
            const long long q = (long long)z * a.pred_r_ts + r, qt = (long long)z * a.rrow_ts + r;
            if (a.pitch_frame) { const float dp = a.pp_r[q] - a.p_tgt_r[qt]; s[2] += dp * dp; }
            if (a.energy_frame) { const float de = a.ep_r[q] - a.e_tgt_r[qt]; s[3] += de * de; }
        }
    for (int k = 0; k < 5; ++k) s[k] = wave_sum(s[k]);
    if (((int)threadIdx.x & 63) == 0) for (int k = 0; k < 5; ++k) red[(int)threadIdx.x >> 6][k] = s[k];
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        partial[((long long)z * gridDim.x + blockIdx.x) * 5 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// losses[task][6] = (total, mel, postnet mel, pitch, energy, duration)
__global__ void loss_final_kernel(const int* meta, const float* partial, int nblocks, int n_mel, float* losses, int pitch_frame = 0,
                                  int energy_frame = 0) {
    const int z = blockIdx.x;
    if (threadIdx.x != 0) return;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b) for (int k = 0; k < 5; ++k) s[k] += (double)partial[((long long)z * nblocks + b) * 5 + k];
    const double nF = (double)meta[z * META_STRIDE + META_NF] * n_mel, nP = (double)meta[z * META_STRIDE + META_NP];
    const double nFr = (double)meta[z * META_STRIDE + META_NF];   // valid frames: the element count of a frame-level MSE
    const float mel = (float)(s[0] / nF), post = (float)(s[1] / nF), p = (float)(s[2] / (pitch_frame ? nFr : nP)),
                e = (float)(s[3] / (energy_frame ? nFr : nP)), d = (float)(s[4] / nP);
    float* o = losses + (long long)z * 6;
    o[0] = mel + post + d + p + e; o[1] = mel; o[2] = post; o[3] = p; o[4] = e; o[5] = d;
}

__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// d(total)/d(predictions) * scale.  dmel, dpost: [Mr][n_mel] (0 on padded / guard rows);
// dpp, dep, dlogd: [Mp]
__global__ void loss_grad_kernel(const int* meta, LossArgs a, float scale, float* dmel, float* dpost, float* dpp,
                                 float* dep, float* dlogd, float* dpp_r = nullptr, float* dep_r = nullptr) {
    const int z = blockIdx.z;
    const int Mr = meta[z * META_STRIDE + META_MR], Mp = meta[z * META_STRIDE + META_MP];
    const float wF = scale / ((float)meta[z * META_STRIDE + META_NF] * (float)a.n_mel);
    const float wP = 2.f * scale / (float)meta[z * META_STRIDE + META_NP];
    const int n4 = a.n_mel / 4;
    const long long tot = (long long)Mr * n4;
    const float* mel = a.mel + (long long)z * a.mel_ts;
    const float* post = a.mel_post + (long long)z * a.mel_ts;
    const float* tgt = a.mel_tgt + (long long)z * a.mel_ts;
    float* dm = dmel + (long long)z * a.mel_ts;
    float* dq = dpost + (long long)z * a.mel_ts;
    const unsigned char* rv = a.rvalid + (long long)z * a.rrow_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / n4);
        float4 o1 = zero4(), o2 = zero4();
        if (rv[r]) {
            const float4 m = ld4(mel + i * 4), p = ld4(post + i * 4), t = ld4(tgt + i * 4);
            o1 = make_float4(wF * sgnf(m.x - t.x), wF * sgnf(m.y - t.y), wF * sgnf(m.z - t.z), wF * sgnf(m.w - t.w));
            o2 = make_float4(wF * sgnf(p.x - t.x), wF * sgnf(p.y - t.y), wF * sgnf(p.z - t.z), wF * sgnf(p.w - t.w));
        }
        st4(dm + i * 4, o1);
        st4(dq + i * 4, o2);
    }
    const unsigned char* pv = a.pvalid + (long long)z * a.prow_ts;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mp; r += gridDim.x * blockDim.x) {
        const long long q = (long long)z * a.pred_ts + r, qt = (long long)z * a.prow_ts + r;
        float gp = 0.f, ge = 0.f, gd = 0.f;
        if (pv[r]) {
            if (!a.pitch_frame) gp = wP * (a.pp[q] - a.p_tgt[qt]);
            if (!a.energy_frame) ge = wP * (a.ep[q] - a.e_tgt[qt]);
            gd = wP * (a.logd[q] - logf((float)a.dur[qt] + 1.f));
        }
        dpp[q] = gp; dep[q] = ge; dlogd[q] = gd;
    }
    if (a.pitch_frame || a.energy_frame) {
        const float wR = 2.f * scale / (float)meta[z * META_STRIDE + META_NF];
        for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mr; r += gridDim.x * blockDim.x) {
            const long long q = (long long)z * a.pred_r_ts + r, qt = (long long)z * a.rrow_ts + r;
            if (a.pitch_frame) dpp_r[q] = rv[r] ? wR * (a.pp_r[q] - a.p_tgt_r[qt]) : 0.f;
            if (a.energy_frame) dep_r[q] = rv[r] ? wR * (a.ep_r[q] - a.e_tgt_r[qt]) : 0.f;
        }
    }
}

// free-running durations (modules.py:132-136): clamp(round(exp(logd) - 1) * d_control, min 0)
__global__ void duration_round_kernel(const int* meta, const float* logd, long long pred_ts, float d_control,
                                      const unsigned char* pvalid, long long prow_ts, float* d_rounded) {
    const int z = blockIdx.z, Mp = meta[z * META_STRIDE + META_MP];
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mp; r += gridDim.x * blockDim.x) {
        const long long q = (long long)z * pred_ts + r;
        float d = rintf(expf(logd[q]) - 1.f) * d_control;
        d_rounded[q] = (d > 0.f) ? d : 0.f;
        (void)pvalid; (void)prow_ts;
    }
}

}  // namespace mtts
