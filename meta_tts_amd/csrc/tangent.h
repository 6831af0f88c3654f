// Tangent (forward-over-reverse) companions of the row kernels: what second-order MAML needs beyond
// GEMMs.  The second-order meta-gradient (reference: learn2learn `create_graph=True` double backward,
// lightning/systems/base_adaptor.py:107 `first_order = not train`) is obtained here WITHOUT autograd as a
// backward recursion over the stored inner steps, each step needing one Hessian-vector product
//     (H v) = d/de  grad L_support(phi + e v, theta_enc) |_{e=0}
// = the tangent of the whole forward+backward pass in direction v.  Linear ops (Linear / Conv1d /
// attention products, gathers, segment sums) reuse the grouped GEMM and row kernels on tangent data;
// the non-linear ops get the closed-form tangents below.  Conventions as in rowops.h; "t" prefixes the
// tangent of a forward value, "tg" the tangent of a gradient.  ReLU and L1 are piecewise linear: their
// tangents are the same masks / zero.
#pragma once
#include "rowops.h"

namespace mtts {

__device__ __forceinline__ float4 f4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }

// ---- LayerNorm --------------------------------------------------------------------------------
// forward tangent: tz = ta (+ tres); m1 = mean(tz), m2 = mean(xhat tz); t_xhat = r (tz - m1 - xhat m2);
// ty = mask ? tgamma*xhat + gamma*t_xhat + tbeta : 0.   tstats = (m1, m2) per row.
template <int NV>   // float4 per lane and row, see layernorm_fwd_kernel
__global__ void ln_jvp_fwd_kernel(const int* meta, int mfield, const float* ta, long long ta_ts, const float* tres,
                                  long long tres_ts, const float* zin, long long z_ts, const float* stats, long long st_ts,
                                  const float* gamma, long long par_ts, const float* tgamma, const float* tbeta,
                                  long long tpar_ts, const unsigned char* mask, long long mask_ts, float* tz_out,
                                  long long tz_ts, float* ty, long long ty_ts, float* tstats, long long tst_ts, int C, DropSpec din, DropSpec dout) {
    // dout: dropout applied to ty before the store (the variance predictors' F.dropout sits BEHIND the LayerNorm, modules.py:222-235)
    // din: dropout applied to ta on load (the tangent of `dropout(sublayer(x)) + residual`, SubLayers.py:54-55,90-91: the mask of the forward site —
    // it used to be a dropout launch of its own in front of this kernel)
    ROW_PROLOGUE(mfield)
    const float* pa = ta + (long long)z * ta_ts + (long long)row * C;
    const float* pr = tres ? tres + (long long)z * tres_ts + (long long)row * C : nullptr;
    const float* pz = zin + (long long)z * z_ts + (long long)row * C;
    const float* st = stats + (long long)z * st_ts + (long long)row * 2;
    const float mean = st[0], rstd = st[1];
    float4 tv[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int c = lane * 4 + 256 * n;
        tv[n] = zero4(); xh[n] = zero4();
        if (c >= C) continue;
        float4 t = ld4(pa + c);
        if (din.thr16) t = drop4(din, z, row, C, c, t);
        if (pr) { const float4 r4 = ld4(pr + c); t = f4(t.x + r4.x, t.y + r4.y, t.z + r4.z, t.w + r4.w); }
        const float4 x = ld4(pz + c);
        tv[n] = t;
        xh[n] = f4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
        s1 += (t.x + t.y) + (t.z + t.w);
        s2 += (t.x * xh[n].x + t.y * xh[n].y) + (t.z * xh[n].z + t.w * xh[n].w);
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    const bool keep = mask ? (mask[(long long)z * mask_ts + row] != 0) : true;
    const float* g = gamma + (long long)z * par_ts;
    const float* tg = tgamma ? tgamma + (long long)z * tpar_ts : nullptr;
    const float* tb = tbeta ? tbeta + (long long)z * tpar_ts : nullptr;
    float* po = ty + (long long)z * ty_ts + (long long)row * C;
    float* ptz = tz_out ? tz_out + (long long)z * tz_ts + (long long)row * C : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c >= C) continue;
        if (ptz) st4(ptz + c, tv[i]);
        float4 o = zero4();
        if (keep) {
            const float4 g4 = ld4(g + c);
            const float4 tg4 = tg ? ld4(tg + c) : zero4(), tb4 = tb ? ld4(tb + c) : zero4();
            const float tv_[4] = {tv[i].x, tv[i].y, tv[i].z, tv[i].w}, xh_[4] = {xh[i].x, xh[i].y, xh[i].z, xh[i].w};
            const float g_[4] = {g4.x, g4.y, g4.z, g4.w}, tg_[4] = {tg4.x, tg4.y, tg4.z, tg4.w}, tb_[4] = {tb4.x, tb4.y, tb4.z, tb4.w};
            float r_[4];
            for (int k = 0; k < 4; ++k) r_[k] = tg_[k] * xh_[k] + g_[k] * rstd * (tv_[k] - m1 - xh_[k] * m2) + tb_[k];
            o = f4(r_[0], r_[1], r_[2], r_[3]);
        }
        if (dout.thr16) o = drop4(dout, z, row, C, c, o);
        st4(po + c, o);
    }
    if (lane == 0) {
        float* ts = tstats + (long long)z * tst_ts + (long long)row * 2;
        ts[0] = m1;
        ts[1] = m2;
    }
}

// backward, primal + tangent in one pass:
//   g = dy*gamma, a1 = mean(g), a2 = mean(g xhat):          dz    = r (g - a1 - xhat a2)
//   tg = tg_y*gamma + dy*tgamma, t_xhat = r (tz - m1 - xhat m2), rdot/r = -r m2:
//   tg_z = -r m2 dz + r (tg - mean(tg) - t_xhat a2 - xhat (mean(tg xhat) + mean(g t_xhat)))
// masked rows give 0 for both; relu_on_z multiplies both by [z > 0].
template <int NV>
__global__ __launch_bounds__(256) void ln_jvp_bwd_kernel(const int* meta, int mfield, const float* dy, long long dy_ts, const float* tgy,
                                  long long tgy_ts, const float* zin, long long z_ts, const float* stats, long long st_ts,
                                  const float* tz, long long tz_ts, const float* tstats, long long tst_ts, const float* gamma,
                                  long long par_ts, const float* tgamma, long long tpar_ts, const unsigned char* mask,
                                  long long mask_ts, float* dz, long long dz_ts, float* tgz, long long tgz_ts, int C,
                                  int relu_on_z, float* dz_drop, long long dzd_ts, float* tgz_drop, long long tgd_ts, DropSpec dd, DropSpec din,
                                  float* partial, int max_chunks) {
    // din: dropout applied to dy and tgy on load (a dropout that sits BEHIND this LayerNorm in the forward: the variance predictors)
    // dz_drop / tgz_drop (optional, both or neither): dropout(dz) / dropout(tgz) with the mask of the forward site — the gradients entering the
    // dropped branch, while dz / tgz continue along the residual path (two dropout launches behind this kernel until round 6)
    // partial (optional): stage 1 of hv(gamma) = sum_rows (tgy xhat + dy t_xhat) and hv(beta) = sum_rows tgy over the unmasked rows rides here —
    // [task][chunk of kLnRows rows][3][C] (rows 0 / 1 used), folded by colfinal_kernel like the primal kernel's (layernorm_bwd_kernel) — instead of a
    // reduction launch (ColArgs mode 5) that read the same five arrays again.  A workgroup = kLnRows rows, two per wavefront (launch: row2_grid).
    __shared__ __attribute__((aligned(16))) float red[4][2][256 * NV];
    const int z = blockIdx.z;
    const int M_ = meta[z * META_STRIDE + mfield];
    if ((int)blockIdx.x * kLnRows >= M_) return;                       // (whole workgroup)
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    float4 pg[NV], pb[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) { pg[n] = zero4(); pb[n] = zero4(); }
    for (int q = 0; q < 2; ++q) {
        const int row = (int)blockIdx.x * kLnRows + wave * 2 + q;
        if (row >= M_) continue;                                         // (wave-uniform)
        float* pdz = dz + (long long)z * dz_ts + (long long)row * C;
        float* ptg = tgz + (long long)z * tgz_ts + (long long)row * C;
        float* pdd = dz_drop ? dz_drop + (long long)z * dzd_ts + (long long)row * C : nullptr;
        float* ptd = dz_drop ? tgz_drop + (long long)z * tgd_ts + (long long)row * C : nullptr;
        const bool keep = mask ? (mask[(long long)z * mask_ts + row] != 0) : true;
        if (!keep) {
            for (int c = lane * 4; c < C; c += 256) {
                st4(pdz + c, zero4()); st4(ptg + c, zero4());
                if (pdd) { st4(pdd + c, zero4()); st4(ptd + c, zero4()); }
            }
            continue;
        }
        const float* pdy = dy + (long long)z * dy_ts + (long long)row * C;
        const float* ptgy = tgy + (long long)z * tgy_ts + (long long)row * C;
        const float* pz = zin + (long long)z * z_ts + (long long)row * C;
        const float* ptz = tz + (long long)z * tz_ts + (long long)row * C;
        const float* st = stats + (long long)z * st_ts + (long long)row * 2;
        const float* ts = tstats + (long long)z * tst_ts + (long long)row * 2;
        const float mean = st[0], rstd = st[1], m1 = ts[0], m2 = ts[1];
        const float* g = gamma + (long long)z * par_ts;
        const float* tgm = tgamma ? tgamma + (long long)z * tpar_ts : nullptr;
        float gv[NV][4], tgv[NV][4], xh[NV][4], txh[NV][4], zz[NV][4];
        float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int c = lane * 4 + 256 * n;
            if (c >= C) continue;
            float4 d4 = ld4(pdy + c), t4 = ld4(ptgy + c);
            const float4 x4 = ld4(pz + c), tz4 = ld4(ptz + c), g4 = ld4(g + c);
            if (din.thr16) { d4 = drop4(din, z, row, C, c, d4); t4 = drop4(din, z, row, C, c, t4); }
            const float4 tg4 = tgm ? ld4(tgm + c) : zero4();
            const float d_[4] = {d4.x, d4.y, d4.z, d4.w}, t_[4] = {t4.x, t4.y, t4.z, t4.w}, x_[4] = {x4.x, x4.y, x4.z, x4.w};
            const float tz_[4] = {tz4.x, tz4.y, tz4.z, tz4.w}, g_[4] = {g4.x, g4.y, g4.z, g4.w}, tgm_[4] = {tg4.x, tg4.y, tg4.z, tg4.w};
            float hg_[4], hb_[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                zz[n][k] = x_[k];
                xh[n][k] = (x_[k] - mean) * rstd;
                txh[n][k] = rstd * (tz_[k] - m1 - xh[n][k] * m2);
                gv[n][k] = d_[k] * g_[k];
                tgv[n][k] = t_[k] * g_[k] + d_[k] * tgm_[k];
                a1 += gv[n][k];
                a2 += gv[n][k] * xh[n][k];
                b1 += tgv[n][k];
                b2 += tgv[n][k] * xh[n][k] + gv[n][k] * txh[n][k];
                hg_[k] = t_[k] * xh[n][k] + d_[k] * txh[n][k];
                hb_[k] = t_[k];
            }
            pg[n].x += hg_[0]; pg[n].y += hg_[1]; pg[n].z += hg_[2]; pg[n].w += hg_[3];
            pb[n].x += hb_[0]; pb[n].y += hb_[1]; pb[n].z += hb_[2]; pb[n].w += hb_[3];
        }
        const float invC = 1.f / (float)C;
        a1 = wave_sum(a1) * invC; a2 = wave_sum(a2) * invC; b1 = wave_sum(b1) * invC; b2 = wave_sum(b2) * invC;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c >= C) continue;
            float o1[4], o2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dzk = rstd * (gv[i][k] - a1 - xh[i][k] * a2);
                float tk = -rstd * m2 * dzk + rstd * (tgv[i][k] - b1 - txh[i][k] * a2 - xh[i][k] * b2);
                float pk = dzk;
                if (relu_on_z && !(zz[i][k] > 0.f)) { pk = 0.f; tk = 0.f; }
                o1[k] = pk; o2[k] = tk;
            }
            st4(pdz + c, f4(o1[0], o1[1], o1[2], o1[3]));
            st4(ptg + c, f4(o2[0], o2[1], o2[2], o2[3]));
            if (pdd) {
                st4(pdd + c, drop4(dd, z, row, C, c, f4(o1[0], o1[1], o1[2], o1[3])));
                st4(ptd + c, drop4(dd, z, row, C, c, f4(o2[0], o2[1], o2[2], o2[3])));
            }
        }
    }
    if (!partial) return;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int c = lane * 4 + 256 * n;
        if (c < C) { st4(&red[wave][0][c], pg[n]); st4(&red[wave][1][c], pb[n]); }
    }
    __syncthreads();
    float* out = partial + ((long long)z * max_chunks + blockIdx.x) * 3 * C;   // (wave order: fixed => run-to-run identical)
    for (int idx = (int)threadIdx.x; idx < 2 * C; idx += 256) {
        const int k = idx >= C ? 1 : 0, c = idx - k * C;
        out[(long long)k * C + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// ---- softmax ----------------------------------------------------------------------------------
// in place on tS: tP = P (tS - sum_j P_j tS_j)
__global__ void softmax_jvp_fwd_kernel(const AttnSeq* seqs, const float* P, float* tS) {
    const AttnSeq q = seqs[blockIdx.z];
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= q.L) return;
    const float* p = P + q.s_off + (long long)row * q.ldS;
    float* t = tS + q.s_off + (long long)row * q.ldS;
    float s = 0.f;
    for (int c = lane; c < q.L; c += 64) s += p[c] * t[c];
    s = wave_sum(s);
    for (int c = lane; c < q.ldS; c += 64) t[c] = (c < q.L) ? p[c] * (t[c] - s) : 0.f;
}

// in place: dP -> dS = alpha P (dP - c), tgP -> tg_S = alpha [tP (dP - c) + P (tgP - cdot)],
// c = sum dP P, cdot = sum (tgP P + dP tP)
__global__ void softmax_jvp_bwd_kernel(const AttnSeq* seqs, const float* P, const float* tP, float* dP, float* tgP, float alpha) {
    const AttnSeq q = seqs[blockIdx.z];
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= q.L) return;
    const long long o = q.s_off + (long long)row * q.ldS;
    const float* p = P + o;
    const float* tp = tP + o;
    float* d = dP + o;
    float* tg = tgP + o;
    float c0 = 0.f, c1 = 0.f;
    for (int c = lane; c < q.L; c += 64) { c0 += d[c] * p[c]; c1 += tg[c] * p[c] + d[c] * tp[c]; }
    c0 = wave_sum(c0);
    c1 = wave_sum(c1);
    for (int c = lane; c < q.ldS; c += 64) {
        float o0 = 0.f, o1 = 0.f;
        if (c < q.L) {
            o0 = alpha * p[c] * (d[c] - c0);
            o1 = alpha * (tp[c] * (d[c] - c0) + p[c] * (tg[c] - c1));
        }
        d[c] = o0;
        tg[c] = o1;
    }
}

// ---- BatchNorm (+tanh) ------------------------------------------------------------------------
// forward tangent.  tsum = [S1 | S0] = [sum tc*xhat | sum tc] per channel (colreduce mode 3 on tc, no tanh);
// t_xhat = r (tc - S0/n - xhat S1/n); ty = tgamma xhat + gamma t_xhat + tbeta; ta = act ? (1 - a^2) ty : ty
__global__ void bn_jvp_apply_kernel(const int* meta, const float* X, long long x_ts, const float* tX, long long tx_ts,
                                    const float* stats, long long st_ts, const float* tsum1, const float* tsum0,
                                    long long tsum_ts, const float* gamma, long long par_ts, const float* tgamma,
                                    const float* tbeta, long long tpar_ts, const float* A, long long a_ts,
                                    const unsigned char* inrect, long long row_ts, int do_tanh, float* tA, long long ta_ts,
                                    int C, float yscale, DropSpec dout) {
    // dout: dropout applied to tA before the store (the mask of the layer's forward dropout; a dropout launch of its own until round 6)
    ROW_PROLOGUE(META_MR)
    float* po = tA + (long long)z * ta_ts + (long long)row * C;
    if (!inrect[(long long)z * row_ts + row]) { for (int c = lane * 4; c < C; c += 256) st4(po + c, zero4()); return; }
    const float inv_n = 1.f / (float)(meta[z * META_STRIDE + META_B] * meta[z * META_STRIDE + META_TCAP]);
    const float* px = X + (long long)z * x_ts + (long long)row * C;
    const float* ptx = tX + (long long)z * tx_ts + (long long)row * C;
    const float* pa = A + (long long)z * a_ts + (long long)row * C;
    const float* st = stats + (long long)z * st_ts;
    const float* s1 = tsum1 + (long long)z * tsum_ts;
    const float* s0 = tsum0 + (long long)z * tsum_ts;
    const float* g = gamma + (long long)z * par_ts;
    const float* tg = tgamma ? tgamma + (long long)z * tpar_ts : nullptr;
    const float* tb = tbeta ? tbeta + (long long)z * tpar_ts : nullptr;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 x4 = ld4(px + c), t4 = ld4(ptx + c), mu = ld4(st + c), rs = ld4(st + C + c), g4 = ld4(g + c);
        const float4 q1 = ld4(s1 + c), q0 = ld4(s0 + c), a4 = ld4(pa + c);
        const float4 tg4 = tg ? ld4(tg + c) : zero4(), tb4 = tb ? ld4(tb + c) : zero4();
        const float x_[4] = {x4.x, x4.y, x4.z, x4.w}, t_[4] = {t4.x, t4.y, t4.z, t4.w}, mu_[4] = {mu.x, mu.y, mu.z, mu.w};
        const float rs_[4] = {rs.x, rs.y, rs.z, rs.w}, g_[4] = {g4.x, g4.y, g4.z, g4.w}, q1_[4] = {q1.x, q1.y, q1.z, q1.w};
        const float q0_[4] = {q0.x, q0.y, q0.z, q0.w}, a_[4] = {a4.x, a4.y, a4.z, a4.w}, tg_[4] = {tg4.x, tg4.y, tg4.z, tg4.w};
        const float tb_[4] = {tb4.x, tb4.y, tb4.z, tb4.w};
        float o[4];
        for (int k = 0; k < 4; ++k) {
            const float xh = (x_[k] - mu_[k]) * rs_[k];
            const float txh = rs_[k] * (t_[k] - q0_[k] * inv_n - xh * q1_[k] * inv_n);
            float ty = tg_[k] * xh + g_[k] * txh + tb_[k];
            if (do_tanh) { const float yy = a_[k] * yscale; ty *= (1.f - yy * yy); }
            o[k] = ty;
        }
        float4 o4 = f4(o[0], o[1], o[2], o[3]);
        if (dout.thr16) o4 = drop4(dout, z, row, C, c, o4);
        st4(po + c, o4);
    }
}

// backward, primal + tangent.  Per channel inputs: primal sums dgamma = sum g xhat, dbeta = sum g
// (g = dy (1-a^2) if tanh), tangent sums tA0 = sum (tg xhat + g t_xhat), tA1 = sum tg with
// tg = tgdy (1-a^2) - 2 a ta dy (tanh) or tgdy, and the forward-tangent sums S1, S0.
//   dc  = gamma r (g - dbeta/n - xhat dgamma/n)
//   tdc = r (tgamma - gamma r m2)(g - dbeta/n - xhat dgamma/n)
//         + gamma r (tg - tA1/n - t_xhat dgamma/n - xhat tA0/n),        m2 = S1/n
__global__ void bn_jvp_bwd_kernel(const int* meta, const float* dY, long long dy_ts, const float* tgY, long long tgy_ts,
                                  const float* A, long long a_ts, const float* tA, long long ta_ts, const float* X,
                                  long long x_ts, const float* tX, long long tx_ts, const float* stats, long long st_ts,
                                  const float* tsum1, const float* tsum0, long long tsum_ts, const float* gamma,
                                  long long par_ts, const float* tgamma, long long tpar_ts, const float* dgamma,
                                  const float* dbeta, long long dg_ts, const float* tA0, const float* tA1, long long tA_ts,
                                  const unsigned char* inrect, long long row_ts, int do_tanh, float* dX, long long dx_ts,
                                  float* tdX, long long tdx_ts, int C, float yscale, DropSpec din) {
    // din: the backward of the dropout behind the layer, applied to dY and tgY on load (two dropout launches in front of this kernel until round 6)
    ROW_PROLOGUE(META_MR)
    float* pdx = dX + (long long)z * dx_ts + (long long)row * C;
    float* ptd = tdX + (long long)z * tdx_ts + (long long)row * C;
    if (!inrect[(long long)z * row_ts + row]) {
        for (int c = lane * 4; c < C; c += 256) { st4(pdx + c, zero4()); st4(ptd + c, zero4()); }
        return;
    }
    const float inv_n = 1.f / (float)(meta[z * META_STRIDE + META_B] * meta[z * META_STRIDE + META_TCAP]);
    const long long ro = (long long)row * C;
    const float *pdy = dY + (long long)z * dy_ts + ro, *ptgy = tgY + (long long)z * tgy_ts + ro, *pa = A + (long long)z * a_ts + ro;
    const float *pta = tA + (long long)z * ta_ts + ro, *px = X + (long long)z * x_ts + ro, *ptx = tX + (long long)z * tx_ts + ro;
    const float* st = stats + (long long)z * st_ts;
    const float *s1 = tsum1 + (long long)z * tsum_ts, *s0 = tsum0 + (long long)z * tsum_ts;
    const float* g = gamma + (long long)z * par_ts;
    const float* tgm = tgamma ? tgamma + (long long)z * tpar_ts : nullptr;
    const float *dg = dgamma + (long long)z * dg_ts, *db = dbeta + (long long)z * dg_ts;
    const float *a0 = tA0 + (long long)z * tA_ts, *a1 = tA1 + (long long)z * tA_ts;
    for (int c = lane * 4; c < C; c += 256) {
        float dy_[4], tgy_[4], a_[4], ta_[4], x_[4], tx_[4], mu_[4], rs_[4], q1_[4], q0_[4], g_[4], tgm_[4], dg_[4], db_[4], a0_[4], a1_[4];
        auto L = [&](const float* p, float (&v)[4]) { const float4 t = ld4(p + c); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; };
        L(pdy, dy_); L(ptgy, tgy_);
        if (din.thr16) {
            const float4 t0 = drop4(din, z, row, C, c, f4(dy_[0], dy_[1], dy_[2], dy_[3])), t1 = drop4(din, z, row, C, c, f4(tgy_[0], tgy_[1], tgy_[2], tgy_[3]));
            dy_[0] = t0.x; dy_[1] = t0.y; dy_[2] = t0.z; dy_[3] = t0.w; tgy_[0] = t1.x; tgy_[1] = t1.y; tgy_[2] = t1.z; tgy_[3] = t1.w;
        }
        L(pa, a_); L(pta, ta_); L(px, x_); L(ptx, tx_); L(st, mu_); L(st + C, rs_);
        L(s1, q1_); L(s0, q0_); L(g, g_); L(dg, dg_); L(db, db_); L(a0, a0_); L(a1, a1_);
        if (tgm) L(tgm, tgm_); else { tgm_[0] = tgm_[1] = tgm_[2] = tgm_[3] = 0.f; }
        float o0[4], o1[4];
        for (int k = 0; k < 4; ++k) {
            const float r = rs_[k];
            const float xh = (x_[k] - mu_[k]) * r;
            const float m2 = q1_[k] * inv_n;
            const float txh = r * (tx_[k] - q0_[k] * inv_n - xh * m2);
            float gk = dy_[k], tgk = tgy_[k];
            if (do_tanh) { const float yy = a_[k] * yscale, tyy = ta_[k] * yscale; const float s = 1.f - yy * yy; tgk = tgy_[k] * s - 2.f * yy * tyy * dy_[k]; gk = dy_[k] * s; }
            const float core = gk - db_[k] * inv_n - xh * dg_[k] * inv_n;
            o0[k] = g_[k] * r * core;
            o1[k] = r * (tgm_[k] - g_[k] * r * m2) * core + g_[k] * r * (tgk - a1_[k] * inv_n - txh * dg_[k] * inv_n - xh * a0_[k] * inv_n);
        }
        st4(pdx + c, f4(o0[0], o0[1], o0[2], o0[3]));
        st4(ptd + c, f4(o1[0], o1[1], o1[2], o1[3]));
    }
}

// ---- 256 -> 1 projection ------------------------------------------------------------------------
// tout[row] = valid ? dot(tx, w) + dot(x, tw) + tb : 0
__global__ void rowdot_jvp_kernel(const int* meta, int mfield, const float* x, long long x_ts, const float* tx, long long tx_ts,
                                  const float* w, long long par_ts, const float* tw, const float* tb, long long tpar_ts,
                                  const unsigned char* valid, long long row_ts, float* tout, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const float* px = x + (long long)z * x_ts + (long long)row * C;
    const float* ptx = tx + (long long)z * tx_ts + (long long)row * C;
    const float* pw = w + (long long)z * par_ts;
    const float* ptw = tw ? tw + (long long)z * tpar_ts : nullptr;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = ld4(px + c), ta = ld4(ptx + c), ww = ld4(pw + c);
        s += (ta.x * ww.x + ta.y * ww.y) + (ta.z * ww.z + ta.w * ww.w);
        if (ptw) { const float4 t = ld4(ptw + c); s += (a.x * t.x + a.y * t.y) + (a.z * t.z + a.w * t.w); }
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float b = tb ? tb[(long long)z * tpar_ts] : 0.f;
        tout[(long long)z * out_ts + row] = valid[(long long)z * row_ts + row] ? s + b : 0.f;
    }
}

// dx = dout w (primal) ; tdx = tgout w + dout tw
__global__ void rowdot_jvp_bwd_kernel(const int* meta, int mfield, const float* dout, const float* tgout, long long dout_ts,
                                      const float* w, long long par_ts, const float* tw, long long tpar_ts, float* dx,
                                      long long dx_ts, float* tdx, long long tdx_ts, int C) {
    ROW_PROLOGUE(mfield)
    const float d = dout[(long long)z * dout_ts + row], td = tgout[(long long)z * dout_ts + row];
    const float* pw = w + (long long)z * par_ts;
    const float* ptw = tw ? tw + (long long)z * tpar_ts : nullptr;
    float* pd = dx + (long long)z * dx_ts + (long long)row * C;
    float* pt = tdx + (long long)z * tdx_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 ww = ld4(pw + c);
        const float4 t = ptw ? ld4(ptw + c) : zero4();
        st4(pd + c, f4(d * ww.x, d * ww.y, d * ww.z, d * ww.w));
        st4(pt + c, f4(td * ww.x + d * t.x, td * ww.y + d * t.y, td * ww.z + d * t.z, td * ww.w + d * t.w));
    }
}

// ---- loss: tangent of the prediction gradients (MSE terms; the L1 terms have zero second derivative) ----
__global__ void loss_tangent_kernel(const int* meta, const float* tpp, const float* tep, const float* tlogd, long long pred_ts,
                                    const unsigned char* pvalid, long long prow_ts, float scale, float* tgpp, float* tgep,
                                    float* tglogd, int pitch_frame, int energy_frame) {
    const int z = blockIdx.z, Mp = meta[z * META_STRIDE + META_MP];
    const float wP = 2.f * scale / (float)meta[z * META_STRIDE + META_NP];
    const unsigned char* pv = pvalid + (long long)z * prow_ts;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mp; r += gridDim.x * blockDim.x) {
        const long long q = (long long)z * pred_ts + r;
        const bool v = pv[r] != 0;
        tgpp[q] = (v && !pitch_frame) ? wP * tpp[q] : 0.f;      // (a frame-level feature has no phoneme-level loss term)
        tgep[q] = (v && !energy_frame) ? wP * tep[q] : 0.f;
        tglogd[q] = v ? wP * tlogd[q] : 0.f;
    }
}
// the same for frame-level pitch / energy predictions (rows of the mel rectangle, normalised by the number of valid frames)
__global__ void loss_tangent_r_kernel(const int* meta, const float* tpp_r, const float* tep_r, long long pred_r_ts,
                                      const unsigned char* rvalid, long long rrow_ts, float scale, float* tgpp_r, float* tgep_r,
                                      int pitch_frame, int energy_frame) {
    const int z = blockIdx.z, Mr = meta[z * META_STRIDE + META_MR];
    const float wR = 2.f * scale / (float)meta[z * META_STRIDE + META_NF];
    const unsigned char* rv = rvalid + (long long)z * rrow_ts;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Mr; r += gridDim.x * blockDim.x) {
        const long long q = (long long)z * pred_r_ts + r;
        if (pitch_frame) tgpp_r[q] = rv[r] ? wR * tpp_r[q] : 0.f;
        if (energy_frame) tgep_r[q] = rv[r] ? wR * tep_r[q] : 0.f;
    }
}

// ---- elementwise: dst[t][i] += alpha * src[t][i];  dst = a + b over row-space slabs ---------------------
__global__ void axpy_kernel(float* dst, long long dst_ts, const float* src, long long src_ts, float alpha, long long n4) {
    float* d = dst + (long long)blockIdx.z * dst_ts;
    const float* s = src + (long long)blockIdx.z * src_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 x = ld4(d + i * 4);
        const float4 y = ld4(s + i * 4);
        x.x += alpha * y.x; x.y += alpha * y.y; x.z += alpha * y.z; x.w += alpha * y.w;
        st4(d + i * 4, x);
    }
}

// out[row] = (x ? x[row] : 0) + (table && idx[row] >= 0 ? table[idx[row]] : 0)   (tangent of x + emb[bucket])
__global__ void embed_add_idx_kernel(const int* meta, int mfield, const float* x, long long x_ts, const float* table,
                                     long long table_ts, const int* idx, long long idx_ts, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const int id = idx[(long long)z * idx_ts + row];
    const float* px = x ? x + (long long)z * x_ts + (long long)row * C : nullptr;
    const float* pt = (table && id >= 0) ? table + (long long)z * table_ts + (long long)id * C : nullptr;
    float* po = out + (long long)z * out_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 o = px ? ld4(px + c) : zero4();
        if (pt) { const float4 t = ld4(pt + c); o = f4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w); }
        st4(po + c, o);
    }
}

// out[row] = inrect ? vec[row_b[row]] : 0   (tangent of enc_out + spk: the encoder carries no tangent)
__global__ void bcast_rowvec_kernel(const int* meta, int mfield, const float* vec, long long vec_ts, const int* row_b,
                                    const unsigned char* inrect, long long row_ts, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const long long r = (long long)z * row_ts + row;
    const bool in = inrect[r] != 0;
    const float* pv = vec + (long long)z * vec_ts + (long long)(in ? row_b[r] : 0) * C;
    float* po = out + (long long)z * out_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) st4(po + c, in ? ld4(pv + c) : zero4());
}

// row-space copy: out[r] = in[r] for r < M (all channels)
__global__ void copy_rows_kernel(const int* meta, int mfield, const float* in, long long in_ts, float* out, long long out_ts, int C) {
    ROW_PROLOGUE(mfield)
    const float* pi = in + (long long)z * in_ts + (long long)row * C;
    float* po = out + (long long)z * out_ts + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) st4(po + c, ld4(pi + c));
}


// ---- iMAML (engine_imaml.inc): proximal SGD step and the conjugate-gradient vector updates; per-task scalars live in
// scal[task * stride + {rs, pAp, alpha, beta, rs_new, active, norm, coef}] --------------------------------------------------
// w <- w - lr * (g + reg * (w - theta))      (imaml.py:69: loss = L + 0.5 * reg * sum (theta - w)^2, one SGD step)
__global__ void sgd_prox_kernel(float* w, const float* g, const float* theta, long long n4, float lr, float reg, long long w_ts, long long g_ts) {
    float* pw = w + (long long)blockIdx.z * w_ts;
    const float* pg = g + (long long)blockIdx.z * g_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 x = ld4(pw + i * 4);
        const float4 d = ld4(pg + i * 4), t = ld4(theta + i * 4);
        x.x -= lr * (d.x + reg * (x.x - t.x)); x.y -= lr * (d.y + reg * (x.y - t.y));
        x.z -= lr * (d.z + reg * (x.z - t.z)); x.w -= lr * (d.w + reg * (x.w - t.w));
        st4(pw + i * 4, x);
    }
}
__global__ void copy_tasks_kernel(const float* src, long long src_ts, float* dst, long long dst_ts, long long n4) {
    const float* s = src + (long long)blockIdx.z * src_ts;
    float* d = dst + (long long)blockIdx.z * dst_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) st4(d + i * 4, ld4(s + i * 4));
}
// partial[task][block] = sum over this block's elements of a * (ca * b + cb * a)
__global__ void dot_partial_kernel(const float* a, long long a_ts, const float* b, long long b_ts, float ca, float cb, long long n4, float* partial,
                                   int nblocks) {
    __shared__ float red[4];
    const float* pa = a + (long long)blockIdx.z * a_ts;
    const float* pb = b + (long long)blockIdx.z * b_ts;
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ld4(pa + i * 4), y = ld4(pb + i * 4);
        s += (x.x * (ca * y.x + cb * x.x) + x.y * (ca * y.y + cb * x.y)) + (x.z * (ca * y.z + cb * x.z) + x.w * (ca * y.w + cb * x.w));
    }
    s = wave_sum(s);
    if (((int)threadIdx.x & 63) == 0) red[(int)threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)blockIdx.z * nblocks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void dot_final_kernel(const float* partial, int nblocks, float* scal, int stride, int slot) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += (double)partial[(long long)blockIdx.x * nblocks + i];
    scal[blockIdx.x * stride + slot] = (float)s;
}
// mode 0: start (active = 1); 1: alpha = rs / pAp; 2: after the residual update — a task whose residual norm fell below tol
// stops WITHOUT adopting this iteration's x (hypergrad's cg breaks before `x_last = x`), else beta = rs_new / rs, rs <- rs_new
__global__ void cg_scalar_kernel(float* scal, int tasks, int mode, float tol) {
    const int t = (int)threadIdx.x;
    if (t >= tasks) return;
    float* s = scal + t * 8;
    if (mode == 0) { s[5] = 1.f; s[2] = 0.f; s[3] = 0.f; return; }
    if (mode == 1) { s[2] = (s[5] != 0.f && s[1] != 0.f) ? s[0] / s[1] : 0.f; return; }
    if (s[5] == 0.f) { s[3] = 0.f; return; }
    if (sqrtf(s[4]) < tol) { s[5] = 0.f; s[3] = 0.f; return; }
    s[3] = s[4] / s[0];
    s[0] = s[4];
}
// r -= alpha * (ca * Hp + cb * p)
__global__ void cg_update_r_kernel(float* r, long long r_ts, const float* p, const float* Hp, long long p_ts, float ca, float cb,
                                   const float* scal, int stride, long long n4) {
    const int z = blockIdx.z;
    const float alpha = scal[z * stride + 2];
    if (alpha == 0.f) return;
    float* pr = r + (long long)z * r_ts;
    const float* pp = p + (long long)z * p_ts;
    const float* ph = Hp + (long long)z * p_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 R = ld4(pr + i * 4);
        const float4 P = ld4(pp + i * 4), H = ld4(ph + i * 4);
        R.x -= alpha * (ca * H.x + cb * P.x); R.y -= alpha * (ca * H.y + cb * P.y);
        R.z -= alpha * (ca * H.z + cb * P.z); R.w -= alpha * (ca * H.w + cb * P.w);
        st4(pr + i * 4, R);
    }
}
// still-active tasks: x += alpha p ; p = r + beta p
__global__ void cg_update_xp_kernel(float* x, long long x_ts, float* p, long long p_ts, const float* r, const float* scal, int stride, long long n4) {
    const int z = blockIdx.z;
    if (scal[z * stride + 5] == 0.f) return;
    const float alpha = scal[z * stride + 2], beta = scal[z * stride + 3];
    float* px = x + (long long)z * x_ts;
    float* pp = p + (long long)z * p_ts;
    const float* pr = r + (long long)z * x_ts;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 X = ld4(px + i * 4), P = ld4(pp + i * 4);
        const float4 R = ld4(pr + i * 4);
        X.x += alpha * P.x; X.y += alpha * P.y; X.z += alpha * P.z; X.w += alpha * P.w;
        P.x = R.x + beta * P.x; P.y = R.y + beta * P.y; P.z = R.z + beta * P.z; P.w = R.w + beta * P.w;
        st4(px + i * 4, X); st4(pp + i * 4, P);
    }
}
// scal[norm] holds |x|^2 on entry: norm = scale_g * |x|, coef = grad_scale * scale_g * min(1, max_norm / (norm + 1e-6))
__global__ void cg_clip_kernel(float* scal, int tasks, float scale_g, float max_norm, float grad_scale) {
    const int t = (int)threadIdx.x;
    if (t >= tasks) return;
    float* s = scal + t * 8;
    const float norm = fabsf(scale_g) * sqrtf(s[6]);
    float c = 1.f;
    if (max_norm > 0.f) { c = max_norm / (norm + 1e-6f); c = c > 1.f ? 1.f : c; }
    s[6] = norm;
    s[7] = grad_scale * scale_g * c;
}
// out[i] = sum_t scal[t][slot] * g[t][i]   (fixed task order)
__global__ void sum_tasks_coef_kernel(const float* g, long long g_ts, int tasks, const float* scal, int stride, int slot, float* out, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 s = zero4();
        for (int t = 0; t < tasks; ++t) {
            const float c = scal[t * stride + slot];
            const float4 x = ld4(g + (long long)t * g_ts + i * 4);
            s.x += c * x.x; s.y += c * x.y; s.z += c * x.z; s.w += c * x.w;
        }
        st4(out + i * 4, s);
    }
}

}  // namespace mtts
