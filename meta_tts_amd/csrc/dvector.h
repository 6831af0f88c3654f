// d-vector speaker encoder, forward only — the `dvec` speaker mode (config/algorithm/dvec.yaml: `speaker_emb: dvec`, frozen).
//
// Reference: lightning/model/speaker_encoder.py:11-31 (GE2E: nn.LSTM(40, 256, 3, batch_first=True) + nn.Linear(256, 256) + ReLU —
// the architecture of the un-vendored resemblyzer `VoiceEncoder` the `dvec` / `encoder` modes instantiate, :54-60), its forward
// (final hidden state of the last layer -> Linear -> ReLU -> L2 normalisation per partial utterance) and :71-76 (utterance
// embedding = L2-normalised mean of the partial embeddings of that utterance's slice).  The reference runs this encoder on the
// CPU (`VoiceEncoder('cpu')`) in front of every forward of the acoustic model.
//
// MI355X layout: the input projection of a layer is ONE GEMM over all (partial, frame) rows (gemm.h, bias = b_ih + b_hh fused);
// the recurrence runs one workgroup per partial utterance with the hidden state in LDS and the recurrent weights read from L2
// as a transposed image ([H][4H]: thread j of the workgroup owns hidden unit j and reads the four gate columns of every row
// coalesced); all partials advance in parallel, the 1 MB weight image is shared through the L2.  Gate order i, f, g, o (torch).
#pragma once
#include <string>
#include <vector>

#include "gemm.h"
#include "rowops.h"
#include "tangent.h"

namespace mtts {

__device__ __forceinline__ float dv_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// xp: [N][T][4H] input projections (biases included); whhT: [H][4H]; hseq: [N][T][H] (all hidden states, the next layer's input);
// hlast: [N][H] final hidden state.  blockDim.x == H.  Training passes also keep what the backward sweep needs: gates [N][T][4H]
// (i, f, g, o AFTER their non-linearities), cseq [N][T][H], hprev [N][T][H] (the hidden state each step started from).
__global__ void lstm_recurrent_kernel(const float* xp, const float* whhT, float* hseq, float* hlast, int T, int H, float* gates, float* cseq,
                                      float* hprev) {
    __shared__ float dv_smem[2 * 1024];   // hidden state, double-buffered (H <= 1024)
    const int n = blockIdx.x, j = threadIdx.x;
    float* h0 = dv_smem;
    float* h1 = dv_smem + H;
    h0[j] = 0.f;
    float c = 0.f, hv = 0.f;
    __syncthreads();
    const float* px = xp + (long long)n * T * 4 * H;
    for (int t = 0; t < T; ++t) {
        const float* hin = (t & 1) ? h1 : h0;
        float* hout = (t & 1) ? h0 : h1;
        float gi = px[(long long)t * 4 * H + j], gf = px[(long long)t * 4 * H + H + j];
        float gg = px[(long long)t * 4 * H + 2 * H + j], go = px[(long long)t * 4 * H + 3 * H + j];
        const float* w = whhT + j;
#pragma unroll 4
        for (int k = 0; k < H; ++k) {
            const float hk = hin[k];
            gi += w[0] * hk; gf += w[H] * hk; gg += w[2 * H] * hk; go += w[3 * H] * hk;
            w += 4 * H;
        }
        gi = dv_sigmoid(gi); gf = dv_sigmoid(gf); gg = tanhf(gg); go = dv_sigmoid(go);
        c = gf * c + gi * gg;
        const long long row = (long long)n * T + t;
        if (gates) {
            float* pg = gates + row * 4 * H;
            pg[j] = gi; pg[H + j] = gf; pg[2 * H + j] = gg; pg[3 * H + j] = go;
            cseq[row * H + j] = c;
            hprev[row * H + j] = hv;
        }
        hv = go * tanhf(c);
        hout[j] = hv;
        if (hseq) hseq[row * H + j] = hv;
        __syncthreads();
    }
    hlast[(long long)n * H + j] = hv;
}

// Backward sweep of one layer (BPTT), one workgroup per partial utterance, thread j = hidden unit j.  dh_ext [N][T][H]: gradient
// reaching h_t from the layer above (null for the top layer); dh_last [N][H]: gradient of the final hidden state (top layer only,
// else null); whh [4H][H] in torch's layout (row r read coalesced over j).  Writes dgates [N][T][4H]: gradients of the gate
// pre-activations, from which the weight / bias / input gradients are plain GEMMs.
__global__ void lstm_bptt_kernel(const float* gates, const float* cseq, const float* dh_ext, const float* dh_last, const float* whh, float* dgates,
                                 int T, int H) {
    __shared__ float dg_s[4 * 1024];
    const int n = blockIdx.x, j = threadIdx.x;
    float dh_rec = 0.f, dc_next = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const long long row = (long long)n * T + t;
        const float* pg = gates + row * 4 * H;
        const float gi = pg[j], gf = pg[H + j], gg = pg[2 * H + j], go = pg[3 * H + j];
        const float c = cseq[row * H + j], c_prev = t > 0 ? cseq[(row - 1) * H + j] : 0.f;
        const float tc = tanhf(c);
        float dh = dh_rec;
        if (dh_ext) dh += dh_ext[row * H + j];
        if (dh_last && t == T - 1) dh += dh_last[(long long)n * H + j];
        const float dc = dc_next + dh * go * (1.f - tc * tc);
        const float ai = dc * gg * gi * (1.f - gi), af = dc * c_prev * gf * (1.f - gf);
        const float ag = dc * gi * (1.f - gg * gg), ao = dh * tc * go * (1.f - go);
        dc_next = dc * gf;
        float* po = dgates + row * 4 * H;
        po[j] = ai; po[H + j] = af; po[2 * H + j] = ag; po[3 * H + j] = ao;
        dg_s[j] = ai; dg_s[H + j] = af; dg_s[2 * H + j] = ag; dg_s[3 * H + j] = ao;
        __syncthreads();
        float acc = 0.f;
        const float* w = whh + j;
#pragma unroll 4
        for (int r = 0; r < 4 * H; ++r) { acc += dg_s[r] * w[0]; w += H; }
        dh_rec = acc;
        __syncthreads();
    }
}

// e = relu(W h + b), then e / ||e||   (one workgroup per partial utterance, blockDim.x = E rounded up to whole wavefronts; wT: [H][E]);
// eraw (optional): e before the normalisation, for the backward pass
__global__ void dvec_head_kernel(const float* hlast, const float* wT, const float* bias, float* part, int H, int E, float* eraw) {
    __shared__ float dv_smem[1024 + 4];
    const int n = blockIdx.x, j = threadIdx.x;
    for (int k = j; k < H; k += blockDim.x) dv_smem[k] = hlast[(long long)n * H + k];
    __syncthreads();
    float e = 0.f;
    if (j < E) {
        e = bias[j];
        for (int k = 0; k < H; ++k) e += wT[(long long)k * E + j] * dv_smem[k];
        e = e > 0.f ? e : 0.f;
    }
    const float ss = wave_sum(e * e);
    if ((j & 63) == 0) dv_smem[H + (j >> 6)] = ss;   // E <= 256: at most four wavefronts
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) tot += dv_smem[H + w];
    if (j < E) { part[(long long)n * E + j] = e / sqrtf(tot); if (eraw) eraw[(long long)n * E + j] = e; }
}

// backward of the utterance reduction: out_b = m / max(||m||, 1e-12), m = mean of the utterance's partials -> dpart[n] for its partials
__global__ void dvec_utterance_bwd_kernel(const float* part, const int* off, const float* dout, float* dpart, int E) {
    __shared__ float red[2][4];
    const int b = blockIdx.x, j = threadIdx.x;
    const int lo = off[b], hi = off[b + 1];
    if (hi <= lo) return;
    float m = 0.f, d = 0.f;
    if (j < E) {
        for (int n = lo; n < hi; ++n) m += part[(long long)n * E + j];
        m /= (float)(hi - lo);
        d = dout[(long long)b * E + j];
    }
    const float s0 = wave_sum(m * m), s1 = wave_sum(m * d);
    if ((j & 63) == 0) { red[0][j >> 6] = s0; red[1][j >> 6] = s1; }
    __syncthreads();
    float mm = 0.f, md = 0.f;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) { mm += red[0][w]; md += red[1][w]; }
    const float nrm = sqrtf(mm);
    float dm;
    if (nrm > 1e-12f) dm = (d - m * md / mm) / nrm;   // d(m / ||m||)
    else dm = d / 1e-12f;                              // clamped branch of F.normalize
    dm /= (float)(hi - lo);
    if (j < E) for (int n = lo; n < hi; ++n) dpart[(long long)n * E + j] = dm;
}

// backward of the head: y = e / ||e||, e = relu(z): dz [N][E] (gradient of the Linear output) and dh_last [N][H] = dz W  (w: [E][H])
__global__ void dvec_head_bwd_kernel(const float* eraw, const float* dpart, const float* w, float* dz, float* dh_last, int H, int E) {
    __shared__ float dz_s[256];
    __shared__ float red[2][4];
    const int n = blockIdx.x, j = threadIdx.x;
    float e = 0.f, d = 0.f;
    if (j < E) { e = eraw[(long long)n * E + j]; d = dpart[(long long)n * E + j]; }
    const float s0 = wave_sum(e * e), s1 = wave_sum(e * d);
    if ((j & 63) == 0) { red[0][j >> 6] = s0; red[1][j >> 6] = s1; }
    __syncthreads();
    float ee = 0.f, ed = 0.f;
    for (int q = 0; q < (int)(blockDim.x + 63) / 64; ++q) { ee += red[0][q]; ed += red[1][q]; }
    const float nrm = sqrtf(ee);
    float g = 0.f;
    if (j < E) {
        g = (d - e * ed / ee) / nrm;   // through the normalisation
        if (!(e > 0.f)) g = 0.f;       // through the ReLU
        dz[(long long)n * E + j] = g;
        dz_s[j] = g;
    }
    __syncthreads();
    for (int k = j; k < H; k += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < E; ++r) acc += dz_s[r] * w[(long long)r * H + k];
        dh_last[(long long)n * H + k] = acc;
    }
}

// utterance b: mean of the partial embeddings [off[b], off[b+1]), then F.normalize (x / max(||x||, 1e-12))
__global__ void dvec_utterance_kernel(const float* part, const int* off, float* out, int E) {
    __shared__ float red[4];
    const int b = blockIdx.x, j = threadIdx.x;
    const int lo = off[b], hi = off[b + 1];
    float m = 0.f;
    if (j < E) {
        for (int n = lo; n < hi; ++n) m += part[(long long)n * E + j];
        m = hi > lo ? m / (float)(hi - lo) : 0.f;
    }
    const float ss = wave_sum(m * m);
    if ((j & 63) == 0) red[j >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) tot += red[w];
    const float nrm = sqrtf(tot);
    if (j < E) out[(long long)b * E + j] = m / (nrm > 1e-12f ? nrm : 1e-12f);
}

// dst[c][r] = src[r][c]
__global__ void dv_transpose_kernel(const float* src, float* dst, int R, int Cc) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)R * Cc; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / Cc), c = (int)(i % Cc);
        dst[(long long)c * R + r] = src[i];
    }
}
__global__ void dv_add_kernel(const float* a, const float* b, float* o, int n) {
    for (int i = blockIdx.x * (int)blockDim.x + threadIdx.x; i < n; i += (int)(gridDim.x * blockDim.x)) o[i] = a[i] + b[i];
}
__global__ void dv_fill_kernel(float* o, float v, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) o[i] = v;
}
__global__ void dv_sum_partials_kernel(const float* partial, int n, float* out) {   // out[0] = sum (NOT its root: a term of a joint norm)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)partial[i];
        out[0] = (float)s;
    }
}

class DVector {
public:
    int n_mels = 40, H = 256, layers = 3, E = 256, cap_N = 0, T = 160, cap_B = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    GemmCtx gx;
    struct Tensor { std::string name; long long off, numel; };
    std::vector<Tensor> tensors;
    float* params = nullptr;
    long long n_params = 0;
    // derived images, rebuilt by load(): transposed recurrent / head weights and the summed biases
    float *whhT = nullptr, *linT = nullptr, *bsum = nullptr;
    float *mels = nullptr, *xp = nullptr, *hseq[2] = {nullptr, nullptr}, *hlast = nullptr, *part = nullptr, *out = nullptr;
    int* off_dev = nullptr;
    bool dirty = true;
    // ---- training state (speaker_emb: encoder / scratch_encoder; allocated by enable_training) ----
    bool train_ready = false;
    std::vector<float*> gates, cseq, hprev, hkeep;   // per layer: [rows][4H], [rows][H], [rows][H], [rows][H] (layer output = next layer's input)
    float *dgates = nullptr, *dxbuf[2] = {nullptr, nullptr}, *eraw = nullptr, *dpart = nullptr, *dz = nullptr, *dh_last = nullptr, *dout = nullptr;
    float *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *ones4 = nullptr, *sq_partial = nullptr, *sq_out = nullptr;
    int last_N = 0, last_B = 0, adam_steps = 0;
    bool have_forward = false;

    void set_error(const std::string& s) { last_error = s; }
#define DV_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return -1; } } while (0)
    long long find(const std::string& n) const { for (auto& t : tensors) if (t.name == n) return t.off; return -1; }
    int in_dim(int l) const { return l == 0 ? n_mels : H; }

    int init(int n_mels_, int hidden, int layers_, int emb, int max_partials, int frames, int max_utts) {
        n_mels = n_mels_; H = hidden; layers = layers_; E = emb; cap_N = max_partials; T = frames; cap_B = max_utts;
        if (n_mels < 4 || (n_mels & 3) || H < 64 || H > 1024 || (H & 63) || E < 4 || E > 256 || layers < 1 || layers > 8 || cap_N < 1 || T < 1 ||
            cap_B < 1) {
            set_error("unsupported d-vector configuration (n_mels % 4, hidden % 64 <= 1024, emb <= 256)");
            return -1;
        }
        auto add = [&](const std::string& n, long long numel) { tensors.push_back(Tensor{n, n_params, numel}); n_params += (numel + 3) & ~3LL; };
        for (int l = 0; l < layers; ++l) {
            const std::string s = std::to_string(l);
            add("lstm.weight_ih_l" + s, 4LL * H * in_dim(l)); add("lstm.weight_hh_l" + s, 4LL * H * H);
            add("lstm.bias_ih_l" + s, 4LL * H); add("lstm.bias_hh_l" + s, 4LL * H);
        }
        add("linear.weight", (long long)E * H); add("linear.bias", E);
        DV_CHECK(hipMalloc((void**)&params, (size_t)n_params * sizeof(float)));
        DV_CHECK(hipMemset(params, 0, (size_t)n_params * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&whhT, (size_t)layers * 4 * H * H * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&linT, (size_t)E * H * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&bsum, (size_t)layers * 4 * H * sizeof(float)));
        const size_t rows = (size_t)cap_N * T;
        DV_CHECK(hipMalloc((void**)&mels, (rows * n_mels + 64) * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&xp, rows * 4 * H * sizeof(float)));
        for (int i = 0; i < 2; ++i) DV_CHECK(hipMalloc((void**)&hseq[i], (rows * H + 64) * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&hlast, (size_t)cap_N * H * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&part, (size_t)cap_N * E * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&out, (size_t)cap_B * E * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&off_dev, (size_t)(cap_B + 1) * sizeof(int)));
        if (gx.alloc_workspace() != 0) { set_error("split-K workspace allocation failed"); return -1; }
        return 0;
    }
    void destroy() {
        for (float* p : {params, whhT, linT, bsum, mels, xp, hseq[0], hseq[1], hlast, part, out}) if (p) hipFree(p);
        if (off_dev) hipFree(off_dev);
        for (auto* v : {&gates, &cseq, &hprev, &hkeep}) for (float* p : *v) if (p) hipFree(p);
        for (float* p : {dgates, dxbuf[0], dxbuf[1], eraw, dpart, dz, dh_last, dout, grads, adam_m, adam_v, ones4, sq_partial, sq_out}) if (p) hipFree(p);
        gx.release();
    }
    int enable_training() {
        if (train_ready) return 0;
        const size_t rows = (size_t)cap_N * T;
        gates.assign(layers, nullptr); cseq.assign(layers, nullptr); hprev.assign(layers, nullptr); hkeep.assign(layers, nullptr);
        for (int l = 0; l < layers; ++l) {
            DV_CHECK(hipMalloc((void**)&gates[l], (rows * 4 * H + 64) * sizeof(float)));
            DV_CHECK(hipMalloc((void**)&cseq[l], (rows * H + 64) * sizeof(float)));
            DV_CHECK(hipMalloc((void**)&hprev[l], (rows * H + 64) * sizeof(float)));
            DV_CHECK(hipMalloc((void**)&hkeep[l], (rows * H + 64) * sizeof(float)));
        }
        DV_CHECK(hipMalloc((void**)&dgates, (rows * 4 * H + 64) * sizeof(float)));
        for (int i = 0; i < 2; ++i) DV_CHECK(hipMalloc((void**)&dxbuf[i], (rows * H + 64) * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&eraw, (size_t)cap_N * E * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&dpart, (size_t)cap_N * E * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&dz, ((size_t)cap_N * E + 64) * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&dh_last, (size_t)cap_N * H * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&dout, (size_t)cap_B * E * sizeof(float)));
        for (float** p : {&grads, &adam_m, &adam_v}) { DV_CHECK(hipMalloc((void**)p, (size_t)n_params * sizeof(float))); DV_CHECK(hipMemset(*p, 0, (size_t)n_params * sizeof(float))); }
        DV_CHECK(hipMalloc((void**)&ones4, (rows * 4 + 64) * sizeof(float)));
        MTTS_LAUNCH(dv_fill_kernel, dim3(256), dim3(256), stream, ones4, 1.f, (long long)(rows * 4));
        DV_CHECK(hipMalloc((void**)&sq_partial, 512 * sizeof(float)));
        DV_CHECK(hipMalloc((void**)&sq_out, 4 * sizeof(float)));
        DV_CHECK(hipMemset(sq_out, 0, 4 * sizeof(float)));
        train_ready = true;
        return 0;
    }
    int load(const char* name, const float* host, long long numel) {
        for (auto& t : tensors)
            if (t.name == name) {
                if (t.numel != numel) { set_error(std::string("size mismatch for ") + name); return -1; }
                DV_CHECK(hipMemcpy(params + t.off, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
                dirty = true;
                return 0;
            }
        set_error(std::string("unknown d-vector tensor ") + name);
        return -1;
    }
    // transposed images + summed biases, rebuilt on the device whenever the weights changed (load / adam_step)
    int refresh() {
        for (int l = 0; l < layers; ++l) {
            const std::string s = std::to_string(l);
            MTTS_LAUNCH(dv_transpose_kernel, dim3(512), dim3(256), stream, (const float*)(params + find("lstm.weight_hh_l" + s)),
                        whhT + (long long)l * 4 * H * H, 4 * H, H);
            MTTS_LAUNCH(dv_add_kernel, dim3(8), dim3(256), stream, (const float*)(params + find("lstm.bias_ih_l" + s)),
                        (const float*)(params + find("lstm.bias_hh_l" + s)), bsum + (long long)l * 4 * H, 4 * H);
        }
        MTTS_LAUNCH(dv_transpose_kernel, dim3(256), dim3(256), stream, (const float*)(params + find("linear.weight")), linT, E, H);
        DV_CHECK(hipGetLastError());
        dirty = false;
        return 0;
    }
    // mels_host [N][T][n_mels]; utt_off [B+1] partial offsets (utt_off[0] = 0, utt_off[B] = N); out_host [B][E]; part_host [N][E] or null
    int embed(const float* mels_host, int N, const int* utt_off, int B, float* out_host, float* part_host, bool train = false) {
        if (train && !train_ready) { set_error("mtts_dvector_enable_training first"); return -1; }
        if (!mels_host || !utt_off || !out_host || N < 1 || N > cap_N || B < 1 || B > cap_B) { set_error("bad d-vector arguments"); return -1; }
        if (utt_off[0] != 0 || utt_off[B] != N) { set_error("utterance offsets must cover [0, N)"); return -1; }
        for (int b = 0; b < B; ++b)   // an utterance without partials has no embedding (the reference's mean over an empty slice is NaN)
            if (utt_off[b + 1] <= utt_off[b]) { set_error("every utterance needs at least one partial utterance (offsets must increase)"); return -1; }
        if (dirty && refresh() != 0) return -1;
        const long long rows = (long long)N * T;
        DV_CHECK(hipMemcpyAsync(mels, mels_host, (size_t)rows * n_mels * sizeof(float), hipMemcpyHostToDevice, stream));
        DV_CHECK(hipMemcpyAsync(off_dev, utt_off, (size_t)(B + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
        const float* x = mels;
        for (int l = 0; l < layers; ++l) {
            const std::string s = std::to_string(l);
            GemmArgs g;
            g.A = x; g.lda = in_dim(l);
            g.B = params + find("lstm.weight_ih_l" + s); g.ldb = in_dim(l);
            g.C = xp; g.ldc = 4 * H;
            g.M = (int)rows; g.N = 4 * H; g.K = in_dim(l);
            g.bias = bsum + (long long)l * 4 * H;
            gemm_launch(gx, GEMM_NT, g, (int)rows, 4 * H, 1, stream, 0, 2.0 * rows * 4.0 * H * in_dim(l), 0);
            float* hs = train ? hkeep[l] : ((l + 1 < layers) ? hseq[l & 1] : nullptr);
            MTTS_LAUNCH(lstm_recurrent_kernel, dim3((unsigned)N), dim3((unsigned)H), stream, (const float*)xp,
                        (const float*)(whhT + (long long)l * 4 * H * H), hs, hlast, T, H, train ? gates[l] : (float*)nullptr,
                        train ? cseq[l] : (float*)nullptr, train ? hprev[l] : (float*)nullptr);
            x = hs;
        }
        MTTS_LAUNCH(dvec_head_kernel, dim3((unsigned)N), dim3((unsigned)((E + 63) & ~63)), stream, (const float*)hlast, (const float*)linT,
                    (const float*)(params + find("linear.bias")), part, H, E, train ? eraw : (float*)nullptr);
        if (train) { last_N = N; last_B = B; have_forward = true; }
        MTTS_LAUNCH(dvec_utterance_kernel, dim3((unsigned)B), dim3((unsigned)((E + 63) & ~63)), stream, (const float*)part, (const int*)off_dev, out, E);
        DV_CHECK(hipGetLastError());
        DV_CHECK(hipMemcpyAsync(out_host, out, (size_t)B * E * sizeof(float), hipMemcpyDeviceToHost, stream));
        if (part_host) DV_CHECK(hipMemcpyAsync(part_host, part, (size_t)N * E * sizeof(float), hipMemcpyDeviceToHost, stream));
        DV_CHECK(hipStreamSynchronize(stream));
        return 0;
    }

    // ---- training: gradient of the utterance embeddings -> parameter gradients (BPTT), joint-norm term, Adam -------------------
    // TN GEMM dW[M][Ncols] = A[rows][M]^T * B[rows][Ncols]; optional column sums of A (bias gradient) ride along (gemm.h: colsum)
    void wgrad(const float* A, int M, const float* Bm, int Ncols, long long rows, float* dW, float* db) {
        GemmArgs g;
        g.A = A; g.lda = M; g.B = Bm; g.ldb = Ncols; g.C = dW; g.ldc = Ncols;
        g.M = M; g.N = Ncols; g.K = (int)rows;
        if (db) { g.colsum = db; g.colsum_w = ones4; }
        gemm_launch(gx, GEMM_TN, g, M, Ncols, 1, stream, 0, 2.0 * rows * (double)M * Ncols, 0);
    }
    // dout_host [B][E]: gradient of the loss w.r.t. the embeddings returned by the last embed(train = true)
    int backward(const float* dout_host) {
        if (!train_ready || !have_forward || !dout_host) { set_error("d-vector backward without a training forward"); return -1; }
        const int N = last_N, B = last_B;
        const long long rows = (long long)N * T;
        const unsigned eb = (unsigned)((E + 63) & ~63);
        DV_CHECK(hipMemsetAsync(grads, 0, (size_t)n_params * sizeof(float), stream));
        DV_CHECK(hipMemcpyAsync(dout, dout_host, (size_t)B * E * sizeof(float), hipMemcpyHostToDevice, stream));
        MTTS_LAUNCH(dvec_utterance_bwd_kernel, dim3((unsigned)B), dim3(eb), stream, (const float*)part, (const int*)off_dev, (const float*)dout, dpart, E);
        MTTS_LAUNCH(dvec_head_bwd_kernel, dim3((unsigned)N), dim3(eb), stream, (const float*)eraw, (const float*)dpart,
                    (const float*)(params + find("linear.weight")), dz, dh_last, H, E);
        wgrad(dz, E, hlast, H, N, grads + find("linear.weight"), grads + find("linear.bias"));
        const float* dh_ext = nullptr;
        for (int l = layers - 1; l >= 0; --l) {
            const std::string s = std::to_string(l);
            MTTS_LAUNCH(lstm_bptt_kernel, dim3((unsigned)N), dim3((unsigned)H), stream, (const float*)gates[l], (const float*)cseq[l], dh_ext,
                        (const float*)(l == layers - 1 ? dh_last : nullptr), (const float*)(params + find("lstm.weight_hh_l" + s)), dgates, T, H);
            const float* x = l == 0 ? mels : hkeep[l - 1];
            wgrad(dgates, 4 * H, x, in_dim(l), rows, grads + find("lstm.weight_ih_l" + s), grads + find("lstm.bias_ih_l" + s));
            wgrad(dgates, 4 * H, hprev[l], H, rows, grads + find("lstm.weight_hh_l" + s), nullptr);
            MTTS_LAUNCH(copy_tasks_kernel, dim3(4), dim3(256), stream, (const float*)(grads + find("lstm.bias_ih_l" + s)), (long long)0,
                        grads + find("lstm.bias_hh_l" + s), (long long)0, (long long)H);   // 4H floats = H float4: d b_hh == d b_ih
            if (l > 0) {   // gradient reaching the layer below: dx = dgates W_ih  ([rows][4H] x [4H][H])
                float* dx = dxbuf[l & 1];
                GemmArgs g;
                g.A = dgates; g.lda = 4 * H; g.B = params + find("lstm.weight_ih_l" + s); g.ldb = H; g.C = dx; g.ldc = H;
                g.M = (int)rows; g.N = H; g.K = 4 * H;
                gemm_launch(gx, GEMM_NN, g, (int)rows, H, 1, stream, 0, 2.0 * rows * 4.0 * H * H, 0);
                dh_ext = dx;
            }
        }
        DV_CHECK(hipGetLastError());
        have_forward = false;
        return 0;
    }
    // device scalar: sum of squares of the parameter gradients (a term of the joint clip_grad_norm_, main.py:61)
    const float* grad_sumsq() {
        MTTS_LAUNCH(sumsq_partial_kernel, dim3(64), dim3(256), stream, (const float*)grads, n_params / 4, sq_partial);
        MTTS_LAUNCH(dv_sum_partials_kernel, dim3(1), dim3(64), stream, (const float*)sq_partial, 64, sq_out);
        return sq_out;
    }
    // norm_dev: device scalar holding the JOINT gradient norm (the engine's, with this encoder's term added); Adam as optimizer.py:9-15
    int adam_step(const float* norm_dev, float max_norm, float lr, float b1, float b2, float eps, float weight_decay) {
        if (!train_ready) { set_error("mtts_dvector_enable_training first"); return -1; }
        ++adam_steps;
        const float bc1 = 1.f - (float)std::pow((double)b1, (double)adam_steps), bc2 = 1.f - (float)std::pow((double)b2, (double)adam_steps);
        MTTS_LAUNCH(adam_clip_kernel, dim3(256), dim3(256), stream, params, (const float*)grads, adam_m, adam_v, n_params / 4, norm_dev,
                    norm_dev ? max_norm : 0.f, lr, b1, b2, eps, bc1, bc2, weight_decay);
        DV_CHECK(hipGetLastError());
        dirty = true;
        return 0;
    }
    // which: 0 parameter, 1 gradient, 2 Adam m, 3 Adam v
    float* state_ptr(int which) { return which == 0 ? params : which == 1 ? grads : which == 2 ? adam_m : which == 3 ? adam_v : nullptr; }
    int export_state(const char* name, int which, float* out_host, long long numel) {
        float* base = state_ptr(which);
        if (!base) { set_error("state not available (enable training first)"); return -1; }
        for (auto& t : tensors)
            if (t.name == name) {
                if (t.numel != numel) { set_error(std::string("size mismatch for ") + name); return -1; }
                DV_CHECK(hipStreamSynchronize(stream));
                DV_CHECK(hipMemcpy(out_host, base + t.off, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost));
                return 0;
            }
        set_error(std::string("unknown d-vector tensor ") + name);
        return -1;
    }
    int import_state(const char* name, int which, const float* host, long long numel) {
        if (which == 0) return load(name, host, numel);
        float* base = state_ptr(which);
        if (!base) { set_error("state not available (enable training first)"); return -1; }
        for (auto& t : tensors)
            if (t.name == name) {
                if (t.numel != numel) { set_error(std::string("size mismatch for ") + name); return -1; }
                DV_CHECK(hipMemcpy(base + t.off, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
                return 0;
            }
        set_error(std::string("unknown d-vector tensor ") + name);
        return -1;
    }
};

}  // namespace mtts
