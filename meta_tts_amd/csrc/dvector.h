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

namespace mtts {

__device__ __forceinline__ float dv_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// xp: [N][T][4H] input projections (biases included); whhT: [H][4H]; hseq: [N][T][H] (all hidden states, the next layer's input);
// hlast: [N][H] final hidden state.  blockDim.x == H.
__global__ void lstm_recurrent_kernel(const float* xp, const float* whhT, float* hseq, float* hlast, int T, int H) {
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
        c = dv_sigmoid(gf) * c + dv_sigmoid(gi) * tanhf(gg);
        hv = dv_sigmoid(go) * tanhf(c);
        hout[j] = hv;
        if (hseq) hseq[((long long)n * T + t) * H + j] = hv;
        __syncthreads();
    }
    hlast[(long long)n * H + j] = hv;
}

// e = relu(W h + b), then e / ||e||   (one workgroup per partial utterance, blockDim.x = E rounded up to whole wavefronts; wT: [H][E])
__global__ void dvec_head_kernel(const float* hlast, const float* wT, const float* bias, float* part, int H, int E) {
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
    if (j < E) part[(long long)n * E + j] = e / sqrtf(tot);
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
        gx.numerics = 0;
        if (gx.alloc_workspace() != 0) { set_error("split-K workspace allocation failed"); return -1; }
        return 0;
    }
    void destroy() {
        for (float* p : {params, whhT, linT, bsum, mels, xp, hseq[0], hseq[1], hlast, part, out}) if (p) hipFree(p);
        if (off_dev) hipFree(off_dev);
        gx.release();
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
    // transposed images + summed biases (once after the weights changed; host-side, a few MB)
    int refresh() {
        std::vector<float> P((size_t)n_params);
        DV_CHECK(hipMemcpy(P.data(), params, P.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<float> wt((size_t)layers * 4 * H * H), bs((size_t)layers * 4 * H), lt((size_t)E * H);
        for (int l = 0; l < layers; ++l) {
            const std::string s = std::to_string(l);
            const float* whh = P.data() + find("lstm.weight_hh_l" + s);
            const float* bi = P.data() + find("lstm.bias_ih_l" + s);
            const float* bh = P.data() + find("lstm.bias_hh_l" + s);
            float* o = wt.data() + (size_t)l * 4 * H * H;
            for (int r = 0; r < 4 * H; ++r)
                for (int k = 0; k < H; ++k) o[(size_t)k * 4 * H + r] = whh[(size_t)r * H + k];
            for (int r = 0; r < 4 * H; ++r) bs[(size_t)l * 4 * H + r] = bi[r] + bh[r];
        }
        const float* lw = P.data() + find("linear.weight");
        for (int r = 0; r < E; ++r)
            for (int k = 0; k < H; ++k) lt[(size_t)k * E + r] = lw[(size_t)r * H + k];
        DV_CHECK(hipMemcpy(whhT, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
        DV_CHECK(hipMemcpy(bsum, bs.data(), bs.size() * sizeof(float), hipMemcpyHostToDevice));
        DV_CHECK(hipMemcpy(linT, lt.data(), lt.size() * sizeof(float), hipMemcpyHostToDevice));
        dirty = false;
        return 0;
    }
    // mels_host [N][T][n_mels]; utt_off [B+1] partial offsets (utt_off[0] = 0, utt_off[B] = N); out_host [B][E]; part_host [N][E] or null
    int embed(const float* mels_host, int N, const int* utt_off, int B, float* out_host, float* part_host) {
        if (!mels_host || !utt_off || !out_host || N < 1 || N > cap_N || B < 1 || B > cap_B) { set_error("bad d-vector arguments"); return -1; }
        if (utt_off[0] != 0 || utt_off[B] != N) { set_error("utterance offsets must cover [0, N)"); return -1; }
        for (int b = 0; b < B; ++b) if (utt_off[b + 1] < utt_off[b]) { set_error("utterance offsets must not decrease"); return -1; }
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
            float* hs = (l + 1 < layers) ? hseq[l & 1] : nullptr;
            MTTS_LAUNCH(lstm_recurrent_kernel, dim3((unsigned)N), dim3((unsigned)H), stream, (const float*)xp,
                        (const float*)(whhT + (long long)l * 4 * H * H), hs, hlast, T, H);
            x = hs;
        }
        MTTS_LAUNCH(dvec_head_kernel, dim3((unsigned)N), dim3((unsigned)((E + 63) & ~63)), stream, (const float*)hlast, (const float*)linT,
                    (const float*)(params + find("linear.bias")), part, H, E);
        MTTS_LAUNCH(dvec_utterance_kernel, dim3((unsigned)B), dim3((unsigned)((E + 63) & ~63)), stream, (const float*)part, (const int*)off_dev, out, E);
        DV_CHECK(hipGetLastError());
        DV_CHECK(hipMemcpyAsync(out_host, out, (size_t)B * E * sizeof(float), hipMemcpyDeviceToHost, stream));
        if (part_host) DV_CHECK(hipMemcpyAsync(part_host, part, (size_t)N * E * sizeof(float), hipMemcpyDeviceToHost, stream));
        DV_CHECK(hipStreamSynchronize(stream));
        return 0;
    }
};

}  // namespace mtts
