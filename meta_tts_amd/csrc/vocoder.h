// MelGAN generator (mel -> waveform) on the grouped fp32-MFMA GEMM — SURVEY.md section 8 row a23.
//
// Reference call site: lightning/utils.py:8-30 (LightningMelGAN: torch.hub "descriptinc/melgan-neurips", `mel2wav`
// on mel / ln 10, x max_wav_value -> int16, cropped to mel_len * hop).  The generator itself is NOT in the reference
// tree (un-vendored torch.hub dependency, no pin): the architecture below is the published one of that hub entry
// (Kumar et al. 2019, Generator(input 80, ngf 32, n_residual_layers 3), ratios 8,8,2,2):
//     ReflectionPad(3) Conv1d(80 -> 512, k7)
//     4 x [ LeakyReLU(0.2) ConvTranspose1d(C -> C/2, k = 2r, stride r, pad r/2 + r%2, out_pad r%2)
//           3 x ResnetBlock(C/2, dilation 1, 3, 9):  x -> shortcut Conv1d k1 (x)
//                                                      + Conv1d k1( LeakyReLU( Conv1d k3 dil d( ReflectionPad(d) LeakyReLU(x) ) ) ) ]
//     LeakyReLU(0.2) ReflectionPad(3) Conv1d(32 -> 1, k7) Tanh
// with weight normalisation folded on the host (meta_tts_amd/vocoder.py).  Parity is therefore "unpinned": checked
// against oracle/melgan_oracle.py, a torch restatement of the same published architecture.
//
// MI355X mapping: activations are channels-last row matrices [time][C] per utterance (group = utterance, per-group row
// count through GemmArgs::dimptr), so
//   * Conv1d k (dilation 1) is the implicit GEMM over overlapping rows of engine.h (K = k*C_in, A rows contiguous);
//   * a dilated k=3 conv is ONE implicit GEMM too: the K-loop walks the taps and shifts the A row by the dilation
//     (GemmArgs::a_tap_k / a_tap_rows; channel counts that are not a multiple of 32 fall back to three accumulate passes);
//   * ConvTranspose1d (k = 2r, stride r) is r polyphase GEMMs: output o = q*r + ph - p reads exactly x[q-1] and x[q]
//     -> A row = the contiguous pair [x[q-1] | x[q]] (K = 2*C_in), B = the phase's [C_out][2*C_in] weight image, C rows
//     interleaved through ldc = r*C_out; the r phases are independent and go out as multi-problem launches;
//   * LeakyReLU + reflection / zero padding are one pass (pad_act_kernel) that writes the padded operand of the next
//     conv; the LeakyReLU between the two convs of a block rides in the GEMM epilogue;
//   * the 32 -> 1 output conv + tanh is a wavefront-per-sample dot product (224 contiguous floats of the padded rows).
#pragma once
#include <string>
#include <vector>

#include "gemm.h"
#include "gemm_glds.h"

namespace mtts {

// y rows [0, T + 2d): y[j] = act(x[src(j - d)]) * scale;  reflect: src(s) = -s (s < 0), 2(T-1) - s (s >= T); else zero rows.
// T = lens[z] * len_mult.  One wavefront per output row.
__global__ void pad_act_kernel(const int* lens, int len_mult, const float* x, long long x_gs, float* y, long long y_gs,
                               int C, int d, float slope, float scale, int reflect) {
    const int z = blockIdx.z;
    const int T = lens[z] * len_mult;
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= T + 2 * d) return;
    int s = row - d;
    bool zero = false;
    if (s < 0) { if (reflect) s = -s; else zero = true; }
    else if (s >= T) { if (reflect) s = 2 * (T - 1) - s; else zero = true; }
    if (s < 0) s = 0;
    if (s >= T) s = T - 1;
    const float* px = x + (long long)z * x_gs + (long long)s * C;
    float* py = y + (long long)z * y_gs + (long long)row * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 v = zero4();
        if (!zero) {
            v = ld4(px + c);
            v.x = (v.x > 0.f ? v.x : slope * v.x) * scale; v.y = (v.y > 0.f ? v.y : slope * v.y) * scale;
            v.z = (v.z > 0.f ? v.z : slope * v.z) * scale; v.w = (v.w > 0.f ? v.w : slope * v.w) * scale;
        }
        st4(py + c, v);
    }
}

// wav[t] = tanh(b + <w[0 .. K), xp[t*C .. t*C + K)>)   (k=7 conv to one channel over the padded rows, K = 7*C)
__global__ void dot_tanh_kernel(const int* lens, int len_mult, const float* xp, long long x_gs, const float* w, const float* b,
                                float* out, long long out_gs, int C, int K) {
    const int z = blockIdx.z;
    const int T = lens[z] * len_mult;
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= T) return;
    const float* px = xp + (long long)z * x_gs + (long long)row * C;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 a = ld4(px + k), ww = ld4(w + k);
        s += (a.x * ww.x + a.y * ww.y) + (a.z * ww.z + a.w * ww.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[(long long)z * out_gs + row] = tanhf(s + b[0]);
}

struct VocoderCfg {
    int n_mel = 80, ngf = 32, n_res = 3, n_ratios = 4;
    int ratios[8] = {8, 8, 2, 2, 0, 0, 0, 0};
};

class Vocoder {
public:
    VocoderCfg cfg;
    int cap_B = 0, cap_T = 0;
    hipStream_t stream = nullptr;
    std::string last_error;

    struct Tensor { std::string name; long long off, numel; };
    std::vector<Tensor> tensors;
    float* params = nullptr;
    long long n_params = 0;
    float* arena = nullptr;
    int* lens_dev = nullptr;
    float* mel_dev = nullptr;
    float* wav_dev = nullptr;
    int hop = 1;
    // per-stage scratch (element offsets into the arena, per-utterance strides)
    struct Buf { long long off, gs; };
    GemmCtx gx;  // this handle's launcher state (gemm.h)
    Buf b_x[2], b_pad, b_h;

    void set_error(const std::string& s) { last_error = s; }
#define VOC_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return -1; } } while (0)

    int ch(int stage) const { int c = cfg.ngf << cfg.n_ratios; for (int s = 0; s < stage; ++s) c >>= 1; return c; }  // channels after `stage` upsamples
    long long find(const std::string& n) const { for (auto& t : tensors) if (t.name == n) return t.off; return -1; }

    int init(const VocoderCfg& c, int max_B, int max_T) {
        cfg = c; cap_B = max_B; cap_T = max_T;
        if (cfg.n_mel % 16 || cfg.ngf % 16 || cfg.n_ratios < 1 || cfg.n_ratios > 8 || max_B < 1 || max_T < 4) { set_error("unsupported vocoder configuration"); return -1; }
        hop = 1;
        for (int s = 0; s < cfg.n_ratios; ++s) { if (cfg.ratios[s] < 2 || (cfg.ratios[s] & 1)) { set_error("ratios must be even"); return -1; } hop *= cfg.ratios[s]; }
        auto add = [&](const std::string& n, long long numel) { tensors.push_back(Tensor{n, n_params, numel}); n_params += (numel + 3) & ~3LL; };
        const int c0 = ch(0);
        add("conv_in.w", (long long)c0 * 7 * cfg.n_mel); add("conv_in.b", c0);
        for (int s = 0; s < cfg.n_ratios; ++s) {
            const int cin = ch(s), cout = ch(s + 1), r = cfg.ratios[s];
            add("up" + std::to_string(s) + ".w", (long long)r * cout * 2 * cin);  // [phase][C_out][x[q-1] | x[q]]
            add("up" + std::to_string(s) + ".b", cout);
            for (int j = 0; j < cfg.n_res; ++j) {
                const std::string p = "res" + std::to_string(s) + "." + std::to_string(j);
                add(p + ".w1", (long long)cout * 3 * cout); add(p + ".b1", cout);   // [C][3][C] dilated k=3
                add(p + ".w2", (long long)cout * cout); add(p + ".b2", cout);       // k=1
                add(p + ".ws", (long long)cout * cout); add(p + ".bs", cout);       // shortcut k=1
            }
        }
        add("conv_out.w", 7LL * ch(cfg.n_ratios)); add("conv_out.b", 1);
        if (gx.alloc_workspace()) { set_error("hipMalloc failed (split-K workspace)"); return -1; }
        VOC_CHECK(hipMalloc((void**)&params, (size_t)n_params * sizeof(float)));
        VOC_CHECK(hipMemset(params, 0, (size_t)n_params * sizeof(float)));
        // activations: rows x channels is largest where rows*C peaks; size every buffer for the worst stage (+ pad rows)
        long long worst = (long long)(cap_T + 32) * std::max(cfg.n_mel, c0);
        long long mult = 1;
        int maxd = 1;
        for (int j = 1; j < cfg.n_res; ++j) maxd *= 3;
        for (int s = 0; s < cfg.n_ratios; ++s) { mult *= cfg.ratios[s]; worst = std::max(worst, ((long long)cap_T * mult + 2 * maxd + 32) * ch(s + 1)); }
        worst = (worst + 63) & ~63LL;
        const long long per = worst;  // per utterance per buffer
        VOC_CHECK(hipMalloc((void**)&arena, (size_t)per * cap_B * 4 * sizeof(float)));
        VOC_CHECK(hipMemset(arena, 0, (size_t)per * cap_B * 4 * sizeof(float)));
        b_x[0] = Buf{0, per}; b_x[1] = Buf{per * cap_B, per}; b_pad = Buf{2 * per * cap_B, per}; b_h = Buf{3 * per * cap_B, per};
        VOC_CHECK(hipMalloc((void**)&lens_dev, cap_B * sizeof(int)));
        VOC_CHECK(hipMalloc((void**)&mel_dev, (size_t)cap_B * cap_T * cfg.n_mel * sizeof(float)));
        VOC_CHECK(hipMalloc((void**)&wav_dev, (size_t)cap_B * cap_T * hop * sizeof(float)));
        return 0;
    }
    void destroy() {
        for (void* p : {(void*)params, (void*)arena, (void*)lens_dev, (void*)mel_dev, (void*)wav_dev}) if (p) hipFree(p);
        params = arena = mel_dev = wav_dev = nullptr; lens_dev = nullptr;
        gx.release();
    }
    int load(const char* name, const float* host, long long numel) {
        for (auto& t : tensors)
            if (t.name == name) {
                if (t.numel != numel) { set_error(std::string("size mismatch for ") + name); return -1; }
                VOC_CHECK(hipMemcpy(params + t.off, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
                return 0;
            }
        set_error(std::string("unknown vocoder tensor ") + name);
        return -1;
    }

    // one grouped GEMM: C[z][M_z, N] (+)= A[z][M_z, K] * W[N, K]^T + bias; M_z = lens[z] * len_mult
    void gemm(const float* A, long long a_gs, int lda, const float* W, int ldb, int K, const float* bias, float* C, long long c_gs,
              int ldc, int N, int B, int max_rows, int len_mult, int flags, int a_tap_k = 0, int a_tap_rows = 0) {
        GemmArgs g;
        if (a_tap_k > 0) { g.a_tap_k = a_tap_k; g.a_tap_rows = a_tap_rows; }
        g.A = A; g.a_gs = a_gs; g.lda = lda;
        g.B = W; g.b_gs = 0; g.ldb = ldb;
        g.C = C; g.c_gs = c_gs; g.ldc = ldc;
        g.M = max_rows; g.N = N; g.K = K;
        g.dimptr = lens_dev; g.dim_stride = 1; g.dim_sel = 0; g.dim_mult = len_mult;
        g.bias = bias; g.flags = flags; g.act_slope = 0.2f;
        gemm_launch(gx, GEMM_NT, g, max_rows, N, B, stream, 0, 2.0 * max_rows * B * (double)N * K, 0);
    }
    void pad_act(const Buf& src, const Buf& dst, int B, int max_rows, int len_mult, int C, int d, float slope, float scale, int reflect) {
        MTTS_LAUNCH(pad_act_kernel, dim3((unsigned)((max_rows + 2 * d + 3) / 4), 1, (unsigned)B), dim3(256), stream, (const int*)lens_dev,
                    len_mult, (const float*)(arena + src.off), src.gs, arena + dst.off, dst.gs, C, d, slope, scale, reflect);
    }

    // mel: device pointer [B][T_max][n_mel] (row-major, already scaled as the caller wants); wav: device [B][T_max * hop]
    // mel_gs: floats between consecutive utterances of `mel` (0: T_max * n_mel, i.e. a dense [B][T_max][n_mel] array)
    int run(const float* mel, int B, int T_max, const int* lens_host, float mel_scale, float* wav, long long wav_gs, long long mel_gs = 0) {
        if (B < 1 || B > cap_B || T_max < 4 || T_max > cap_T) { set_error("vocoder batch exceeds capacity"); return -1; }
        for (int b = 0; b < B; ++b) if (lens_host[b] < 4 || lens_host[b] > T_max) { set_error("mel length out of range (need 4 <= len <= T_max)"); return -1; }
        VOC_CHECK(hipMemcpyAsync(lens_dev, lens_host, B * sizeof(int), hipMemcpyHostToDevice, stream));
        const int nm = cfg.n_mel;
        // conv_in: reflection pad 3 of the scaled mel, k=7 conv
        {
            MTTS_LAUNCH(pad_act_kernel, dim3((unsigned)((T_max + 6 + 3) / 4), 1, (unsigned)B), dim3(256), stream, (const int*)lens_dev, 1, mel,
                        mel_gs > 0 ? mel_gs : (long long)T_max * nm, arena + b_pad.off, b_pad.gs, nm, 3, 1.f, mel_scale, 1);
            gemm(arena + b_pad.off, b_pad.gs, nm, params + find("conv_in.w"), 7 * nm, 7 * nm, params + find("conv_in.b"), arena + b_x[0].off,
                 b_x[0].gs, ch(0), ch(0), B, T_max, 1, 0);
        }
        int cur = 0, mult = 1;
        for (int s = 0; s < cfg.n_ratios; ++s) {
            const int cin = ch(s), cout = ch(s + 1), r = cfg.ratios[s], p = r / 2 + (r & 1);
            const int rows_in = T_max * mult;
            // LeakyReLU + one zero row each side, then r polyphase GEMMs
            pad_act(b_x[cur], b_pad, B, rows_in, mult, cin, 1, 0.2f, 1.f, 0);
            const float* W = params + find("up" + std::to_string(s) + ".w");
            const float* bias = params + find("up" + std::to_string(s) + ".b");
            float* out = arena + b_x[cur ^ 1].off;
            const long long out_gs = b_x[cur ^ 1].gs;
            gemm_batch_begin(gx);
            for (int ph = 0; ph < r; ++ph) {
                const int q0 = ph >= p ? 0 : 1;
                gemm(arena + b_pad.off + (long long)q0 * cin, b_pad.gs, cin, W + (long long)ph * cout * 2 * cin, 2 * cin, 2 * cin, bias,
                     out + (long long)(q0 * r + ph - p) * cout, out_gs, r * cout, cout, B, rows_in, mult, 0);
            }
            gemm_batch_end(gx, stream);
            cur ^= 1; mult *= r;
            const int rows = T_max * mult;
            int d = 1;
            for (int j = 0; j < cfg.n_res; ++j, d *= 3) {
                const std::string pre = "res" + std::to_string(s) + "." + std::to_string(j);
                const float* w1 = params + find(pre + ".w1");
                // xp = reflect-pad(LeakyReLU(x), d);  h = LeakyReLU(conv k3 dil d (xp))
                pad_act(b_x[cur], b_pad, B, rows, mult, cout, d, 0.2f, 1.f, 1);
                float* h = arena + b_h.off;
                if (d == 1) {
                    gemm(arena + b_pad.off, b_pad.gs, cout, w1, 3 * cout, 3 * cout, params + find(pre + ".b1"), h, b_h.gs, cout, cout, B, rows, mult, GEMM_LRELU);
                } else if (cout % 32 == 0) {
                    // dilated taps inside the K-loop: k = t * C + c reads row m + t * d
                    gemm(arena + b_pad.off, b_pad.gs, cout, w1, 3 * cout, 3 * cout, params + find(pre + ".b1"), h, b_h.gs, cout, cout, B, rows, mult,
                         GEMM_LRELU, cout, d);
                } else {
                    for (int t = 0; t < 3; ++t)
                        gemm(arena + b_pad.off + (long long)t * d * cout, b_pad.gs, cout, w1 + (long long)t * cout, 3 * cout, cout,
                             t == 0 ? params + find(pre + ".b1") : nullptr, h, b_h.gs, cout, cout, B, rows, mult,
                             (t ? GEMM_ACCUM : 0) | (t == 2 ? GEMM_LRELU : 0));
                }
                // out = shortcut(x) + conv k1 (h)
                float* o = arena + b_x[cur ^ 1].off;
                gemm(arena + b_x[cur].off, b_x[cur].gs, cout, params + find(pre + ".ws"), cout, cout, params + find(pre + ".bs"), o, b_x[cur ^ 1].gs, cout,
                     cout, B, rows, mult, 0);
                gemm(h, b_h.gs, cout, params + find(pre + ".w2"), cout, cout, params + find(pre + ".b2"), o, b_x[cur ^ 1].gs, cout, cout, B, rows, mult,
                     GEMM_ACCUM);
                cur ^= 1;
            }
        }
        // LeakyReLU, reflection pad 3, k=7 conv to one channel, tanh
        const int cl = ch(cfg.n_ratios), rows = T_max * mult;
        pad_act(b_x[cur], b_pad, B, rows, mult, cl, 3, 0.2f, 1.f, 1);
        MTTS_LAUNCH(dot_tanh_kernel, dim3((unsigned)((rows + 3) / 4), 1, (unsigned)B), dim3(256), stream, (const int*)lens_dev, mult,
                    (const float*)(arena + b_pad.off), b_pad.gs, (const float*)(params + find("conv_out.w")), (const float*)(params + find("conv_out.b")),
                    wav, wav_gs, cl, 7 * cl);
        return 0;
    }

    // host entry: mel_host [B][T_max][n_mel] -> wav_host [B][T_max * hop] (samples beyond len * hop are left untouched)
    int infer_host(const float* mel_host, int B, int T_max, const int* lens_host, float mel_scale, float* wav_host) {
        if (B < 1 || B > cap_B || T_max < 4 || T_max > cap_T) { set_error("vocoder batch exceeds capacity"); return -1; }
        VOC_CHECK(hipMemcpyAsync(mel_dev, mel_host, (size_t)B * T_max * cfg.n_mel * sizeof(float), hipMemcpyHostToDevice, stream));
        if (run(mel_dev, B, T_max, lens_host, mel_scale, wav_dev, (long long)T_max * hop) != 0) return -1;
        VOC_CHECK(hipStreamSynchronize(stream));
        for (int b = 0; b < B; ++b)
            VOC_CHECK(hipMemcpy(wav_host + (long long)b * T_max * hop, wav_dev + (long long)b * T_max * hop, (size_t)lens_host[b] * hop * sizeof(float),
                                hipMemcpyDeviceToHost));
        return 0;
    }
#undef VOC_CHECK
};

}  // namespace mtts
