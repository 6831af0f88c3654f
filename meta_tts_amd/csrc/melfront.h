// Waveform -> log-mel spectrogram + frame energy on the device: the feature front-end of the preprocessing stage.
//
// Reference: audio/stft.py:15-77 (STFT.transform: reflect padding by n_fft/2, a strided Conv1d against the windowed real / imaginary
// Fourier basis, magnitude) and :128-178 (TacotronSTFT.mel_spectrogram: mel_basis @ magnitude, log(clamp(., 1e-5)), energy = L2 norm
// of the magnitudes over frequency), called by audio/tools.py:8-15 (get_mel_from_wav: clip to [-1, 1] first) from the preprocessor.
//
// MI355X layout: the strided Conv1d is an implicit GEMM over OVERLAPPING rows of the padded waveform — frame t is the contiguous
// span x[t*hop .. t*hop + n_fft), i.e. an A operand with lda = hop < K = n_fft, no framing copy — against the [2*(n_fft/2+1)][n_fft]
// basis on the fp32 matrix cores (gemm.h); magnitude and energy are one wavefront-per-frame pass; the mel projection is a second
// GEMM whose epilogue-side log/clamp runs as a small row kernel.  Output rows are [frame][n_mel] (the engine's mel layout).
#pragma once
#include <string>
#include <vector>

#include "gemm.h"
#include "rowops.h"

namespace mtts {

// xp[j] = clip(x[reflect(j - pad)], -1, 1) for j < n + 2 * pad   (F.pad(..., mode="reflect") of stft.py:60-65 after tools.py:9)
__global__ void wav_reflect_pad_kernel(const float* x, int n, int pad, float* xp) {
    const long long total = (long long)n + 2 * pad;
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < total; j += (long long)gridDim.x * blockDim.x) {
        long long s = j - pad;
        if (s < 0) s = -s;
        else if (s >= n) s = 2LL * (n - 1) - s;
        float v = x[s];
        v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
        xp[j] = v;
    }
}

// spec: [T][ld_spec] = [re(0..F) | im(0..F)] per frame  ->  mag [T][ld_mag] (columns >= F zeroed), energy[t] = ||mag[t]||_2
__global__ void stft_magnitude_kernel(const float* spec, int ld_spec, int T, int F, float* mag, int ld_mag, float* energy) {
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= T) return;
    const float* p = spec + (long long)row * ld_spec;
    float* m = mag + (long long)row * ld_mag;
    float ss = 0.f;
    for (int f = lane; f < ld_mag; f += 64) {
        float v = 0.f;
        if (f < F) {
            const float re = p[f], im = p[F + f];
            const float sq = re * re + im * im;
            v = sqrtf(sq);
            ss += sq;
        }
        m[f] = v;
    }
    ss = wave_sum(ss);
    if (lane == 0) energy[row] = sqrtf(ss);
}

// in place: x = log(max(x, clip))   (audio_processing.py:85-91, C = 1)
__global__ void log_clamp_kernel(float* x, long long n, float clip) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        x[i] = logf(v > clip ? v : clip);
    }
}

class MelFront {
public:
    int n_fft = 1024, hop = 256, n_mel = 80, F = 513, cap_samples = 0, cap_T = 0;
    int ld_spec = 0, ld_mag = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    GemmCtx gx;
    float *basis = nullptr, *melb = nullptr;   // [2F][n_fft] windowed Fourier basis; [n_mel][ld_mag] mel filter bank (zero padded)
    float *wav = nullptr, *wavp = nullptr, *spec = nullptr, *mag = nullptr, *mel = nullptr, *energy = nullptr;
    bool have_basis = false, have_mel = false;

    void set_error(const std::string& s) { last_error = s; }
#define MF_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return -1; } } while (0)

    int init(int filter_length, int hop_length, int n_mel_channels, int max_samples) {
        n_fft = filter_length; hop = hop_length; n_mel = n_mel_channels; cap_samples = max_samples;
        if (n_fft < 16 || (n_fft & 3) || hop < 4 || (hop & 3) || hop > n_fft || n_mel < 1 || n_mel > 1024 || max_samples <= n_fft / 2) {
            set_error("unsupported STFT configuration (filter_length % 4, hop_length % 4 <= filter_length, max_samples > filter_length / 2)");
            return -1;
        }
        F = n_fft / 2 + 1;
        ld_spec = (2 * F + 3) & ~3;
        ld_mag = (F + 3) & ~3;
        cap_T = max_samples / hop + 1;
        MF_CHECK(hipMalloc((void**)&basis, (size_t)2 * F * n_fft * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&melb, (size_t)n_mel * ld_mag * sizeof(float)));
        MF_CHECK(hipMemset(melb, 0, (size_t)n_mel * ld_mag * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&wav, (size_t)max_samples * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&wavp, ((size_t)max_samples + n_fft + 64) * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&spec, ((size_t)cap_T * ld_spec + 64) * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&mag, ((size_t)cap_T * ld_mag + 64) * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&mel, ((size_t)cap_T * n_mel + 64) * sizeof(float)));
        MF_CHECK(hipMalloc((void**)&energy, (size_t)cap_T * sizeof(float)));
        if (gx.alloc_workspace()) { set_error("hipMalloc failed (split-K workspace)"); return -1; }
        return 0;
    }
    void destroy() {
        for (float* p : {basis, melb, wav, wavp, spec, mag, mel, energy}) if (p) hipFree(p);
        gx.release();
    }
    // forward_basis: [2F][n_fft] (stft.py:27-46, window applied); mel_basis: [n_mel][F] (stft.py:143-147)
    int load(const float* forward_basis, const float* mel_basis) {
        if (forward_basis) { MF_CHECK(hipMemcpy(basis, forward_basis, (size_t)2 * F * n_fft * sizeof(float), hipMemcpyHostToDevice)); have_basis = true; }
        if (mel_basis) {
            std::vector<float> padded((size_t)n_mel * ld_mag, 0.f);
            for (int m = 0; m < n_mel; ++m)
                for (int f = 0; f < F; ++f) padded[(size_t)m * ld_mag + f] = mel_basis[(size_t)m * F + f];
            MF_CHECK(hipMemcpy(melb, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
            have_mel = true;
        }
        return 0;
    }
    int frames_of(int n_samples) const { return n_samples / hop + 1; }   // conv1d over the padded signal: (n + n_fft - n_fft) / hop + 1
    // wav_host [n_samples] -> mel_host [T][n_mel] (log-mel), energy_host [T]; returns T, < 0 on error
    int mel_spectrogram(const float* wav_host, int n_samples, float* mel_host, float* energy_host) {
        if (!have_basis || !have_mel) { set_error("STFT bases not loaded"); return -1; }
        if (!wav_host || !mel_host || !energy_host || n_samples <= n_fft / 2 || n_samples > cap_samples) {
            set_error("bad waveform length (need filter_length / 2 < n_samples <= max_samples: reflection padding reads n_fft / 2 samples)");
            return -1;
        }
        const int T = frames_of(n_samples);
        MF_CHECK(hipMemcpyAsync(wav, wav_host, (size_t)n_samples * sizeof(float), hipMemcpyHostToDevice, stream));
        MTTS_LAUNCH(wav_reflect_pad_kernel, dim3(1024), dim3(256), stream, (const float*)wav, n_samples, n_fft / 2, wavp);
        {   // frames x basis: C[T][2F] = A[T][n_fft] (rows overlap: lda = hop) * basis[2F][n_fft]^T
            GemmArgs g;
            g.A = wavp; g.lda = hop; g.B = basis; g.ldb = n_fft; g.C = spec; g.ldc = ld_spec;
            g.M = T; g.N = 2 * F; g.K = n_fft;
            gemm_launch(gx, GEMM_NT, g, T, 2 * F, 1, stream, 0, 2.0 * T * 2.0 * F * n_fft, 0);
        }
        MTTS_LAUNCH(stft_magnitude_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), stream, (const float*)spec, ld_spec, T, F, mag, ld_mag, energy);
        {   // mel[T][n_mel] = mag[T][F] * mel_basis[n_mel][F]^T   (both zero padded to ld_mag columns)
            GemmArgs g;
            g.A = mag; g.lda = ld_mag; g.B = melb; g.ldb = ld_mag; g.C = mel; g.ldc = n_mel;
            g.M = T; g.N = n_mel; g.K = ld_mag;
            gemm_launch(gx, GEMM_NT, g, T, n_mel, 1, stream, 0, 2.0 * T * (double)n_mel * F, 0);
        }
        MTTS_LAUNCH(log_clamp_kernel, dim3(256), dim3(256), stream, mel, (long long)T * n_mel, 1e-5f);
        MF_CHECK(hipGetLastError());
        MF_CHECK(hipMemcpyAsync(mel_host, mel, (size_t)T * n_mel * sizeof(float), hipMemcpyDeviceToHost, stream));
        MF_CHECK(hipMemcpyAsync(energy_host, energy, (size_t)T * sizeof(float), hipMemcpyDeviceToHost, stream));
        MF_CHECK(hipStreamSynchronize(stream));
        return T;
    }
};

}  // namespace mtts
