// Batch ingestion: the row-space index arrays of a batch plan (engine.h: P / F / R spaces) are built ON THE DEVICE from a
// compact image of the batch — lengths, phoneme ids, durations, pitch / energy targets, speaker ids and the raw padded mels —
// that the host packs into one pinned staging buffer and ships with ONE asynchronous copy per slot.  Replaces what the
// reference does per batch on the host (PL batch transfer of the 12-tuple, lightning/collate.py:47-60, then
// get_mask_from_lengths utils/tools.py:91-99, LengthRegulator.expand's per-phoneme .item() loop modules.py:180-187 and the
// masked_select compactions of loss.py:54-92): masks, the length-regulator gather map and the packed-frame maps all come out of
// four small kernels; the only host work is O(B) scalars per task and the attention descriptor tables.
#pragma once
#include "rowops.h"

namespace mtts {

struct PlanTaskHdr {  // one per task at the head of the compact image
    int B, S, Tmax_in, Tcap, Mp, Mf, Mr, with_frames;
    int has_targets, has_mels, pad0, pad1;
    long long mel_off;  // float offset of this task's [B][Tmax_in][n_mel] mel block inside the image's mel area
};

struct PlanImage {  // device view of one slot's compact image (fixed capacity strides)
    const PlanTaskHdr* hdr;
    const int* src_len;   // [task][cap_B]
    const int* flen;      // [task][cap_B]   frames kept per utterance (min(mel_len, Tcap))
    const int* foff;      // [task][cap_B]   first packed-frame row of the utterance
    const int* spk;       // [task][cap_B + 1]
    const int* texts;     // [task][cap_B * cap_S]   (row-major [B][S] of THIS task, S = hdr.S)
    const int* dur;       // [task][cap_B * cap_S]
    const float* pitch;   // [task][pe_cap_p]: [B][S] phoneme-level or [B][Tmax_in] frame-level targets
    const float* energy;  // [task][pe_cap_e]
    const float* mels;    // mel area
    int cap_B, cap_S;
    long long pe_cap_p, pe_cap_e;      // per-task capacity of the pitch / energy areas (floats)
    int pitch_frame, energy_frame;     // feature levels (preprocess config)
};

struct PlanOut {  // the plan arrays the kernels of rowops.h / gemm.h consume (per-task strides in elements)
    int *p_row_b, *p_row_t, *p_tok, *p_first, *p_count, *p_dur, *p_seg_start, *p_seg_len;
    unsigned char *p_valid, *p_inrect;
    float *p_pitch_t, *p_energy_t;
    int *f_row_b, *f_row_t, *f_src, *f_seg_start, *f_seg_len, *f2r;
    unsigned char* f_valid;
    int* r2f;
    unsigned char *r_valid, *r_inrect;
    // the same masks as [row][4] floats (1.0 / 0.0): the B operand of the column-sum tile of a weight-gradient GEMM (gemm.h: GemmArgs::colsum_w)
    float *p_valid_w, *p_inrect_w, *f_valid_w, *r_valid_w, *r_inrect_w;
    float *r_pitch_t, *r_energy_t;   // frame-level targets on the mel rows (null when the feature is phoneme-level)
    float* mel_tgt;
    int* spk_ids;
    long long ts_p, ts_f, ts_r, ts_mel, ts_seg, ts_spk;
};

constexpr int kPlanG = 4;  // == engine.h G (guard rows)

// phoneme rectangle rows + per-utterance segment tables + speaker ids
__global__ void plan_rows_p_kernel(PlanImage im, PlanOut o) {
    const int z = blockIdx.z;
    const PlanTaskHdr h = im.hdr[z];
    const int bs = im.cap_B * im.cap_S;
    if (blockIdx.x == 0) {
        for (int i = (int)threadIdx.x; i < im.cap_B + 1; i += (int)blockDim.x) o.spk_ids[(long long)z * o.ts_spk + i] = im.spk[(long long)z * (im.cap_B + 1) + i];
        for (int i = (int)threadIdx.x; i < h.B; i += (int)blockDim.x) {
            o.p_seg_start[(long long)z * o.ts_seg + i] = kPlanG + i * (h.S + kPlanG);
            o.p_seg_len[(long long)z * o.ts_seg + i] = h.S;
            o.f_seg_start[(long long)z * o.ts_seg + i] = h.with_frames ? im.foff[(long long)z * im.cap_B + i] : 0;
            o.f_seg_len[(long long)z * o.ts_seg + i] = h.with_frames ? im.flen[(long long)z * im.cap_B + i] : 0;
        }
    }
    for (int r = blockIdx.x * (int)blockDim.x + (int)threadIdx.x; r < h.Mp; r += (int)(gridDim.x * blockDim.x)) {
        const long long q = (long long)z * o.ts_p + r;
        int rb = 0, rt = -1, tok = 0;
        unsigned char valid = 0, inrect = 0;
        float pt = 0.f, et = 0.f;
        if (r >= kPlanG) {
            const int i = (r - kPlanG) / (h.S + kPlanG), s = (r - kPlanG) - i * (h.S + kPlanG);
            if (i < h.B && s < h.S) {
                rb = i; rt = s; inrect = 1;
                valid = s < im.src_len[(long long)z * im.cap_B + i];
                const long long e = (long long)z * bs + (long long)i * h.S + s;
                tok = valid ? im.texts[e] : 0;
                const long long ei = (long long)i * h.S + s;
                if (h.has_targets && !im.pitch_frame) pt = im.pitch[(long long)z * im.pe_cap_p + ei];
                if (h.has_targets && !im.energy_frame) et = im.energy[(long long)z * im.pe_cap_e + ei];
            }
        }
        o.p_row_b[q] = rb; o.p_row_t[q] = rt; o.p_tok[q] = tok; o.p_valid[q] = valid; o.p_inrect[q] = inrect;
        { const float v = valid ? 1.f : 0.f, w = inrect ? 1.f : 0.f; st4(o.p_valid_w + 4 * q, make_float4(v, v, v, v)); st4(o.p_inrect_w + 4 * q, make_float4(w, w, w, w)); }
        o.p_pitch_t[q] = pt; o.p_energy_t[q] = et;
        if (!inrect || !h.with_frames) { o.p_first[q] = 0; o.p_count[q] = 0; o.p_dur[q] = 0; }
    }
}

// packed frame rows: utterance / position of every row, the F -> R map; f_src starts at -1 (plan_prefix_kernel scatters it)
__global__ void plan_rows_f_kernel(PlanImage im, PlanOut o) {
    const int z = blockIdx.z;
    const PlanTaskHdr h = im.hdr[z];
    if (!h.with_frames) return;
    for (int r = blockIdx.x * (int)blockDim.x + (int)threadIdx.x; r < h.Mf; r += (int)(gridDim.x * blockDim.x)) {
        const long long q = (long long)z * o.ts_f + r;
        int rb = 0, rt = -1, f2r = -1;
        unsigned char valid = 0;
        for (int i = 0; i < h.B; ++i) {
            const int f0 = im.foff[(long long)z * im.cap_B + i], n = im.flen[(long long)z * im.cap_B + i];
            if (r >= f0 && r < f0 + n) { rb = i; rt = r - f0; valid = 1; f2r = kPlanG + i * (h.Tcap + kPlanG) + rt; break; }
        }
        o.f_row_b[q] = rb; o.f_row_t[q] = rt; o.f_valid[q] = valid; o.f2r[q] = f2r; o.f_src[q] = -1;
        { const float v = valid ? 1.f : 0.f; st4(o.f_valid_w + 4 * q, make_float4(v, v, v, v)); }
    }
}

// one thread per utterance: exclusive prefix of the (clamped) durations -> first frame / frame count of every phoneme inside the
// kept window, and the frame -> phoneme gather map of the length regulator (modules.py:167-190)
__global__ void plan_prefix_kernel(PlanImage im, PlanOut o) {
    const int z = blockIdx.z, i = (int)threadIdx.x;
    const PlanTaskHdr h = im.hdr[z];
    if (!h.with_frames || i >= h.B) return;
    const int bs = im.cap_B * im.cap_S;
    const int f0 = im.foff[(long long)z * im.cap_B + i], n = im.flen[(long long)z * im.cap_B + i];
    int cum = 0;
    for (int s = 0; s < h.S; ++s) {
        const int r = kPlanG + i * (h.S + kPlanG) + s;
        const long long q = (long long)z * o.ts_p + r;
        int dd = im.dur[(long long)z * bs + (long long)i * h.S + s];
        if (dd < 0) dd = 0;
        const int lo = cum < n ? cum : n;
        long long hi64 = (long long)cum + dd;
        const int hi = hi64 < n ? (int)hi64 : n;
        o.p_dur[q] = dd; o.p_first[q] = f0 + lo; o.p_count[q] = hi - lo;
        for (int f = lo; f < hi; ++f) o.f_src[(long long)z * o.ts_f + f0 + f] = r;
        cum = hi64 < (1 << 28) ? (int)hi64 : (1 << 28);
    }
}

// mel rectangle rows: masks, the R -> F map and the target frames (one wavefront per row)
__global__ void plan_rows_r_kernel(PlanImage im, PlanOut o, int n_mel) {
    const int z = blockIdx.z;
    const PlanTaskHdr h = im.hdr[z];
    if (!h.with_frames) return;
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= h.Mr) return;
    const long long q = (long long)z * o.ts_r + row;
    int i = -1, t = 0;
    if (row >= kPlanG) { i = (row - kPlanG) / (h.Tcap + kPlanG); t = (row - kPlanG) - i * (h.Tcap + kPlanG); }
    const bool inrect = i >= 0 && i < h.B && t < h.Tcap;
    const bool valid = inrect && t < im.flen[(long long)z * im.cap_B + i];
    if (lane == 0) {
        o.r_inrect[q] = inrect; o.r_valid[q] = valid;
        { const float v = valid ? 1.f : 0.f, w = inrect ? 1.f : 0.f; st4(o.r_valid_w + 4 * q, make_float4(v, v, v, v)); st4(o.r_inrect_w + 4 * q, make_float4(w, w, w, w)); }
        o.r2f[q] = valid ? im.foff[(long long)z * im.cap_B + i] + t : -1;
        // frame-level targets: the padded part of the caller's [B][T_max] array is kept as it is (the reference bucketises it too)
        const bool intgt = inrect && h.has_targets && t < h.Tmax_in;
        if (im.pitch_frame) o.r_pitch_t[q] = intgt ? im.pitch[(long long)z * im.pe_cap_p + (long long)i * h.Tmax_in + t] : 0.f;
        if (im.energy_frame) o.r_energy_t[q] = intgt ? im.energy[(long long)z * im.pe_cap_e + (long long)i * h.Tmax_in + t] : 0.f;
    }
    if (!h.has_mels) return;
    float* dst = o.mel_tgt + (long long)z * o.ts_mel + (long long)row * n_mel;
    const float* src = valid ? im.mels + h.mel_off + ((long long)i * h.Tmax_in + t) * n_mel : nullptr;
    for (int c = lane * 4; c < n_mel; c += 256) st4(dst + c, src ? ld4(src + c) : zero4());
}

// free-running: predicted durations (float, P rows) -> compact [task][B * S] image for the single read-back
__global__ void plan_gather_durations_kernel(const int* meta, const float* d_rounded, long long d_ts, float* out, int cap_B, int cap_S) {
    const int z = blockIdx.z;
    const int B = meta[z * META_STRIDE + META_B], S = meta[z * META_STRIDE + META_SMAX];
    for (int e = blockIdx.x * (int)blockDim.x + (int)threadIdx.x; e < B * S; e += (int)(gridDim.x * blockDim.x)) {
        const int i = e / S, s = e - i * S;
        out[(long long)z * cap_B * cap_S + e] = d_rounded[(long long)z * d_ts + kPlanG + i * (S + kPlanG) + s];
    }
}

}  // namespace mtts
