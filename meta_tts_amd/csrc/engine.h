// Host-side engine of libmtts: parameter space, batch plans (row spaces), FastSpeech2 forward and
// hand-derived backward as sequences of grouped launches (gemm.h / rowops.h), the MAML inner/outer
// loop with per-task fast weights resident in HBM, and the fused clip + Adam outer update.
//
// Reference path restated here: lightning/model/fastspeech2.py:40-112 (+ base_adaptor.py:41-95
// learner variant), lightning/model/modules.py:102-158 (variance adaptor), :167-190 (length
// regulator), transformer/{Models,Layers,SubLayers,Modules}.py, lightning/model/loss.py:19-92,
// lightning/systems/base_adaptor.py:98-124 (adapt / meta_learn), lightning/optimizer.py,
// lightning/scheduler.py, main.py:61 (clip).
//
// Row spaces (per task; every activation is a [rows][C] matrix):
//   P  phoneme rectangle   row(b,s) = G + b*(Smax+G) + s          encoder, variance adaptor
//   F  frames, packed      row(b,t) = foff[b] + t, t < len_b      decoder (no padded frames)
//   R  mel rectangle       row(b,t) = G + b*(Tcap+G) + t          mel_linear output, PostNet, loss
// G = 4 zero guard rows separate sequences so a Conv1d is a GEMM over overlapping rows.  The
// rectangle spaces keep padded positions because the reference computes on them (variance
// predictor convs see speaker-embedding rows beyond src_len; PostNet BatchNorm statistics include
// padded frames).  The decoder only ever exposes valid frames to anything downstream, so it runs
// on the packed space (35 % fewer rows on LibriTTS-shaped batches).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "gemm.h"
#include "gemm_glds.h"
#include "gemm_bf16.h"
#include "attention.h"
#include "rowops.h"
#include "plan.h"
#include "tangent.h"

namespace mtts {

constexpr int G = 4;  // guard rows between sequences (>= max conv half-width)

struct ModelCfg {
    int d_model, enc_layers, dec_layers, enc_heads, dec_heads, d_ff, k1, k2;
    int vp_filter, vp_kernel, n_bins, max_seq_len, n_mel, vocab, n_speaker;
    int postnet_dim, postnet_kernel, postnet_layers;
    float pitch_min, pitch_max, energy_min, energy_max;
    float enc_dropout = 0.f, dec_dropout = 0.f, vp_dropout = 0.f, postnet_dropout = 0.5f;
    // preprocess_config["preprocessing"]["pitch" | "energy"]["feature"] == "frame_level" (modules.py:28-33,139-148): the predictor,
    // the bucketised embedding and the loss of that feature live on the mel-frame rectangle instead of the phoneme rectangle
    int pitch_frame = 0, energy_frame = 0;
    // which top-level modules are adapted in the inner loop (bit i of: encoder,
    // variance_adaptor, decoder, mel_linear, postnet, speaker_emb)
    int adapt_mask;
};
enum { MOD_ENCODER = 0, MOD_VA, MOD_DECODER, MOD_MEL, MOD_POSTNET, MOD_SPK, MOD_COUNT };
static const char* kModNames[MOD_COUNT] = {"encoder", "variance_adaptor", "decoder", "mel_linear", "postnet", "speaker_emb"};

struct HostBatch {  // one padded batch in the reference 12-tuple layout (collate.py:47-60), host memory
    int B, S_max, T_max;
    const long long* speakers;   // [B]
    const long long* texts;      // [B][S_max]
    const long long* src_lens;   // [B]
    const float* mels;           // [B][T_max][n_mel] or null
    const long long* mel_lens;   // [B] or null
    const float* pitches;        // [B][S_max] or null
    const float* energies;       // [B][S_max] or null
    const long long* durations;  // [B][S_max] or null
    const float* spk_emb = nullptr;  // [B][d_model] or null: external speaker embeddings instead of the table lookup (speaker_emb: dvec)
};

struct ParamEntry {
    std::string name;
    std::vector<int> shape;  // torch shape
    long long off = 0, numel = 0;
    int conv = 0;  // 1: torch (Cout, Cin, k) stored as [Cout][k][Cin]
    int module = 0;
};

struct TS { float* p; long long ts; };

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return -1; } } while (0)

class Engine {
public:
    ModelCfg cfg;
    int cap_tasks, cap_B, cap_S, cap_T, cap_Tc;
    int capMp, capMf, capMr;
    hipStream_t stream = nullptr;
    std::string last_error;

    // ---------------- parameter space -------------------------------------------------
    std::vector<ParamEntry> entries;
    std::map<std::string, int> by_name;
    long long n_total = 0, adapt_start = 0, n_adapt = 0;
    float *theta = nullptr, *adam_m = nullptr, *adam_v = nullptr, *fast = nullptr, *grad = nullptr, *outer = nullptr;
    float *fast_cur = nullptr, *grad_dst = nullptr;  // what W() / Gd() resolve to (second-order sweep redirects them)
    float *pos_table = nullptr, *pitch_bins = nullptr, *energy_bins = nullptr;
    int pos_rows = 0;
    std::vector<float*> bn_rm, bn_rv;
    std::vector<long long> bn_tracked;
    float *norm_partial = nullptr, *norm_out = nullptr;
    long long adam_step_count = 0;

    struct FFTP { long long wqkv, bqkv, ln1g, ln1b, wfc, bfc, w1, b1, w2, b2, ln2g, ln2b; };
    struct PredP { long long c1w, c1b, l1g, l1b, c2w, c2b, l2g, l2b, lw, lb; };
    struct PostP { long long w, b, g, beta; int cin, cout; };
    std::vector<FFTP> encP, decP;
    PredP durP, pitP, eneP;
    std::vector<PostP> postP;
    long long word_emb, pitch_emb, energy_emb, mel_w, mel_b, spk_table;

    // ---------------- plans -----------------------------------------------------------
    struct TaskIn {  // host copy of one batch (the caller's 12-tuple memory is only borrowed during set_batches)
        int B = 0, S = 0, T_max = 0;
        bool has_targets = false;
        std::vector<long long> texts, src_lens, mel_lens, durations;
        std::vector<float> pitches, energies;
        std::vector<float> d_rounded;  // free-running only: clamp(round(exp(logd) - 1) * d_control, 0) as the reference returns it (not truncated)
        const float* mels = nullptr;
        std::vector<float> mels_keep;  // own copy, only when T_max > max_seq_len (the plan may be rebuilt untruncated, see retarget)
        std::vector<int> spk_ids;
        std::vector<float> spk_emb;    // [B][d_model] external speaker embeddings (empty: table lookup)
    };
    struct Plan {
        int tasks = 0;
        std::vector<int> hB, hSmax, hTcap, hMp, hMf, hMr;
        int maxMp = 0, maxMf = 0, maxMr = 0, maxB = 0, enc_maxL = 0, dec_maxL = 0, n_enc_groups = 0, n_dec_groups = 0;
        int average_spk = 0;
        bool ext_spk = false;   // speaker vectors come with the batch (speaker_emb: dvec), no table lookup and no table gradient
        bool has_targets = false, frames_ready = false;
        bool over_max = false, truncated = true;  // some T_max > max_seq_len / frames beyond it dropped in the current row spaces
        unsigned drop_seed = 0;  // seed of the last train-mode forward on this plan (backward replays it)
        std::vector<TaskIn> in;
        double sum_nP = 0, sum_nF = 0, sum_attn_p = 0, sum_attn_f = 0;
        long long sumMp = 0, sumMf = 0, sumMr = 0, sumLp = 0, sumLf = 0;  // total rows over tasks (tile heuristic)  // valid rows / sum L^2 (algorithmic flop accounting)
        int* meta = nullptr;
        // P space
        int *p_row_b, *p_row_t, *p_tok, *p_first, *p_count, *p_dur, *p_seg_start, *p_seg_len;
        unsigned char *p_valid, *p_inrect;
        float *p_valid_w, *p_inrect_w, *f_valid_w, *r_valid_w, *r_inrect_w;   // [row][4] float images of the masks (plan.h)
        float *p_pitch_t, *p_energy_t;
        // F space
        int *f_row_b, *f_row_t, *f_src, *f_seg_start, *f_seg_len, *f2r;
        unsigned char* f_valid;
        // R space
        int* r2f;
        unsigned char *r_valid, *r_inrect;
        float *r_pitch_t = nullptr, *r_energy_t = nullptr;  // frame-level targets on the R rows
        float* mel_tgt;
        int* spk_ids;
        AttnSeq *enc_seqs, *dec_seqs;
        GemmGroupDesc *enc_tab[6], *dec_tab[6];
        // compact batch image (plan.h): device copy + two pinned staging buffers (the host may prepare step k+1 while the copy of
        // step k is still in flight)
        char* img_dev = nullptr;
        char* img_host[2] = {nullptr, nullptr};
        hipEvent_t img_done[2] = {nullptr, nullptr};
        bool img_pending[2] = {false, false};
        int img_parity = 0;
        float* dur_readback = nullptr;       // device [tasks][cap_B * cap_S]: predicted durations of a free-running pass
    };
    // byte offsets inside a compact image (fixed by the capacities; computed in init)
    size_t pe_cap_p = 0, pe_cap_e = 0;  // per-task floats of the pitch / energy areas of an image
    struct ImgLayout { size_t meta, hdr, src_len, flen, foff, spk, spk_emb, texts, dur, pitch, energy, seq_e, seq_d, tab_e[6], tab_d[6], mels, total; } img;
    enum { TAB_QK = 0, TAB_PV, TAB_DP, TAB_DV, TAB_DQ, TAB_DK };
    Plan plans[2];
    long long row_ts_p, row_ts_f, row_ts_r;  // strides of per-row index arrays

    // ---------------- workspace -------------------------------------------------------
    struct LayerBuf { TS qkv, P, O, z1, st1, y1, h, z2, st2, y2; };
    // Deferred weight gradients (under-filled launches only: single-task ranks, few-shot adaptation).  The gradients a layer's four
    // weight-gradient GEMMs consume (dz2 / dh / dz1 / dqkv) are written into per-layer buffers instead of the shared scratch, so the
    // GEMMs no longer have to run before the scratch is reused: they go out as ONE multi-problem launch per layer on a SIDE stream,
    // concurrently with the backward chain of the layers below, and are joined before anything reads the parameter gradients.
    struct LayerGrad { TS dc, gh, da, gqkv; float *part2, *part1; };   // part2 / part1: the 8-row gamma / beta partials LN2's / LN1's backward kernel leaves (folded on the side stream)
    std::vector<LayerGrad> encG, decG;
    // Beyond the deferred regime (more tasks per launch than defer_tasks): the LayerNorm gamma / beta folds of the FFT blocks still leave the critical
    // stream — every site keeps its backward kernel's partials in a buffer of its own ([cap_tasks][ln_chunks][3][d_model], 2 per layer) and the two
    // colfinal launches of a layer go out on the side stream behind the layer's backward (fft_bwd: fold_part).  MTTS_LN_FOLD_SIDE=0: on the main stream.
    float* arena_lnpart = nullptr;
    std::vector<float*> encLnPart, decLnPart;   // [2 * layer + 0] = LN2's partials, [2 * layer + 1] = LN1's
    struct PredGrad { TS g2a, g2b; float *part2, *part1; };   // variance predictor: d(conv2 out), d(conv1 out), the gamma / beta partials of its two LayerNorms
    PredGrad predG[3];
    std::vector<TS> postG;               // per PostNet layer: gradient of the conv output (what its weight-gradient GEMM reads)
    char* arena_defer = nullptr;
    size_t arena_defer_bytes = 0;
    float* arena_pred = nullptr;
    // ---- bf16 operand planes (bf16 numerics mode; enable_planes) ----
    // arena_h / arena_defer_h: a bf16 twin of every float of the two activation arenas at the same element index (a buffer's plane is a
    // pointer offset away: H()).  sh_*: shadows of the Conv1d weights in the forward layout (_f) and the input-gradient layout (_t), for
    // the masters (theta) and the per-task fast weights; refreshed by every forward pass (refresh_shadows).
    bf16_t *arena_h = nullptr, *arena_defer_h = nullptr;
    bf16_t *sh_theta_f = nullptr, *sh_theta_t = nullptr, *sh_fast_f = nullptr, *sh_fast_t = nullptr;
    ShadowEnt *d_ents_all = nullptr, *d_ents_theta = nullptr, *d_ents_fast = nullptr;
    int n_ents_all = 0, n_ents_theta = 0, n_ents_fast = 0, tiles_all = 0, tiles_theta = 0, tiles_fast = 0;
    std::map<long long, long long> shadow_off;   // weight offset in theta -> offset in the shadow vectors
    long long n_shadow = 0;
    const float* shadow_fast_src = nullptr;      // the fast-weight copy the fast shadows were made from
    bool shadows_current = false;                // the shadows hold the weights as a forward of the CURRENT numerics mode last saw them: cleared by
                                                 // whatever writes theta or switches the mode (a mode switch takes effect at the next forward)
    bool planes_ready = false;
    hipEvent_t ev_shadow = nullptr;
    bool shadow_wait = false;                    // an asynchronous refresh is in flight: the next shadow reader waits for ev_shadow
    bool planes_wanted = true;                   // numerics mode 1 (planes) vs 2 (bf16 operands rounded in the staging pass only)
    int defer_tasks = 0;                 // task capacity of the deferred buffers (0: not available)
    hipStream_t side = nullptr;
    static constexpr int kSideEvents = 32;   // more than the forks of one backward pass (19 at base.yaml): no event is re-recorded while a wait on it can be pending
    hipEvent_t ev_side[kSideEvents] = {};
    hipEvent_t ev_join = nullptr;
    int ev_next = 0;
    GemmCtx gx_side;
    float* col_partial_side = nullptr;   // the side stream's own scratch of the two-stage column reduction
    // Encoder run-ahead (same regime): a non-adapted encoder does not depend on the fast weights, so the encoder forwards of ALL inner
    // steps (they differ only by their dropout seeds) are enqueued on a second side stream before the inner loop and overlap it; step s
    // waits for its event and reads the kept copy of the encoder output
    static constexpr int kAhead = 8;
    hipStream_t side2 = nullptr;
    hipEvent_t ev_enc[kAhead] = {};
    TS enc_ahead[kAhead];
    GemmCtx gx_side2;
    bool defer_live = false;             // side-stream work of the current backward pass is outstanding
    struct PredBuf { TS r1, st1, n1, r2, st2, n2, out; };
    struct PostBuf { TS c, a, stats, dgamma_tmp; };
    std::vector<LayerBuf> encB, decB;
    PredBuf durB, pitB, eneB;
    std::vector<PostBuf> postB;
    TS emb_out, spk, x0, x1, x2, dec_in, mel, mel_post;
    int *pidx = nullptr, *eidx = nullptr;
    // frame-level features: activations of the variance adaptor's second half on the frame rectangle (row space R, d channels)
    bool any_frame_level() const { return cfg.pitch_frame || cfg.energy_frame; }
    TS xr0, xr1, xr2, gRx, gRf1, gRf2, gFx, dpred_r[2];
    TS va_out{nullptr, 0};
    PredBuf pitR, eneR;
    int *pidx_r = nullptr, *eidx_r = nullptr;
    TS d_rounded;
    // backward scratch
    TS gPm, gFm;  // dropout-masked copies of a LayerNorm input gradient
    TS gPxE, gPxP, gPxD, gPx2;           // input gradients of the energy / pitch / duration predictors (pred_bwd_early); dL/d(x2)
    hipEvent_t ev_pred = nullptr;
    TS gP0, gP1, gPqkv, gPh, gPf1, gPf2, gF0, gF1, gFqkv, gFh, dSp, dSf, gR0, gR1, gRm, gRp, gMelF, dspk, dpred[3];
    float *loss_partial = nullptr, *losses = nullptr, *col_partial = nullptr;
    int col_max_chunks = 0;
    long long col_partial_ts = 0;                 // floats per task of col_partial
    static int ln_chunks(int rows) { return (rows + kLnRows - 1) / kLnRows; }   // partial chunks of the LayerNorm backward (rowops.h)
    long long S_ts_p = 0, S_ts_f = 0;

    char* arena = nullptr;
    size_t arena_bytes = 0;
    // Gradients of a backward pass that second-order MAML reads again in its reverse sweep (engine_so.inc): per decoder layer the
    // gradient leaving the layer (g0) and dh, dy1, dO, dqkv; the gradient reaching every PostNet layer's output; the final mel gradient.
    // By default (gs_alias) every field aliases the shared backward scratch — one layer alive at a time, as a first-order pass needs —;
    // meta_grad_so gives the backward of inner step s a set of its own (gs_steps[s]), so that step's tangent backward finds the primal
    // gradients instead of recomputing them (one forward-equivalent of GEMMs per step).
    struct LayerKeep { TS g0, gh, dy1, dO, gqkv; };
    struct GradSet { std::vector<LayerKeep> dec; TS dec_top{nullptr, 0}; std::vector<TS> post_cur; TS gRm{nullptr, 0}; };
    GradSet gs_alias;
    std::vector<GradSet> gs_steps;
    std::vector<char*> gs_mem;
    int gs_bound = -1;                // -1: gs_alias
    GradSet& GK() { return gs_bound < 0 ? gs_alias : gs_steps[gs_bound]; }
    size_t act_bytes = 0;             // leading part of the arena: the activation set (layout_act)
    std::vector<char*> act_sets;      // extra activation sets (second-order MAML: one per inner step); set 0 is the arena's own
    int act_bound = 0;

    struct Pass;
    GemmCtx gx;  // this handle's launcher state: batching queue, profiler, split-K workspace, numerics mode (gemm.h)
struct GemmBatchScope {  // RAII around gemm_batch_begin / gemm_batch_end (gemm.h)
    GemmCtx& cx;
    hipStream_t s;
    GemmBatchScope(GemmCtx& c, hipStream_t st) : cx(c), s(st) { gemm_batch_begin(cx); }
    ~GemmBatchScope() { gemm_batch_end(cx, s); }
};
    // dropout (off by default: parity runs patch it to identity)
    // gradient accumulation (main.py:62 accumulate_grad_batches): meta_grad / plain_grad ADD their result to the outer buffer
    bool outer_accumulate = false;
    bool dropout_on = false;
    unsigned drop_base = 0x1234567u, drop_counter = 0;
    int site_base = 0;  // set by the caller of fft_* / pred_*: identifies the layer for the mask stream
    bool drop_active(const Pass& ps) const { return dropout_on && ps.train; }

    void set_error(const std::string& s) { last_error = s; }

    // =================================================================================
    // construction
    // =================================================================================
    void add_param(const std::string& name, std::vector<int> shape, int module, int conv = 0) {
        ParamEntry e;
        e.name = name; e.shape = shape; e.module = module; e.conv = conv;
        e.numel = 1;
        for (int s : shape) e.numel *= s;
        entries.push_back(e);
    }

    void build_param_table() {
        const int d = cfg.d_model;
        std::vector<ParamEntry> all;
        auto fft = [&](const std::string& pre, int layers, int module) {
            for (int i = 0; i < layers; ++i) {
                const std::string p = pre + ".layer_stack." + std::to_string(i);
                // q, k, v weights then biases stay adjacent: one fused [3d][d] projection
                add_param(p + ".slf_attn.w_qs.weight", {d, d}, module);
                add_param(p + ".slf_attn.w_ks.weight", {d, d}, module);
                add_param(p + ".slf_attn.w_vs.weight", {d, d}, module);
                add_param(p + ".slf_attn.w_qs.bias", {d}, module);
                add_param(p + ".slf_attn.w_ks.bias", {d}, module);
                add_param(p + ".slf_attn.w_vs.bias", {d}, module);
                add_param(p + ".slf_attn.layer_norm.weight", {d}, module);
                add_param(p + ".slf_attn.layer_norm.bias", {d}, module);
                add_param(p + ".slf_attn.fc.weight", {d, d}, module);
                add_param(p + ".slf_attn.fc.bias", {d}, module);
                add_param(p + ".pos_ffn.w_1.weight", {cfg.d_ff, d, cfg.k1}, module, 1);
                add_param(p + ".pos_ffn.w_1.bias", {cfg.d_ff}, module);
                add_param(p + ".pos_ffn.w_2.weight", {d, cfg.d_ff, cfg.k2}, module, 1);
                add_param(p + ".pos_ffn.w_2.bias", {d}, module);
                add_param(p + ".pos_ffn.layer_norm.weight", {d}, module);
                add_param(p + ".pos_ffn.layer_norm.bias", {d}, module);
            }
        };
        // module order: non-adapted modules first, adapted ones last => the fast weights of a
        // task are one contiguous slice [adapt_start, n_total)
        for (int pass = 0; pass < 2; ++pass) {
            for (int mod = 0; mod < MOD_COUNT; ++mod) {
                const bool adapted = (cfg.adapt_mask >> mod) & 1;
                if ((pass == 1) != adapted) continue;
                if (mod == MOD_ENCODER) {
                    add_param("encoder.src_word_emb.weight", {cfg.vocab, d}, mod);
                    fft("encoder", cfg.enc_layers, mod);
                } else if (mod == MOD_VA) {
                    const int f = cfg.vp_filter, k = cfg.vp_kernel;
                    for (const char* pr : {"duration_predictor", "pitch_predictor", "energy_predictor"}) {
                        const std::string p = std::string("variance_adaptor.") + pr;
                        add_param(p + ".conv_layer.conv1d_1.conv.weight", {f, d, k}, mod, 1);
                        add_param(p + ".conv_layer.conv1d_1.conv.bias", {f}, mod);
                        add_param(p + ".conv_layer.layer_norm_1.weight", {f}, mod);
                        add_param(p + ".conv_layer.layer_norm_1.bias", {f}, mod);
                        add_param(p + ".conv_layer.conv1d_2.conv.weight", {f, f, k}, mod, 1);
                        add_param(p + ".conv_layer.conv1d_2.conv.bias", {f}, mod);
                        add_param(p + ".conv_layer.layer_norm_2.weight", {f}, mod);
                        add_param(p + ".conv_layer.layer_norm_2.bias", {f}, mod);
                        add_param(p + ".linear_layer.weight", {1, f}, mod);
                        add_param(p + ".linear_layer.bias", {1}, mod);
                    }
                    add_param("variance_adaptor.pitch_embedding.weight", {cfg.n_bins, d}, mod);
                    add_param("variance_adaptor.energy_embedding.weight", {cfg.n_bins, d}, mod);
                } else if (mod == MOD_DECODER) {
                    fft("decoder", cfg.dec_layers, mod);
                } else if (mod == MOD_MEL) {
                    add_param("mel_linear.weight", {cfg.n_mel, d}, mod);
                    add_param("mel_linear.bias", {cfg.n_mel}, mod);
                } else if (mod == MOD_POSTNET) {
                    for (int i = 0; i < cfg.postnet_layers; ++i) {
                        const int cin = i == 0 ? cfg.n_mel : cfg.postnet_dim;
                        const int cout = i == cfg.postnet_layers - 1 ? cfg.n_mel : cfg.postnet_dim;
                        const std::string p = "postnet.convolutions." + std::to_string(i);
                        add_param(p + ".0.conv.weight", {cout, cin, cfg.postnet_kernel}, mod, 1);
                        add_param(p + ".0.conv.bias", {cout}, mod);
                        add_param(p + ".1.weight", {cout}, mod);
                        add_param(p + ".1.bias", {cout}, mod);
                    }
                } else if (mod == MOD_SPK) {
                    add_param("speaker_emb.model.weight", {cfg.n_speaker, d}, mod);
                }
            }
            if (pass == 0) {
                long long off = 0;
                for (auto& e : entries) off += (e.numel + 3) & ~3LL;
                adapt_start = off;
            }
        }
        long long off = 0;
        for (size_t i = 0; i < entries.size(); ++i) {
            entries[i].off = off;
            off += (entries[i].numel + 3) & ~3LL;  // keep every tensor 16-byte aligned
            by_name[entries[i].name] = (int)i;
        }
        n_total = off;
        n_adapt = n_total - adapt_start;
        auto O = [&](const std::string& n) { return entries[by_name.at(n)].off; };
        auto fftp = [&](const std::string& pre, int layers, std::vector<FFTP>& out) {
            for (int i = 0; i < layers; ++i) {
                const std::string p = pre + ".layer_stack." + std::to_string(i);
                FFTP f;
                f.wqkv = O(p + ".slf_attn.w_qs.weight"); f.bqkv = O(p + ".slf_attn.w_qs.bias");
                f.ln1g = O(p + ".slf_attn.layer_norm.weight"); f.ln1b = O(p + ".slf_attn.layer_norm.bias");
                f.wfc = O(p + ".slf_attn.fc.weight"); f.bfc = O(p + ".slf_attn.fc.bias");
                f.w1 = O(p + ".pos_ffn.w_1.weight"); f.b1 = O(p + ".pos_ffn.w_1.bias");
                f.w2 = O(p + ".pos_ffn.w_2.weight"); f.b2 = O(p + ".pos_ffn.w_2.bias");
                f.ln2g = O(p + ".pos_ffn.layer_norm.weight"); f.ln2b = O(p + ".pos_ffn.layer_norm.bias");
                out.push_back(f);
            }
        };
        fftp("encoder", cfg.enc_layers, encP);
        fftp("decoder", cfg.dec_layers, decP);
        auto pred = [&](const char* pr) {
            const std::string p = std::string("variance_adaptor.") + pr;
            PredP q;
            q.c1w = O(p + ".conv_layer.conv1d_1.conv.weight"); q.c1b = O(p + ".conv_layer.conv1d_1.conv.bias");
            q.l1g = O(p + ".conv_layer.layer_norm_1.weight"); q.l1b = O(p + ".conv_layer.layer_norm_1.bias");
            q.c2w = O(p + ".conv_layer.conv1d_2.conv.weight"); q.c2b = O(p + ".conv_layer.conv1d_2.conv.bias");
            q.l2g = O(p + ".conv_layer.layer_norm_2.weight"); q.l2b = O(p + ".conv_layer.layer_norm_2.bias");
            q.lw = O(p + ".linear_layer.weight"); q.lb = O(p + ".linear_layer.bias");
            return q;
        };
        durP = pred("duration_predictor"); pitP = pred("pitch_predictor"); eneP = pred("energy_predictor");
        for (int i = 0; i < cfg.postnet_layers; ++i) {
            const std::string p = "postnet.convolutions." + std::to_string(i);
            PostP q;
            q.w = O(p + ".0.conv.weight"); q.b = O(p + ".0.conv.bias"); q.g = O(p + ".1.weight"); q.beta = O(p + ".1.bias");
            q.cin = i == 0 ? cfg.n_mel : cfg.postnet_dim;
            q.cout = i == cfg.postnet_layers - 1 ? cfg.n_mel : cfg.postnet_dim;
            postP.push_back(q);
        }
        word_emb = O("encoder.src_word_emb.weight");
        pitch_emb = O("variance_adaptor.pitch_embedding.weight");
        energy_emb = O("variance_adaptor.energy_embedding.weight");
        mel_w = O("mel_linear.weight"); mel_b = O("mel_linear.bias");
        spk_table = O("speaker_emb.model.weight");
    }

    // ---- arena -----------------------------------------------------------------------
    size_t arena_off = 0;
    bool arena_dry = true;
    void* take(size_t bytes) {
        arena_off = (arena_off + 255) & ~(size_t)255;
        void* p = arena_dry ? nullptr : (void*)(arena + arena_off);
        arena_off += bytes;
        return p;
    }
    TS rows(int capM, int C) {  // [cap_tasks][G + capM + G][C], pointer at row 0
        const long long ts = (long long)(capM + 2 * G) * C;
        float* p = (float*)take((size_t)ts * cap_tasks * sizeof(float));
        return TS{arena_dry ? nullptr : p + (long long)G * C, ts};
    }
    TS flat(long long n) {
        n = (n + 3) & ~3LL;
        return TS{(float*)take((size_t)n * cap_tasks * sizeof(float)), n};
    }
    template <class T> T* arr(long long n_per_task) { return (T*)take((size_t)n_per_task * cap_tasks * sizeof(T)); }

    static long long attn_elems(int B, int H, int L) { return (long long)B * H * L * ((L + 3) & ~3); }

    // The arena starts with everything a forward pass writes and a later backward (or tangent) pass reads — the activation set —
    // followed by the backward scratch and the plan arrays.  Second-order MAML binds the activation set to one buffer per inner
    // step (bind_act) so that the reverse sweep finds the activations of step s where the first sweep left them.
    void layout() {
        layout_act();
        act_bytes = (arena_off + 255) & ~(size_t)255;
        layout_rest();
        gs_alias.dec.assign(cfg.dec_layers, LayerKeep{gF0, gFh, gF1, gF1, gFqkv});
        gs_alias.dec_top = gF0;
        gs_alias.post_cur.assign(std::max(cfg.postnet_layers - 1, 0), gR1);
        gs_alias.gRm = gRm;
    }
    void layout_gradset(GradSet& g) {
        const int d = cfg.d_model;
        g.dec.resize(cfg.dec_layers);
        for (LayerKeep& k : g.dec) {
            k.g0 = rows(capMf, d); k.gh = rows(capMf, cfg.d_ff); k.dy1 = rows(capMf, d); k.dO = rows(capMf, d); k.gqkv = rows(capMf, 3 * d);
        }
        g.dec_top = rows(capMf, d);
        g.post_cur.resize(std::max(cfg.postnet_layers - 1, 0));
        for (TS& t : g.post_cur) t = rows(capMr, std::max(cfg.postnet_dim, cfg.n_mel));
        g.gRm = rows(capMr, cfg.n_mel);
    }
    // n gradient sets of their own memory (zeroed, like the arena: guard rows stay zero); false when the device cannot hold them
    bool ensure_grad_sets(int n) {
        while ((int)gs_steps.size() < n) {
            char* save_arena = arena; const size_t save_off = arena_off; const bool save_dry = arena_dry;
            GradSet g;
            arena_dry = true; arena_off = 0; layout_gradset(g);
            const size_t bytes = arena_off + 256;
            char* mem = nullptr;
            bool ok = hipMalloc((void**)&mem, bytes) == hipSuccess;
            if (ok && hipMemset(mem, 0, bytes) != hipSuccess) { hipFree(mem); ok = false; }
            if (ok) { arena = mem; arena_dry = false; arena_off = 0; layout_gradset(g); }
            arena = save_arena; arena_off = save_off; arena_dry = save_dry;
            if (!ok) { (void)hipGetLastError(); return false; }
            gs_mem.push_back(mem);
            gs_steps.push_back(g);
        }
        return true;
    }
    // point every activation buffer at set k (0 = the arena's own; k >= 1 = act_sets[k - 1]); host-side only
    void bind_act(int k) {
        if (k == act_bound) return;
        char* save_arena = arena; const size_t save_off = arena_off; const bool save_dry = arena_dry;
        if (k > 0) arena = act_sets[k - 1];
        arena_dry = false; arena_off = 0;
        layout_act();
        arena = save_arena; arena_off = save_off; arena_dry = save_dry;
        act_bound = k;
    }
    // n extra activation sets (zeroed: the guard rows of every row buffer stay zero); false when the device cannot hold them
    bool ensure_act_sets(int n) {
        while ((int)act_sets.size() < n) {
            char* a = nullptr;
            if (hipMalloc((void**)&a, act_bytes + 256) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipMemset(a, 0, act_bytes + 256) != hipSuccess) { hipFree(a); (void)hipGetLastError(); return false; }
            act_sets.push_back(a);
        }
        return true;
    }
    void layout_act() {
        const int d = cfg.d_model, f = cfg.vp_filter;
        S_ts_p = attn_elems(cap_B, cfg.enc_heads, cap_S);
        S_ts_f = attn_elems(cap_B, cfg.dec_heads, cap_Tc);
        auto layer = [&](int capM, long long S_ts, std::vector<LayerBuf>& v, int n) {
            v.resize(n);
            for (int i = 0; i < n; ++i) {
                v[i].qkv = rows(capM, 3 * d); v[i].P = flat(S_ts); v[i].O = rows(capM, d); v[i].z1 = rows(capM, d);
                v[i].st1 = rows(capM, 2); v[i].y1 = rows(capM, d); v[i].h = rows(capM, cfg.d_ff); v[i].z2 = rows(capM, d);
                v[i].st2 = rows(capM, 2); v[i].y2 = rows(capM, d);
            }
        };
        emb_out = rows(capMp, d);
        layer(capMp, S_ts_p, encB, cfg.enc_layers);
        spk = flat((long long)cap_B * d);
        x0 = rows(capMp, d); x1 = rows(capMp, d); x2 = rows(capMp, d);
        for (PredBuf* pb : {&durB, &pitB, &eneB}) {
            pb->r1 = rows(capMp, f); pb->st1 = rows(capMp, 2); pb->n1 = rows(capMp, f);
            pb->r2 = rows(capMp, f); pb->st2 = rows(capMp, 2); pb->n2 = rows(capMp, f);
            pb->out = rows(capMp, 1);
        }
        pidx = arr<int>(capMp); eidx = arr<int>(capMp);
        d_rounded = rows(capMp, 1);
        dec_in = rows(capMf, d);
        if (any_frame_level()) {
            xr0 = rows(capMr, d); xr1 = rows(capMr, d); xr2 = rows(capMr, d);
            for (PredBuf* pb : {&pitR, &eneR}) {
                pb->r1 = rows(capMr, f); pb->st1 = rows(capMr, 2); pb->n1 = rows(capMr, f);
                pb->r2 = rows(capMr, f); pb->st2 = rows(capMr, 2); pb->n2 = rows(capMr, f);
                pb->out = rows(capMr, 1);
            }
            pidx_r = arr<int>(capMr); eidx_r = arr<int>(capMr);
            gRx = rows(capMr, d); gRf1 = rows(capMr, f); gRf2 = rows(capMr, f); gFx = rows(capMf, d);
            dpred_r[0] = rows(capMr, 1); dpred_r[1] = rows(capMr, 1);
        }
        layer(capMf, S_ts_f, decB, cfg.dec_layers);
        mel = rows(capMr, cfg.n_mel); mel_post = rows(capMr, cfg.n_mel);
        postB.resize(cfg.postnet_layers);
        for (int i = 0; i < cfg.postnet_layers; ++i) {
            const int c = postP[i].cout;
            postB[i].c = rows(capMr, c); postB[i].a = rows(capMr, c); postB[i].stats = flat(3LL * c);
            postB[i].dgamma_tmp = flat(2LL * c);
        }
    }
    void layout_rest() {
        const int d = cfg.d_model, f = cfg.vp_filter;
        // backward scratch
        gPm = rows(capMp, d); gFm = rows(capMf, d);
        gP0 = rows(capMp, d); gP1 = rows(capMp, d); gPqkv = rows(capMp, 3 * d); gPh = rows(capMp, cfg.d_ff);
        gPf1 = rows(capMp, f); gPf2 = rows(capMp, f);
        gF0 = rows(capMf, d); gF1 = rows(capMf, d); gFqkv = rows(capMf, 3 * d); gFh = rows(capMf, cfg.d_ff);
        dSp = flat(S_ts_p); dSf = flat(S_ts_f);
        const int pc = std::max(cfg.postnet_dim, cfg.n_mel);
        gR0 = rows(capMr, pc); gR1 = rows(capMr, pc); gRm = rows(capMr, cfg.n_mel); gRp = rows(capMr, cfg.n_mel);
        gMelF = rows(capMf, cfg.n_mel);
        dspk = flat((long long)cap_B * d);
        for (int i = 0; i < 3; ++i) dpred[i] = rows(capMp, 1);
        col_max_chunks = (std::max(std::max(capMp, capMf), capMr) + kRC - 1) / kRC;
        {   // two-stage column reductions: colpart's 32-row chunks x 3 x 1024, or the LayerNorm backward's 8-row chunks x 3 x (its width)
            const size_t ln_w = (size_t)std::max(cfg.d_model, cfg.vp_filter);
            const size_t per_task = std::max((size_t)col_max_chunks * 3 * 1024, (size_t)ln_chunks(std::max(std::max(capMp, capMf), capMr)) * 3 * ln_w);
            col_partial_ts = (long long)per_task;
            col_partial = (float*)take((size_t)cap_tasks * per_task * sizeof(float));
        }
        loss_partial = (float*)take((size_t)cap_tasks * kLossBlocks * 5 * sizeof(float));
        losses = (float*)take((size_t)cap_tasks * 6 * sizeof(float));
        // plans
        row_ts_p = capMp; row_ts_f = capMf; row_ts_r = capMr;
        for (int s = 0; s < 2; ++s) {
            Plan& p = plans[s];
            p.p_row_b = arr<int>(capMp); p.p_row_t = arr<int>(capMp); p.p_tok = arr<int>(capMp);
            p.p_first = arr<int>(capMp); p.p_count = arr<int>(capMp); p.p_dur = arr<int>(capMp);
            p.p_seg_start = arr<int>(cap_B); p.p_seg_len = arr<int>(cap_B);
            p.p_valid = arr<unsigned char>(capMp); p.p_inrect = arr<unsigned char>(capMp);
            p.p_valid_w = arr<float>(4LL * capMp); p.p_inrect_w = arr<float>(4LL * capMp); p.f_valid_w = arr<float>(4LL * capMf);
            p.r_valid_w = arr<float>(4LL * capMr); p.r_inrect_w = arr<float>(4LL * capMr);
            p.p_pitch_t = arr<float>(capMp); p.p_energy_t = arr<float>(capMp);
            p.f_row_b = arr<int>(capMf); p.f_row_t = arr<int>(capMf); p.f_src = arr<int>(capMf);
            p.f_seg_start = arr<int>(cap_B); p.f_seg_len = arr<int>(cap_B); p.f2r = arr<int>(capMf);
            p.f_valid = arr<unsigned char>(capMf);
            p.r2f = arr<int>(capMr); p.r_valid = arr<unsigned char>(capMr); p.r_inrect = arr<unsigned char>(capMr);
            p.mel_tgt = rows(capMr, cfg.n_mel).p;
            if (any_frame_level()) { p.r_pitch_t = arr<float>(capMr); p.r_energy_t = arr<float>(capMr); }
            p.spk_ids = arr<int>(cap_B + 1);
            p.dur_readback = arr<float>((long long)cap_B * cap_S);
        }
    }

    int init(const ModelCfg& c, int max_tasks, int max_B, int max_S, int max_T) {
        cfg = c;
        cap_tasks = max_tasks; cap_B = max_B; cap_S = max_S; cap_T = max_T;
        cap_Tc = max_T;  // eval-mode synthesis may exceed max_seq_len (Models.py:145-152 extends the table)
        if (cfg.d_model % 4 || cfg.d_ff % 16 || cfg.vp_filter % 16 || cfg.n_mel % 16 || cfg.postnet_dim % 16 ||
            cfg.d_model % 16 || cfg.d_model > 1024 || cfg.vp_filter > 1024 || cfg.postnet_dim > 1024 ||
            (cfg.d_model / cfg.enc_heads) % 16 || (cfg.d_model / cfg.dec_heads) % 16 || cfg.k1 / 2 > G ||
            cfg.k2 / 2 > G || cfg.vp_kernel / 2 > G || cfg.postnet_kernel / 2 > G) {
            set_error("unsupported model dimensions (channels must be multiples of 16 and <= 1024, kernels <= 9)");
            return -1;
        }
        capMp = G + cap_B * (cap_S + G);
        capMf = G + cap_B * (cap_Tc + G);
        capMr = capMf;
        build_param_table();
        HIP_CHECK(hipMalloc((void**)&theta, n_total * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&adam_m, n_total * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&adam_v, n_total * sizeof(float)));
        sync_tail = 8;   // 6 loss scalars (+ 2 pad), then per PostNet layer running mean | running var
        for (int i = 0; i < cfg.postnet_layers; ++i) sync_tail += 2LL * postP[i].cout;
        sync_tail = (sync_tail + 3) & ~3LL;
        HIP_CHECK(hipMalloc((void**)&outer, (n_total + sync_tail) * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&fast, (size_t)std::max<long long>(n_adapt, 4) * cap_tasks * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&grad, (size_t)n_total * cap_tasks * sizeof(float)));
        HIP_CHECK(hipMemset(theta, 0, n_total * sizeof(float)));
        HIP_CHECK(hipMemset(adam_m, 0, n_total * sizeof(float)));
        HIP_CHECK(hipMemset(adam_v, 0, n_total * sizeof(float)));
        HIP_CHECK(hipMemset(outer, 0, (n_total + sync_tail) * sizeof(float)));
        HIP_CHECK(hipMemset(grad, 0, (size_t)n_total * cap_tasks * sizeof(float)));
        fast_cur = fast; grad_dst = grad;
        if (gx.alloc_workspace()) { set_error("hipMalloc failed (split-K workspace)"); return -1; }
        if (init_defer() != 0) return -1;
        if (upd_setup() < 0) return -1;
        HIP_CHECK(hipMalloc((void**)&norm_partial, 1024 * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&norm_out, 4 * sizeof(float)));
        // frozen tables: sinusoid positions (Models.py:10-30, float64 math), linear bins
        pos_rows = std::max(cfg.max_seq_len + 1, std::max(cap_S, cap_T) + 1);
        std::vector<float> pt((size_t)pos_rows * cfg.d_model);
        for (int p = 0; p < pos_rows; ++p)
            for (int j = 0; j < cfg.d_model; ++j) {
                const double ang = (double)p / std::pow(10000.0, 2.0 * (double)(j / 2) / (double)cfg.d_model);
                pt[(size_t)p * cfg.d_model + j] = (float)((j % 2 == 0) ? std::sin(ang) : std::cos(ang));
            }
        HIP_CHECK(hipMalloc((void**)&pos_table, pt.size() * sizeof(float)));
        HIP_CHECK(hipMemcpy(pos_table, pt.data(), pt.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc((void**)&pitch_bins, cfg.n_bins * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&energy_bins, cfg.n_bins * sizeof(float)));
        set_bins(cfg.pitch_min, cfg.pitch_max, cfg.energy_min, cfg.energy_max);
        bn_rm.resize(cfg.postnet_layers); bn_rv.resize(cfg.postnet_layers); bn_tracked.assign(cfg.postnet_layers, 0);
        for (int i = 0; i < cfg.postnet_layers; ++i) {
            const int cc = postP[i].cout;
            HIP_CHECK(hipMalloc((void**)&bn_rm[i], cc * sizeof(float)));
            HIP_CHECK(hipMalloc((void**)&bn_rv[i], cc * sizeof(float)));
        }
        reset_bn();
        arena_dry = true; arena_off = 0;
        layout();
        arena_bytes = arena_off + 256;
        HIP_CHECK(hipMalloc((void**)&arena, arena_bytes));
        HIP_CHECK(hipMemset(arena, 0, arena_bytes));
        arena_dry = false; arena_off = 0;
        layout();
        return init_images();
    }

    // compact batch images: layout fixed by the capacities; one device copy + two pinned staging buffers per slot
    int init_images() {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t nt = cap_tasks, bs = (size_t)cap_B * cap_S, ne = nt * cap_B * cfg.enc_heads, nd = nt * cap_B * cfg.dec_heads;
        size_t o = 0;
        img.meta = o; o = al(o + nt * META_STRIDE * sizeof(int));
        img.hdr = o; o = al(o + nt * sizeof(PlanTaskHdr));
        img.src_len = o; o = al(o + nt * cap_B * sizeof(int));
        img.flen = o; o = al(o + nt * cap_B * sizeof(int));
        img.foff = o; o = al(o + nt * cap_B * sizeof(int));
        img.spk = o; o = al(o + nt * (cap_B + 1) * sizeof(int));
        img.spk_emb = o; o = al(o + nt * (size_t)cap_B * cfg.d_model * sizeof(float));
        img.texts = o; o = al(o + nt * bs * sizeof(int));
        img.dur = o; o = al(o + nt * bs * sizeof(int));
        pe_cap_p = cfg.pitch_frame ? (size_t)cap_B * cap_T : bs;
        pe_cap_e = cfg.energy_frame ? (size_t)cap_B * cap_T : bs;
        img.pitch = o; o = al(o + nt * pe_cap_p * sizeof(float));
        img.energy = o; o = al(o + nt * pe_cap_e * sizeof(float));
        img.seq_e = o; o = al(o + ne * sizeof(AttnSeq));
        img.seq_d = o; o = al(o + nd * sizeof(AttnSeq));
        for (int k = 0; k < 6; ++k) { img.tab_e[k] = o; o = al(o + ne * sizeof(GemmGroupDesc)); }
        for (int k = 0; k < 6; ++k) { img.tab_d[k] = o; o = al(o + nd * sizeof(GemmGroupDesc)); }
        img.mels = o; o = al(o + nt * (size_t)cap_B * cap_T * cfg.n_mel * sizeof(float));
        img.total = o;
        for (int sl = 0; sl < 2; ++sl) {
            Plan& p = plans[sl];
            HIP_CHECK(hipMalloc((void**)&p.img_dev, img.total));
            HIP_CHECK(hipMemset(p.img_dev, 0, img.mels));
            for (int b = 0; b < 2; ++b) {
                HIP_CHECK(hipHostMalloc((void**)&p.img_host[b], img.total));
                memset(p.img_host[b], 0, img.mels);
                HIP_CHECK(hipEventCreate(&p.img_done[b]));
            }
            p.meta = (int*)(p.img_dev + img.meta);
            p.enc_seqs = (AttnSeq*)(p.img_dev + img.seq_e);
            p.dec_seqs = (AttnSeq*)(p.img_dev + img.seq_d);
            for (int k = 0; k < 6; ++k) {
                p.enc_tab[k] = (GemmGroupDesc*)(p.img_dev + img.tab_e[k]);
                p.dec_tab[k] = (GemmGroupDesc*)(p.img_dev + img.tab_d[k]);
            }
        }
        return 0;
    }
    void destroy_images() {
        for (int sl = 0; sl < 2; ++sl) {
            Plan& p = plans[sl];
            if (p.img_dev) hipFree(p.img_dev);
            for (int b = 0; b < 2; ++b) {
                if (p.img_host[b]) hipHostFree(p.img_host[b]);
                if (p.img_done[b]) hipEventDestroy(p.img_done[b]);
            }
        }
    }

    void set_bins(float pmin, float pmax, float emin, float emax) {
        // np.linspace(min, max, n_bins - 1) in float64, cast to fp32 (meta_tts_amd.synth.make_params)
        const int nb = cfg.n_bins - 1;
        std::vector<float> pb(nb), eb(nb);
        for (int i = 0; i < nb; ++i) {
            const double t = nb > 1 ? (double)i / (double)(nb - 1) : 0.0;
            pb[i] = (float)((double)pmin + ((double)pmax - (double)pmin) * t);
            eb[i] = (float)((double)emin + ((double)emax - (double)emin) * t);
        }
        if (nb > 1) { pb[nb - 1] = pmax; eb[nb - 1] = emax; }
        hipMemcpy(pitch_bins, pb.data(), nb * sizeof(float), hipMemcpyHostToDevice);
        hipMemcpy(energy_bins, eb.data(), nb * sizeof(float), hipMemcpyHostToDevice);
    }

    void reset_bn() {
        for (int i = 0; i < cfg.postnet_layers; ++i) {
            const int cc = postP[i].cout;
            std::vector<float> one(cc, 1.f);
            hipMemset(bn_rm[i], 0, cc * sizeof(float));
            hipMemcpy(bn_rv[i], one.data(), cc * sizeof(float), hipMemcpyHostToDevice);
            bn_tracked[i] = 0;
        }
    }

    static bool ln_fold_side_on() { static const int on = [] { const char* e = getenv("MTTS_LN_FOLD_SIDE"); return e ? atoi(e) : 1; }(); return on != 0; }
    // buffers, stream and events of the deferred weight-gradient path (see LayerGrad)
    int init_defer() {
        static const int max_defer_tasks = [] { const char* e = getenv("MTTS_DEFER_TASKS"); return e ? atoi(e) : 2; }();   // (A/B runs: 4 / 8)
        defer_tasks = std::min(cap_tasks, max_defer_tasks);
        if (defer_tasks < 1) { defer_tasks = 0; return 0; }
        const int d = cfg.d_model;
        const long long per_row = 2LL * d + cfg.d_ff + 3LL * d;
        const int post_c = std::max(cfg.postnet_dim, cfg.n_mel);
        const size_t bytes = (size_t)defer_tasks * per_row * sizeof(float) *
                             ((size_t)cfg.enc_layers * (capMp + 2 * G) + (size_t)cfg.dec_layers * (capMf + 2 * G)) +
                             (size_t)defer_tasks * cfg.postnet_layers * (size_t)(capMr + 2 * G) * post_c * sizeof(float) +
                             (size_t)defer_tasks * 3 * 2 * (size_t)(capMp + 2 * G) * cfg.vp_filter * sizeof(float) +
                             (size_t)defer_tasks * 2 * ((size_t)cfg.enc_layers * ln_chunks(capMp) + (size_t)cfg.dec_layers * ln_chunks(capMf)) * 3 * d * sizeof(float) +
                             (size_t)defer_tasks * 3 * 2 * (size_t)ln_chunks(capMp) * 3 * cfg.vp_filter * sizeof(float) + 64 * 256 +
                             4096;
        HIP_CHECK(hipMalloc((void**)&arena_defer, bytes));
        arena_defer_bytes = bytes;
        HIP_CHECK(hipMemset(arena_defer, 0, bytes));
        char* cur = arena_defer;
        auto rows_d = [&](int capM, int C) {   // [defer_tasks][G + capM + G][C], pointer at row 0 (same shape as rows())
            const long long ts = (long long)(capM + 2 * G) * C;
            float* p0 = (float*)cur;
            cur += (size_t)ts * defer_tasks * sizeof(float);
            return TS{p0 + (long long)G * C, ts};
        };
        auto part_d = [&](int capM, int C) {   // [defer_tasks][ln_chunks(capM)][3][C]: the LayerNorm backward's partial sums of one site
            cur = (char*)(((uintptr_t)cur + 255) & ~(uintptr_t)255);
            float* p0 = (float*)cur;
            cur += (size_t)defer_tasks * ln_chunks(capM) * 3 * C * sizeof(float);
            return p0;
        };
        auto mk = [&](std::vector<LayerGrad>& v, int n, int capM) {
            v.resize(n);
            for (int i = 0; i < n; ++i) { v[i].dc = rows_d(capM, d); v[i].gh = rows_d(capM, cfg.d_ff); v[i].da = rows_d(capM, d); v[i].gqkv = rows_d(capM, 3 * d);
                                          v[i].part2 = part_d(capM, d); v[i].part1 = part_d(capM, d); }
        };
        mk(encG, cfg.enc_layers, capMp);
        mk(decG, cfg.dec_layers, capMf);
        if (cap_tasks > defer_tasks && ln_fold_side_on()) {
            const size_t pe = (((size_t)cap_tasks * ln_chunks(capMp) * 3 * d * sizeof(float)) + 255) & ~(size_t)255;
            const size_t pd = (((size_t)cap_tasks * ln_chunks(capMf) * 3 * d * sizeof(float)) + 255) & ~(size_t)255;
            HIP_CHECK(hipMalloc((void**)&arena_lnpart, 2 * (cfg.enc_layers * pe + cfg.dec_layers * pd) + 256));
            char* c2 = (char*)arena_lnpart;
            encLnPart.resize(2 * (size_t)cfg.enc_layers); decLnPart.resize(2 * (size_t)cfg.dec_layers);
            for (auto& q : encLnPart) { q = (float*)c2; c2 += pe; }
            for (auto& q : decLnPart) { q = (float*)c2; c2 += pd; }
        }
        {   // the early predictor backward's buffers: every task of a launch (not only the deferred regime's)
            const long long ts = (long long)(capMp + 2 * G) * d;
            HIP_CHECK(hipMalloc((void**)&arena_pred, (size_t)(4 + kAhead) * cap_tasks * ts * sizeof(float)));
            HIP_CHECK(hipMemset(arena_pred, 0, (size_t)(4 + kAhead) * cap_tasks * ts * sizeof(float)));
            TS* const bufs[4] = {&gPxE, &gPxP, &gPxD, &gPx2};
            for (int i = 0; i < 4; ++i) *bufs[i] = TS{arena_pred + (long long)i * cap_tasks * ts + (long long)G * d, ts};
            for (int i = 0; i < kAhead; ++i) enc_ahead[i] = TS{arena_pred + (long long)(4 + i) * cap_tasks * ts + (long long)G * d, ts};   // (the encoder run-ahead's outputs)
        }
        for (auto& pg : predG) { pg.g2a = rows_d(capMp, cfg.vp_filter); pg.g2b = rows_d(capMp, cfg.vp_filter); pg.part2 = part_d(capMp, cfg.vp_filter); pg.part1 = part_d(capMp, cfg.vp_filter); }
        postG.resize(cfg.postnet_layers);
        for (auto& t : postG) t = rows_d(capMr, post_c);
        HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));   // non-blocking (a blocking stream would serialise with the legacy default stream on every launch)
        for (auto& e : ev_side) HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventCreate(&ev_join));
        HIP_CHECK(hipEventCreate(&ev_pred));
        HIP_CHECK(hipStreamCreateWithFlags(&side2, hipStreamNonBlocking));
        for (auto& e : ev_enc) HIP_CHECK(hipEventCreate(&e));
        gx_side2.no_glds = gx_side.no_glds;
        if (gx_side2.alloc_workspace()) { set_error("hipMalloc failed (split-K workspace of the run-ahead stream)"); return -1; }
        const int side_chunks = (std::max(std::max(capMp, capMf), capMr) + kRC - 1) / kRC;   // == col_max_chunks (set by layout(), later)
        {   // (sized like col_partial: the early predictor backward runs its LayerNorm reductions through it too)
            const size_t ln_w = (size_t)std::max(cfg.d_model, cfg.vp_filter);
            const size_t per_task = std::max((size_t)side_chunks * 3 * 1024, (size_t)ln_chunks(std::max(std::max(capMp, capMf), capMr)) * 3 * ln_w);
            HIP_CHECK(hipMalloc((void**)&col_partial_side, (size_t)cap_tasks * per_task * sizeof(float)));
        }
        gx_side.no_glds = true;
        // The weight-gradient side stream's launches take the BK = 16 kernels whatever their K (round 6): 20 KB of LDS per workgroup instead of 37, so a
        // 752-workgroup weight-gradient batch leaves room for TWO main-stream workgroups per CU instead of one — the critical stream's next launch (w_2's
        // input gradient: 496 short tiles) no longer queues behind it.  Same k-ordered MFMA chain per tile, results equal to fp32 roundoff (tests/test_gpu_timed_config.py);
        // single-task rank 31.05 -> 30.70 ms, its second order 77.3 -> 76.6 ms, C2 fp32 12.81 -> 12.69 ms, 8-task step unchanged (profiles/r06_ab_log.md).
        { static const int bk16 = [] { const char* e = getenv("MTTS_SIDE_BK16"); return e ? atoi(e) : 1; }(); gx_side.prefer_bk16 = bk16 != 0; }
        if (gx_side.alloc_workspace()) { set_error("hipMalloc failed (split-K workspace of the side stream)"); return -1; }
        return 0;
    }
    // =================================================================================
    // bf16 operand planes (bf16 numerics mode): allocation, weight shadows, plane lookup
    // =================================================================================
    // Called when the bf16 mode is first selected.  Planes serve the long NT problems of the step — the FFT blocks' k = 9 / k = 1
    // convolutions and the PostNet's, forward and input gradient: the input-gradient conv runs as an NT problem over the transposed
    // shadow (weight_shadow_kernel) instead of the NN form.
    int enable_planes() {
        if (planes_ready) return 0;
        std::vector<ShadowEnt> all, th, fa;
        n_shadow = 0;
        auto add = [&](long long off, int cout, int k, int cin) {
            if (cin % 8 != 0 || cout % 8 != 0) return;
            ShadowEnt e{off, n_shadow, cout, k, cin, 0};
            shadow_off[off] = n_shadow;
            all.push_back(e);
            if (n_adapt > 0 && off >= adapt_start && off < adapt_start + n_adapt) { e.off = off - adapt_start; fa.push_back(e); }
            else th.push_back(e);
            n_shadow += ((long long)cout * k * cin + 7) & ~7LL;
        };
        for (const auto* v : {&encP, &decP})
            for (const FFTP& P : *v) { add(P.w1, cfg.d_ff, cfg.k1, cfg.d_model); add(P.w2, cfg.d_model, cfg.k2, cfg.d_ff); }
        for (const PostP& P : postP) add(P.w, P.cout, cfg.postnet_kernel, P.cin);
        auto finish = [&](std::vector<ShadowEnt>& v, ShadowEnt** dev, int* n, int* tiles) -> int {
            int t0 = 0;
            for (ShadowEnt& e : v) { e.tile0 = t0; t0 += e.k * ((e.cout + 31) / 32) * ((e.cin + 31) / 32); }
            *n = (int)v.size(); *tiles = t0;
            if (v.empty()) return 0;
            if (hipMalloc((void**)dev, v.size() * sizeof(ShadowEnt)) != hipSuccess) return -1;
            return hipMemcpy(*dev, v.data(), v.size() * sizeof(ShadowEnt), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
        };
        if (finish(all, &d_ents_all, &n_ents_all, &tiles_all) || finish(th, &d_ents_theta, &n_ents_theta, &tiles_theta) ||
            finish(fa, &d_ents_fast, &n_ents_fast, &tiles_fast)) { destroy_planes(); set_error("hipMalloc failed (weight shadow tables)"); return -1; }
        // compact shadow vectors (every shadow starts on a multiple of 8 elements: 16-byte loads), one per task for the fast weights
        const size_t th_b = (size_t)(n_shadow + 8) * sizeof(bf16_t), fa_b = (size_t)cap_tasks * (size_t)n_shadow * sizeof(bf16_t) + 16;
        if (hipMalloc((void**)&sh_theta_f, th_b) != hipSuccess || hipMalloc((void**)&sh_theta_t, th_b) != hipSuccess ||
            (n_adapt > 0 && (hipMalloc((void**)&sh_fast_f, fa_b) != hipSuccess || hipMalloc((void**)&sh_fast_t, fa_b) != hipSuccess)) ||
            hipMalloc((void**)&arena_h, arena_bytes / 2 + 64) != hipSuccess ||
            (arena_defer && hipMalloc((void**)&arena_defer_h, arena_defer_bytes / 2 + 64) != hipSuccess)) {
            destroy_planes();
            set_error("hipMalloc failed (bf16 operand planes: half the activation arena again + the weight shadows)");
            return -1;
        }
        hipMemset(arena_h, 0, arena_bytes / 2 + 64);
        if (arena_defer_h) hipMemset(arena_defer_h, 0, arena_defer_bytes / 2 + 64);
        if (hipEventCreate(&ev_shadow) != hipSuccess) ev_shadow = nullptr;
        planes_ready = true;
        return 0;
    }
    void destroy_planes() {
        for (void* q : {(void*)arena_h, (void*)arena_defer_h, (void*)sh_theta_f, (void*)sh_theta_t, (void*)sh_fast_f, (void*)sh_fast_t,
                        (void*)d_ents_all, (void*)d_ents_theta, (void*)d_ents_fast})
            if (q) hipFree(q);
        if (ev_shadow) hipEventDestroy(ev_shadow);
        ev_shadow = nullptr;
        arena_h = arena_defer_h = sh_theta_f = sh_theta_t = sh_fast_f = sh_fast_t = nullptr;
        d_ents_all = d_ents_theta = d_ents_fast = nullptr;
        shadow_off.clear();
        planes_ready = false;
    }
    bool planes_on() const {
        static const int on = [] { const char* e = getenv("MTTS_BF16_PLANES"); return e ? atoi(e) : 1; }();
        return on && planes_ready && planes_wanted && gx.bf16;
    }
    // the weights this pass reads -> their shadows
    // async: on the second side stream (idle outside an encoder run-ahead), beside the pass's first kernels; the first conv that reads a
    // shadow waits for it (shadow_wait)
    void refresh_shadows(const Pass& ps, bool async = false) {
        if (!planes_on()) return;
        hipStream_t st = stream;
        async = async && side2 != nullptr && ev_shadow != nullptr;
        if (async) {
            hipEvent_t ev = ev_side[ev_next];
            ev_next = (ev_next + 1) % kSideEvents;
            hipEventRecord(ev, stream);          // (after whatever last wrote the weights on this stream: the optimizer, the inner update)
            hipStreamWaitEvent(side2, ev, 0);
            st = side2;
        }
        auto run = [&](const ShadowEnt* ents, int n, int tiles, const float* src, long long src_ts, bf16_t* f, bf16_t* t, long long dst_ts, int nt) {
            if (n > 0) MTTS_LAUNCH(weight_shadow_kernel, dim3((unsigned)tiles, 1, (unsigned)nt), dim3(256), st, ents, n, src, src_ts, f, t, dst_ts);
        };
        if (ps.use_fast && n_adapt > 0) {
            run(d_ents_theta, n_ents_theta, tiles_theta, theta, 0, sh_theta_f, sh_theta_t, 0, 1);
            run(d_ents_fast, n_ents_fast, tiles_fast, fast_cur, n_adapt, sh_fast_f, sh_fast_t, n_shadow, ps.pl->tasks);
            shadow_fast_src = fast_cur;
        } else {
            run(d_ents_all, n_ents_all, tiles_all, theta, 0, sh_theta_f, sh_theta_t, 0, 1);
        }
        shadows_current = true;
        if (async) { hipEventRecord(ev_shadow, side2); shadow_wait = true; }
    }
    void await_shadows() { if (shadow_wait) { hipStreamWaitEvent(stream, ev_shadow, 0); shadow_wait = false; } }
    struct HP { const bf16_t* p; long long ts; };
    // shadow of the weight behind W(ps, off) (tr: the input-gradient layout); null when there is none or it is not current
    HP Wh(const Pass& ps, TS w, bool tr) const {
        if (!planes_on() || !shadows_current) return HP{nullptr, 0};
        const bool is_fast = w.ts != 0;
        const long long off = is_fast ? (w.p - fast_cur) + adapt_start : w.p - theta;
        if (is_fast ? (fast_cur != shadow_fast_src || off < adapt_start || off >= adapt_start + n_adapt) : (off < 0 || off >= n_total)) return HP{nullptr, 0};
        // (a theta shadow of an adapted weight is not refreshed by a pass that reads the fast weights)
        if (!is_fast && ps.use_fast && n_adapt > 0 && off >= adapt_start && off < adapt_start + n_adapt) return HP{nullptr, 0};
        const auto it = shadow_off.find(off);
        if (it == shadow_off.end()) return HP{nullptr, 0};
        if (is_fast) return HP{(tr ? sh_fast_t : sh_fast_f) + it->second, n_shadow};
        return HP{(tr ? sh_theta_t : sh_theta_f) + it->second, 0};
    }
    // the plane of an arena buffer (null outside the two arenas: activation sets of second-order MAML, external buffers)
    bf16_t* H(const float* q) const {
        if (!planes_on()) return nullptr;
        const char* c = (const char*)q;
        if (c >= arena && c < arena + arena_bytes) return arena_h + (q - (const float*)arena);
        if (arena_defer && c >= arena_defer && c < arena_defer + arena_defer_bytes) return arena_defer_h + (q - (const float*)arena_defer);
        return nullptr;
    }
    // make the plane of the slab [tasks][G + rows + G][C] behind x (x.p = row 0 of task 0) from its fp32 values; returns it
    const bf16_t* make_plane(TS x, int C, int nt) {
        bf16_t* h = H(x.p);
        if (!h || (x.ts % 8) != 0 || (((long long)G * C) % 8) != 0) return nullptr;
        const long long n8 = x.ts * nt / 8, o = (long long)G * C;
        MTTS_LAUNCH(to_bf16_kernel, dim3((unsigned)std::min<long long>((n8 + 255) / 256, 4096)), dim3(256), stream, (const float*)(x.p - o), h - o, n8);
        return h;
    }
    void destroy() {
        ar_destroy();
        upd_destroy();
        if (side) { hipStreamSynchronize(side); hipStreamDestroy(side); }
        for (auto& e : ev_side) if (e) hipEventDestroy(e);
        if (ev_join) hipEventDestroy(ev_join);
        if (ev_pred) hipEventDestroy(ev_pred);
        if (side2) { hipStreamSynchronize(side2); hipStreamDestroy(side2); }
        for (auto& e : ev_enc) if (e) hipEventDestroy(e);
        gx_side2.release();
        gx_side.release();
        if (arena_defer) hipFree(arena_defer);
        if (arena_lnpart) hipFree(arena_lnpart);
        if (arena_pred) hipFree(arena_pred);
        destroy_planes();
        if (col_partial_side) hipFree(col_partial_side);
        for (float* p : {theta, adam_m, adam_v, outer, fast, grad, norm_partial, norm_out, pos_table, pitch_bins, energy_bins})
            if (p) hipFree(p);
        for (float* p : bn_rm) hipFree(p);
        for (float* p : bn_rv) hipFree(p);
        gx.release();
        destroy_imaml();
        destroy_images();
        if (arena) hipFree(arena);
        for (char* a : act_sets) if (a) hipFree(a);
        for (char* a : gs_mem) if (a) hipFree(a);
        if (arena_so) hipFree(arena_so);
        if (arena_so_defer) hipFree(arena_so_defer);
        if (arena_so_defer_post) hipFree(arena_so_defer_post);
        if (hv) hipFree(hv);
        if (fast_hist) hipFree(fast_hist);
    }

    // =================================================================================
    // parameter import / export (torch layout <-> internal layout)
    // =================================================================================
    // which: 0 theta, 4 Adam exp_avg, 5 Adam exp_avg_sq (checkpoint resume)
    int load_param(const std::string& name, const float* host, long long numel, int which = 0) {
        auto it = by_name.find(name);
        if (it == by_name.end()) { set_error("unknown parameter: " + name); return -1; }
        const ParamEntry& e = entries[it->second];
        if (numel != e.numel) { set_error("size mismatch for " + name); return -1; }
        std::vector<float> tmp;
        const float* src = host;
        if (e.conv) {
            const int co = e.shape[0], ci = e.shape[1], k = e.shape[2];
            tmp.resize(e.numel);
            for (int o = 0; o < co; ++o)
                for (int c = 0; c < ci; ++c)
                    for (int kk = 0; kk < k; ++kk)
                        tmp[((size_t)o * k + kk) * ci + c] = host[((size_t)o * ci + c) * k + kk];
            src = tmp.data();
        }
        float* dst = which == 0 ? theta : (which == 4 ? adam_m : (which == 5 ? adam_v : nullptr));
        if (!dst) { set_error("bad import selector"); return -1; }
        HIP_CHECK(hipMemcpy(dst + e.off, src, e.numel * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }

    // which: 0 theta, 1 outer gradient, 2 per-task gradient, 3 fast weights of `task`, 4 adam m, 5 adam v
    int export_param(const std::string& name, int which, int task, float* host, long long numel) {
        auto it = by_name.find(name);
        if (it == by_name.end()) { set_error("unknown parameter: " + name); return -1; }
        const ParamEntry& e = entries[it->second];
        if (numel != e.numel) { set_error("size mismatch for " + name); return -1; }
        const float* src = nullptr;
        if (which == 0) src = theta + e.off;
        else if (which == 1) src = outer + e.off;
        else if (which == 2) src = grad + (long long)task * n_total + e.off;
        else if (which == 3) {
            if (e.off < adapt_start) src = theta + e.off;
            else src = fast + (long long)task * n_adapt + (e.off - adapt_start);
        } else if (which == 4) src = adam_m + e.off;
        else if (which == 5) src = adam_v + e.off;
        else if (which == 6 && hv) src = hv + (long long)task * n_total + e.off;
        else { set_error("bad export selector"); return -1; }
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<float> tmp(e.numel);
        HIP_CHECK(hipMemcpy(tmp.data(), src, e.numel * sizeof(float), hipMemcpyDeviceToHost));
        if (e.conv) {
            const int co = e.shape[0], ci = e.shape[1], k = e.shape[2];
            for (int o = 0; o < co; ++o)
                for (int c = 0; c < ci; ++c)
                    for (int kk = 0; kk < k; ++kk)
                        host[((size_t)o * ci + c) * k + kk] = tmp[((size_t)o * k + kk) * ci + c];
        } else {
            memcpy(host, tmp.data(), e.numel * sizeof(float));
        }
        return 0;
    }

    // =================================================================================
    // batch plans
    // =================================================================================
    // Copy `tasks` host batches into plan slot `slot` and build its row spaces.  Teacher-forced batches
    // (durations given) get all three spaces now; free-running batches (no durations) get the phoneme
    // space only — forward() sizes the frame spaces once the predicted durations are known.
    // spk_from (optional, per task): ids whose table rows are averaged (query pass of MAML).
    int set_batches(int slot, int tasks, const HostBatch* hb, const HostBatch* spk_from, int average_spk) {
        if (tasks < 1 || tasks > cap_tasks) { set_error("task count exceeds capacity"); return -1; }
        Plan& p = plans[slot];
        p.tasks = tasks;
        p.average_spk = average_spk;
        p.in.assign(tasks, TaskIn());
        bool any_tf = false, any_fr = false;
        for (int t = 0; t < tasks; ++t) {
            const HostBatch& b = hb[t];
            TaskIn& in = p.in[t];
            if (b.B < 1 || b.B > cap_B || b.S_max < 1 || b.S_max > cap_S) { set_error("batch exceeds engine capacity (B or S_max)"); return -1; }
            in.B = b.B; in.S = b.S_max; in.T_max = b.T_max;
            in.has_targets = b.durations && b.mel_lens && b.mels && b.pitches && b.energies;
            if (!in.has_targets && (b.durations || b.mels)) { set_error("teacher-forced batches need mels, mel_lens, pitches, energies, durations"); return -1; }
            (in.has_targets ? any_tf : any_fr) = true;
            const size_t BS = (size_t)b.B * b.S_max;
            in.texts.assign(b.texts, b.texts + BS);
            in.src_lens.assign(b.src_lens, b.src_lens + b.B);
            for (int i = 0; i < b.B; ++i)
                if (in.src_lens[i] < 1 || in.src_lens[i] > b.S_max) { set_error("src_len out of range"); return -1; }
            for (int i = 0; i < b.B; ++i)  // phoneme ids index the embedding table on the device: reject what nn.Embedding would
                for (long long s2 = 0; s2 < in.src_lens[i]; ++s2) {
                    const long long tk = in.texts[(size_t)i * b.S_max + s2];
                    if (tk < 0 || tk >= cfg.vocab) { set_error("phoneme token id out of range"); return -1; }
                }
            if (in.has_targets) {
                if (b.T_max < 1 || b.T_max > cap_T) { set_error("batch exceeds engine capacity (T_max)"); return -1; }
                const size_t BT = (size_t)b.B * b.T_max;
                in.pitches.assign(b.pitches, b.pitches + (cfg.pitch_frame ? BT : BS));     // [B][T_max] when the feature is frame-level
                in.energies.assign(b.energies, b.energies + (cfg.energy_frame ? BT : BS));
                if (any_frame_level() && b.T_max > cfg.max_seq_len) { set_error("frame-level pitch / energy with T_max > max_seq_len is not supported"); return -1; }
                in.durations.assign(b.durations, b.durations + BS);
                in.mel_lens.assign(b.mel_lens, b.mel_lens + b.B);
                in.mels = b.mels;
            }
            const HostBatch& sb = spk_from ? spk_from[t] : b;
            if (sb.B > cap_B) { set_error("speaker id list exceeds capacity"); return -1; }
            in.spk_ids.assign(cap_B + 1, 0);
            for (int i = 0; i < sb.B; ++i) {
                if (sb.speakers[i] < 0 || sb.speakers[i] >= cfg.n_speaker) { set_error("speaker id out of range"); return -1; }
                in.spk_ids[i] = (int)sb.speakers[i];
            }
            in.spk_ids[cap_B] = sb.B;
            if (!average_spk && sb.B != b.B) { set_error("speaker id count != batch size"); return -1; }
            in.spk_emb.clear();
            if (b.spk_emb) {
                if (average_spk || spk_from) { set_error("external speaker embeddings cannot be combined with spk_from / average_spk"); return -1; }
                in.spk_emb.assign(b.spk_emb, b.spk_emb + (size_t)b.B * cfg.d_model);
            }
            if (t > 0 && (in.spk_emb.empty() != p.in[0].spk_emb.empty())) { set_error("either every task or none carries speaker embeddings"); return -1; }
        }
        p.ext_spk = !p.in[0].spk_emb.empty();
        if (any_tf && any_fr) { set_error("cannot mix teacher-forced and free-running batches in one slot"); return -1; }
        p.has_targets = any_tf;
        p.over_max = false;
        for (auto& in : p.in)
            if (in.has_targets && in.T_max > cfg.max_seq_len) {   // rare: keep the targets so the plan can be rebuilt (retarget)
                p.over_max = true;
                in.mels_keep.assign(in.mels, in.mels + (size_t)in.B * in.T_max * cfg.n_mel);
            }
        const int rc = build_plan(slot, any_tf, true);  // copies the caller's mels into the pinned image: nothing borrowed afterwards
        for (auto& in : p.in) in.mels = in.mels_keep.empty() ? nullptr : in.mels_keep.data();
        return rc;
    }

    // The reference's Decoder drops frames beyond max_seq_len in training mode only; in eval mode it extends the sinusoid
    // table instead (transformer/Models.py:145-162).  Teacher-forced plans are built truncated (training is the hot path);
    // an eval-mode forward of a batch longer than max_seq_len rebuilds the row spaces untruncated, and back.
    int retarget(int slot, bool train) {
        Plan& p = plans[slot];
        if (!p.over_max || !p.has_targets || p.truncated == train) return 0;
        return build_plan(slot, true, train);
    }

    // (Re)build plan `slot` from p.in.  with_frames: durations / mel_lens are known.  truncate: frames beyond max_seq_len are
    // dropped (Decoder in training mode, Models.py:154-162).  The host computes O(B) scalars per task and the attention
    // descriptor tables, packs the batch into the pinned compact image and enqueues ONE copy + the four plan kernels (plan.h):
    // nothing here waits for the device (the staging buffer is double-buffered behind an event).
    int build_plan(int slot, bool with_frames, bool truncate) {
        Plan& p = plans[slot];
        const int tasks = p.tasks;
        p.hB.assign(tasks, 0); p.hSmax.assign(tasks, 0); p.hTcap.assign(tasks, 0);
        p.hMp.assign(tasks, 0); p.hMf.assign(tasks, 0); p.hMr.assign(tasks, 0);
        p.maxMp = p.maxMf = p.maxMr = p.maxB = p.enc_maxL = p.dec_maxL = 0;
        p.frames_ready = with_frames;
        p.truncated = truncate;
        p.sum_nP = p.sum_nF = p.sum_attn_p = p.sum_attn_f = 0;
        p.sumMp = p.sumMf = p.sumMr = p.sumLp = p.sumLf = 0;
        const int par = p.img_parity;
        p.img_parity ^= 1;
        if (p.img_pending[par]) { HIP_CHECK(hipEventSynchronize(p.img_done[par])); p.img_pending[par] = false; }
        char* H = p.img_host[par];
        int* meta = (int*)(H + img.meta);
        PlanTaskHdr* hdr = (PlanTaskHdr*)(H + img.hdr);
        int* h_src = (int*)(H + img.src_len); int* h_flen = (int*)(H + img.flen); int* h_foff = (int*)(H + img.foff);
        int* h_spk = (int*)(H + img.spk); int* h_txt = (int*)(H + img.texts); int* h_dur = (int*)(H + img.dur);
        float* h_pit = (float*)(H + img.pitch); float* h_ene = (float*)(H + img.energy);
        float* h_mel = (float*)(H + img.mels);
        AttnSeq* seqs[2] = {(AttnSeq*)(H + img.seq_e), (AttnSeq*)(H + img.seq_d)};
        GemmGroupDesc* tabs[2][6];
        for (int k = 0; k < 6; ++k) { tabs[0][k] = (GemmGroupDesc*)(H + img.tab_e[k]); tabs[1][k] = (GemmGroupDesc*)(H + img.tab_d[k]); }
        int nseq[2] = {0, 0};
        long long mel_used = 0;
        const int d = cfg.d_model;
        const size_t bs = (size_t)cap_B * cap_S;
        bool any_mels = false;
        for (int t = 0; t < tasks; ++t) {
            const TaskIn& b = p.in[t];
            const int B = b.B, S = b.S;
            const int Tcap = with_frames ? (truncate ? std::min(b.T_max, cfg.max_seq_len) : b.T_max) : 0;
            if (with_frames && (Tcap < 1 || Tcap > cap_T)) { set_error("mel length exceeds engine capacity (T_max)"); return -1; }
            const int Mp = G + B * (S + G);
            int foff = G, nP = 0, nF = 0;
            for (int i = 0; i < B; ++i) {
                const int sl = (int)b.src_lens[i];
                const int ml = with_frames ? (int)std::max<long long>(0, std::min<long long>(b.mel_lens[i], Tcap)) : 0;
                h_src[(size_t)t * cap_B + i] = sl;
                h_flen[(size_t)t * cap_B + i] = ml;
                h_foff[(size_t)t * cap_B + i] = foff;
                if (with_frames) { foff += ml + G; nF += ml; }
                nP += sl;
                for (int s2 = 0; s2 < S; ++s2) {
                    const size_t e = (size_t)t * bs + (size_t)i * S + s2;
                    h_txt[e] = (int)b.texts[(size_t)i * S + s2];
                    if (with_frames) h_dur[e] = (int)std::max<long long>(std::min<long long>(b.durations[(size_t)i * S + s2], 1 << 28), -1);
                    if (!b.pitches.empty() && !cfg.pitch_frame) h_pit[(size_t)t * pe_cap_p + (size_t)i * S + s2] = b.pitches[(size_t)i * S + s2];
                    if (!b.energies.empty() && !cfg.energy_frame) h_ene[(size_t)t * pe_cap_e + (size_t)i * S + s2] = b.energies[(size_t)i * S + s2];
                }
            }
            if (!b.pitches.empty() && cfg.pitch_frame) memcpy(h_pit + (size_t)t * pe_cap_p, b.pitches.data(), b.pitches.size() * sizeof(float));
            if (!b.energies.empty() && cfg.energy_frame) memcpy(h_ene + (size_t)t * pe_cap_e, b.energies.data(), b.energies.size() * sizeof(float));
            const int Mf = with_frames ? foff : 0, Mr = with_frames ? G + B * (Tcap + G) : 0;
            if (Mf > capMf || Mr > capMr || Mp > capMp) { set_error("row space exceeds capacity"); return -1; }
            for (int i = 0; i <= cap_B; ++i) h_spk[(size_t)t * (cap_B + 1) + i] = b.spk_ids[i];
            if (!b.spk_emb.empty()) memcpy((float*)(H + img.spk_emb) + (size_t)t * cap_B * d, b.spk_emb.data(), b.spk_emb.size() * sizeof(float));
            PlanTaskHdr& h = hdr[t];
            h.B = B; h.S = S; h.Tmax_in = b.T_max; h.Tcap = Tcap; h.Mp = Mp; h.Mf = Mf; h.Mr = Mr; h.with_frames = with_frames;
            h.has_targets = !b.pitches.empty(); h.has_mels = b.mels != nullptr; h.pad0 = h.pad1 = 0; h.mel_off = mel_used;
            if (b.mels) {
                const size_t n = (size_t)B * b.T_max * cfg.n_mel;
                memcpy(h_mel + mel_used, b.mels, n * sizeof(float));
                mel_used += (long long)n;
                any_mels = true;
            }
            // attention groups (valid rows only)
            long long soff[2] = {0, 0};
            for (int i = 0; i < B; ++i)
                for (int which = 0; which < (with_frames ? 2 : 1); ++which) {
                    const int Hh = which ? cfg.dec_heads : cfg.enc_heads, dk = d / Hh;
                    const int L = which ? h_flen[(size_t)t * cap_B + i] : h_src[(size_t)t * cap_B + i];
                    const int ro = which ? h_foff[(size_t)t * cap_B + i] : G + i * (S + G);
                    const long long task_s = which ? S_ts_f : S_ts_p;
                    int& maxL = which ? p.dec_maxL : p.enc_maxL;
                    maxL = std::max(maxL, L);
                    (which ? p.sum_attn_f : p.sum_attn_p) += (double)Hh * L * L;
                    (which ? p.sumLf : p.sumLp) += (long long)Hh * L;
                    const int ldS = (L + 3) & ~3;
                    const int capM = which ? capMf : capMp;
                    for (int hh = 0; hh < Hh; ++hh) {
                        const long long so = (long long)t * task_s + soff[which];
                        soff[which] += (long long)L * ldS;
                        const int n = nseq[which]++;
                        seqs[which][n] = AttnSeq{so, L, ldS};
                        const long long qo = (long long)t * ((long long)(capM + 2 * G) * 3 * d) + (long long)ro * 3 * d + hh * dk;
                        const long long oo = (long long)t * ((long long)(capM + 2 * G) * d) + (long long)ro * d + hh * dk;
                        tabs[which][TAB_QK][n] = GemmGroupDesc{qo, qo + d, so, L, L, dk, 0, 0, ldS};          // S  = Q K^T
                        tabs[which][TAB_PV][n] = GemmGroupDesc{so, qo + 2 * d, oo, L, dk, L, ldS, 0, 0};     // O  = P V
                        tabs[which][TAB_DP][n] = GemmGroupDesc{oo, qo + 2 * d, so, L, L, dk, 0, 0, ldS};     // dP = dO V^T
                        tabs[which][TAB_DV][n] = GemmGroupDesc{so, oo, qo + 2 * d, L, dk, L, ldS, 0, 0};     // dV = P^T dO
                        tabs[which][TAB_DQ][n] = GemmGroupDesc{so, qo + d, qo, L, dk, L, ldS, 0, 0};         // dQ = dS K
                        tabs[which][TAB_DK][n] = GemmGroupDesc{so, qo, qo + d, L, dk, L, ldS, 0, 0};         // dK = dS^T Q
                    }
                }
            p.hB[t] = B; p.hSmax[t] = S; p.hTcap[t] = Tcap; p.hMp[t] = Mp; p.hMf[t] = Mf; p.hMr[t] = Mr;
            p.maxMp = std::max(p.maxMp, Mp); p.maxMf = std::max(p.maxMf, Mf); p.maxMr = std::max(p.maxMr, Mr);
            p.maxB = std::max(p.maxB, B);
            int* m = &meta[(size_t)t * META_STRIDE];
            m[META_B] = B; m[META_SMAX] = S; m[META_TCAP] = Tcap; m[META_MP] = Mp; m[META_MF] = Mf; m[META_MR] = Mr;
            m[META_NP] = nP; m[META_NF] = nF;
            p.sum_nP += nP; p.sum_nF += nF;
            p.sumMp += Mp; p.sumMf += Mf; p.sumMr += Mr;
        }
        p.n_enc_groups = nseq[0]; p.n_dec_groups = nseq[1];
        // Attention groups longest first.  A (sequence, head) pair is ~L / 32 workgroups of cost ~L each and a grid is dispatched group by group, so in
        // batch order the last dispatch round of the fused forward (1.1 k workgroups on 512 slots at 8 tasks) and of the backward's grouped GEMMs is
        // whatever pairs happen to come last; sorted, the long pairs start first and the short ones fill the tail.  Every table entry carries its own
        // offsets, so the order is free; the sort is stable (the heads of a sequence stay neighbours).  MTTS_ATTN_SORT=0: batch order.
        static const bool attn_sort = [] { const char* e = getenv("MTTS_ATTN_SORT"); return e ? atoi(e) != 0 : true; }();
        for (int which = 0; attn_sort && which < 2; ++which) {
            const int n = nseq[which];
            if (n < 2) continue;
            std::vector<int> idx((size_t)n);
            for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return seqs[which][a].L > seqs[which][b].L; });
            std::vector<AttnSeq> sq(seqs[which], seqs[which] + n);
            for (int i = 0; i < n; ++i) seqs[which][i] = sq[(size_t)idx[(size_t)i]];
            for (int k = 0; k < 6; ++k) {
                std::vector<GemmGroupDesc> tb(tabs[which][k], tabs[which][k] + n);
                for (int i = 0; i < n; ++i) tabs[which][k][i] = tb[(size_t)idx[(size_t)i]];
            }
        }
        // ONE transfer: everything up to the mel area, plus the used part of the mel area
        const size_t bytes = any_mels ? img.mels + (size_t)mel_used * sizeof(float) : img.mels;
        HIP_CHECK(hipMemcpyAsync(p.img_dev, H, bytes, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipEventRecord(p.img_done[par], stream));
        p.img_pending[par] = true;
        PlanImage im;
        im.hdr = (const PlanTaskHdr*)(p.img_dev + img.hdr);
        im.src_len = (const int*)(p.img_dev + img.src_len); im.flen = (const int*)(p.img_dev + img.flen);
        im.foff = (const int*)(p.img_dev + img.foff); im.spk = (const int*)(p.img_dev + img.spk);
        im.texts = (const int*)(p.img_dev + img.texts); im.dur = (const int*)(p.img_dev + img.dur);
        im.pitch = (const float*)(p.img_dev + img.pitch); im.energy = (const float*)(p.img_dev + img.energy);
        im.mels = (const float*)(p.img_dev + img.mels);
        im.cap_B = cap_B; im.cap_S = cap_S;
        im.pe_cap_p = (long long)pe_cap_p; im.pe_cap_e = (long long)pe_cap_e; im.pitch_frame = cfg.pitch_frame; im.energy_frame = cfg.energy_frame;
        PlanOut o;
        o.r_pitch_t = p.r_pitch_t; o.r_energy_t = p.r_energy_t;
        o.p_row_b = p.p_row_b; o.p_row_t = p.p_row_t; o.p_tok = p.p_tok; o.p_first = p.p_first; o.p_count = p.p_count; o.p_dur = p.p_dur;
        o.p_seg_start = p.p_seg_start; o.p_seg_len = p.p_seg_len; o.p_valid = p.p_valid; o.p_inrect = p.p_inrect;
        o.p_pitch_t = p.p_pitch_t; o.p_energy_t = p.p_energy_t;
        o.f_row_b = p.f_row_b; o.f_row_t = p.f_row_t; o.f_src = p.f_src; o.f_seg_start = p.f_seg_start; o.f_seg_len = p.f_seg_len;
        o.f2r = p.f2r; o.f_valid = p.f_valid; o.r2f = p.r2f; o.r_valid = p.r_valid; o.r_inrect = p.r_inrect; o.mel_tgt = p.mel_tgt;
        o.spk_ids = p.spk_ids;
        o.p_valid_w = p.p_valid_w; o.p_inrect_w = p.p_inrect_w; o.f_valid_w = p.f_valid_w; o.r_valid_w = p.r_valid_w; o.r_inrect_w = p.r_inrect_w;
        o.ts_p = row_ts_p; o.ts_f = row_ts_f; o.ts_r = row_ts_r; o.ts_mel = (long long)(capMr + 2 * G) * cfg.n_mel; o.ts_seg = cap_B; o.ts_spk = cap_B + 1;
        MTTS_LAUNCH(plan_rows_p_kernel, dim3((unsigned)((p.maxMp + 255) / 256), 1, (unsigned)tasks), dim3(256), stream, im, o);
        if (with_frames) {
            MTTS_LAUNCH(plan_rows_f_kernel, dim3((unsigned)((p.maxMf + 255) / 256), 1, (unsigned)tasks), dim3(256), stream, im, o);
            MTTS_LAUNCH(plan_prefix_kernel, dim3(1, 1, (unsigned)tasks), dim3(64), stream, im, o);
            MTTS_LAUNCH(plan_rows_r_kernel, row_grid(p.maxMr, tasks), dim3(256), stream, im, o, cfg.n_mel);
        }
        return 0;
    }

    // free-running: predicted durations (device) -> host, then size and build the frame spaces
    // (modules.py:132-137,167-190: the reference reads every duration with .item(); here one copy per task)

    // =================================================================================
    // pass context + small launch helpers
    // =================================================================================
    struct Pass {
        Plan* pl;
        bool use_fast;   // adapted modules read the per-task fast weights
        bool train;      // BatchNorm batch statistics (+ running update); decoder truncation
        float p_control = 1.f, e_control = 1.f, d_control = 1.f;
        unsigned seed_override = 0;  // != 0: replay this dropout seed (second-order HVP re-runs inner step k)
        bool update_bn = true;  // momentum update of the BatchNorm running buffers (off when a pass is re-run for a HVP)
        TS enc_out{nullptr, 0}; // set: the encoder output of this pass was computed ahead of time (run_encoder_ahead): skip embedding + encoder
    };
    enum Space { SP_P = 0, SP_F = 1, SP_R = 2 };

    TS W(const Pass& ps, long long off) const {
        if (ps.use_fast && off >= adapt_start) return TS{fast_cur + (off - adapt_start), n_adapt};
        return TS{theta + off, 0};
    }
    TS Gd(long long off) const { return TS{grad_dst + off, n_total}; }
    int mfield(Space s) const { return s == SP_P ? META_MP : (s == SP_F ? META_MF : META_MR); }
    int maxM(const Plan& p, Space s) const { return s == SP_P ? p.maxMp : (s == SP_F ? p.maxMf : p.maxMr); }
    // [row][4] float image of one of the plan's byte masks
    const float* mask_w(const Plan& p, const unsigned char* m) const {
        if (m == p.p_valid) return p.p_valid_w;
        if (m == p.p_inrect) return p.p_inrect_w;
        if (m == p.f_valid) return p.f_valid_w;
        if (m == p.r_valid) return p.r_valid_w;
        if (m == p.r_inrect) return p.r_inrect_w;
        return nullptr;
    }
    const unsigned char* valid_mask(const Plan& p, Space s) const { return s == SP_P ? p.p_valid : (s == SP_F ? p.f_valid : p.r_valid); }
    const unsigned char* inrect_mask(const Plan& p, Space s) const { return s == SP_P ? p.p_inrect : (s == SP_F ? p.f_valid : p.r_inrect); }
    long long sumM(const Plan& p, Space s) const { return s == SP_P ? p.sumMp : (s == SP_F ? p.sumMf : p.sumMr); }
    double alg_rows(const Plan& p, Space s) const { return s == SP_P ? p.sum_nP : p.sum_nF; }
    long long row_ts(Space s) const { return s == SP_P ? row_ts_p : (s == SP_F ? row_ts_f : row_ts_r); }

    // dst = dropout(src) with the mask stream (plan seed, site); no-op alias when dropout is off
    TS drop(const Pass& ps, Space s, TS src, TS dst, int C, float prob, int site) {
        if (!drop_active(ps) || prob <= 0.f) return src;
        const Plan& p = *ps.pl;
        const unsigned seed = p.drop_seed * 0x9E3779B1u + (unsigned)site * 0x85EBCA6Bu + 0xC2B2AE35u;
        const unsigned thr = (unsigned)std::lround((double)prob * 65536.0);
        MTTS_LAUNCH(dropout_kernel, row_grid(maxM(p, s), p.tasks), dim3(256), stream, (const int*)p.meta, mfield(s), (const float*)src.p,
                    src.ts, dst.p, dst.ts, C, seed, thr, 1.f / (1.f - prob));
        return dst;
    }
    float block_dropout(Space s) const { return s == SP_P ? cfg.enc_dropout : cfg.dec_dropout; }
    // the same mask stream as drop(), for kernels that apply the dropout in passing (LayerNorm fwd / bwd)
    DropSpec drop_spec(const Pass& ps, float prob, int site) const {
        DropSpec d;
        if (!drop_active(ps) || prob <= 0.f) return d;
        d.seed = ps.pl->drop_seed * 0x9E3779B1u + (unsigned)site * 0x85EBCA6Bu + 0xC2B2AE35u;
        d.thr16 = (unsigned)std::lround((double)prob * 65536.0);
        d.scale = 1.f / (1.f - prob);
        return d;
    }

    // row-space GEMM over all tasks: C[M,N] (+)= op(A, B); M (or the reduction length for TN) is
    // the task's row count
    GemmArgs rowgemm(const Plan& p, Space s, int form) const {
        GemmArgs g;
        g.dimptr = p.meta + mfield(s);
        g.dim_stride = META_STRIDE;
        g.dim_sel = (form == GEMM_TN) ? 2 : 0;
        g.host_dims = (s == SP_P) ? p.hMp.data() : (s == SP_F) ? p.hMf.data() : p.hMr.data();   // (8-task launches: task-per-XCD schedule, gemm.h)
        return g;
    }

    // Y[M,N] = conv_k(X)[M, k*Cin] * W[N][k*Cin]^T + b   (k = 1: Linear)
    // (x2, w2): second source of a dual-source launch, y = conv(x; w) + conv(x2; w2) (GemmArgs::A2 — the tangent pairs of engine_so.inc)
    // x_plane: x's operand plane is current (its producer wrote it); y_twin: write y's plane in the epilogue (bf16 mode, see H())
    // ln (optional): row-complete epilogue — bias + dropout + residual + LayerNorm behind the GEMM (gemm.h: LnFuse; ln_fused_fwd below)
    void conv_fwd(const Pass& ps, Space s, TS x, int cin, int k, TS w, TS b, int cout, TS y, int flags,
                  const unsigned char* rowmask, TS relu_ref = TS{nullptr, 0}, TS x2 = TS{nullptr, 0}, TS w2 = TS{nullptr, 0},
                  bool x_plane = false, bool y_twin = false, const LnFuse* ln = nullptr) {
        const Plan& p = *ps.pl;
        GemmArgs g = rowgemm(p, s, GEMM_NT);
        const int pad = k / 2;
        const double nsrc = x2.p ? 2.0 : 1.0;
        g.A = x.p - (long long)pad * cin; g.a_gs = x.ts; g.lda = cin;
        g.B = w.p; g.b_gs = w.ts; g.ldb = k * cin;
        if (x2.p) { g.A2 = x2.p - (long long)pad * cin; g.a2_gs = x2.ts; g.B2 = w2.p; g.b2_gs = w2.ts; }
        g.C = y.p; g.c_gs = y.ts; g.ldc = cout;
        g.N = cout; g.K = k * cin;
        g.bias = b.p; g.bias_gs = b.ts;
        g.flags = flags;
        g.rowmask = rowmask; g.rowmask_gs = row_ts(s);
        if (relu_ref.p) { g.relu_ref = relu_ref.p; g.relu_ref_gs = relu_ref.ts; g.ld_relu = cout; }
        if (!x2.p) {   // bf16 mode: the operands' planes (the weight's shadow + the input slab's twin), where both exist
            const HP wh = Wh(ps, w, false);
            if (wh.p) {
                const bf16_t* xh = x_plane ? H(x.p) : make_plane(x, cin, p.tasks);
                if (xh) { g.Ah = xh - (long long)pad * cin; g.Bh = wh.p; g.bh_gs = wh.ts; await_shadows(); }
            }
        }
        if (y_twin) g.Ch = H(y.p);
        if (ln) g.ln = *ln;
        gemm_launch(gx, GEMM_NT, g, maxM(p, s), cout, p.tasks, stream, 0, nsrc * 2.0 * alg_rows(p, s) * cout * k * cin, sumM(p, s),
                    4.0 * (alg_rows(p, s) * (nsrc * cin + cout) + nsrc * (double)p.tasks * cout * k * cin));
    }
    // sublayer GEMM + `LayerNorm(dropout(.) + residual)` (SubLayers.py:54-55,90-91): the GEMM followed by layernorm_fwd_kernel — or, with
    // MTTS_LN_FUSE=1, ONE launch when the GEMM can carry the row-complete epilogue (fp32 mode, C <= 256, a launcher context with counters).
    // Built and measured in round 5, bit-identical results, but SLOWER: 8-task step 157.3 -> 158.8 ms, single-task rank
    // 32.5 -> 34.2 ms, second order 81.4 -> 83.1 ms (profiles/r05_ab_log.md) — the write-through stores, the drain + counter rendezvous and the
    // one-workgroup-per-m-tile tail cost more than the 5-17 us launch they replace — so it stays opt-in.
    // z receives the sublayer output a, then (in place) a' = dropout(a) + res — what the backward keeps; y the normalised rows.
    void ln_fused_fwd(const Pass& ps, Space s, TS x, int cin, int k, TS w, TS b, TS z, TS res, long long g_off, long long b_off,
                      const unsigned char* mask, TS y, TS st, int C, DropSpec din, bool x_plane, bool y_twin) {
        static const bool fuse_on = [] { const char* e = getenv("MTTS_LN_FUSE"); return e ? atoi(e) != 0 : false; }();
        const Plan& p = *ps.pl;
        GemmArgs probe;
        probe.N = C;
        if (fuse_on && !ablate_ln() && !gx.batch.open && gemm_ln_fusable(gx, GEMM_NT, probe, maxM(p, s), p.tasks)) {
            LnFuse f;
            TS gm = W(ps, g_off), bt = W(ps, b_off);
            f.res = res.p; f.res_gs = res.ts;
            f.gamma = gm.p; f.beta = bt.p; f.par_gs = gm.ts;
            f.mask = mask; f.mask_gs = row_ts(s);
            f.y = y.p; f.y_gs = y.ts;
            f.stats = st.p; f.st_gs = st.ts;
            f.drop_seed = din.seed; f.drop_thr16 = din.thr16; f.drop_scale = din.scale;
            conv_fwd(ps, s, x, cin, k, w, b, C, z, 0, nullptr, TS{nullptr, 0}, TS{nullptr, 0}, TS{nullptr, 0}, x_plane, false, &f);
            return;
        }
        conv_fwd(ps, s, x, cin, k, w, b, C, z, 0, nullptr, TS{nullptr, 0}, TS{nullptr, 0}, TS{nullptr, 0}, x_plane, false);
        ln_fwd(ps, s, z, res, g_off, b_off, mask, z, y, st, C, din, DropSpec(), y_twin);
    }
    // dX[M,Cin] (+)= sum_taps dY[M +- tap, Cout] * W  (conv dgrad over the same [Cout][k][Cin] image)
    void conv_dgrad(const Pass& ps, Space s, TS dy, int cout, int k, TS w, int cin, TS dx, int flags,
                    const unsigned char* rowmask, TS relu_ref = TS{nullptr, 0}, TS dy2 = TS{nullptr, 0}, TS w2 = TS{nullptr, 0},
                    bool dy_plane = false, bool dx_twin = false) {
        const Plan& p = *ps.pl;
        const int pad = k / 2;
        if (!dy2.p) {
            // bf16 mode with planes: dX = conv(dY; flipped transposed W) as an NT problem over the weight's input-gradient shadow — the
            // forward conv's kernel path with Cin and Cout exchanged (weight_shadow_kernel)
            const HP wt = Wh(ps, w, true);
            const bf16_t* dyh = !wt.p ? nullptr : (dy_plane ? H(dy.p) : make_plane(dy, cout, p.tasks));
            if (dyh) {
                await_shadows();
                GemmArgs g = rowgemm(p, s, GEMM_NT);
                g.A = dy.p - (long long)pad * cout; g.a_gs = dy.ts; g.lda = cout;
                g.Ah = dyh - (long long)pad * cout;
                g.B = nullptr; g.b_gs = 0; g.ldb = k * cout; g.Bh = wt.p; g.bh_gs = wt.ts; g.plane_only = true;
                g.C = dx.p; g.c_gs = dx.ts; g.ldc = cin;
                if (dx_twin) g.Ch = H(dx.p);
                g.N = cin; g.K = k * cout;
                g.flags = flags;
                g.rowmask = rowmask; g.rowmask_gs = row_ts(s);
                if (relu_ref.p) { g.relu_ref = relu_ref.p; g.relu_ref_gs = relu_ref.ts; g.ld_relu = cin; }
                gemm_launch(gx, GEMM_NT, g, maxM(p, s), cin, p.tasks, stream, 0, 2.0 * alg_rows(p, s) * cout * k * cin, sumM(p, s),
                            4.0 * (alg_rows(p, s) * (cin + cout) + (double)p.tasks * cout * k * cin));
                return;
            }
        }
        GemmArgs g = rowgemm(p, s, GEMM_NN);
        const double nsrc = dy2.p ? 2.0 : 1.0;
        g.A = dy.p - (long long)pad * cout; g.a_gs = dy.ts; g.lda = cout;
        g.B = w.p; g.b_gs = w.ts; g.ldb = k * cin;
        if (dy2.p) { g.A2 = dy2.p - (long long)pad * cout; g.a2_gs = dy2.ts; g.B2 = w2.p; g.b2_gs = w2.ts; }
        g.C = dx.p; g.c_gs = dx.ts; g.ldc = cin;
        if (dx_twin) g.Ch = H(dx.p);
        g.N = cin; g.K = k * cout;
        g.taps = k; g.tap_k = cout; g.tap_bstride = cin;
        g.flags = flags;
        g.rowmask = rowmask; g.rowmask_gs = row_ts(s);
        if (relu_ref.p) { g.relu_ref = relu_ref.p; g.relu_ref_gs = relu_ref.ts; g.ld_relu = cin; }
        gemm_launch(gx, GEMM_NN, g, maxM(p, s), cin, p.tasks, stream, 0, nsrc * 2.0 * alg_rows(p, s) * cout * k * cin, sumM(p, s),
                    4.0 * (alg_rows(p, s) * (cin + nsrc * cout) + nsrc * (double)p.tasks * cout * k * cin));
    }
    // dW[Cout][k*Cin] = dY^T * conv_k(X), db = colsum(dY)
    // cx / st: launch context (default: the engine's own context and stream; the deferred path passes the side stream's)
    void conv_wgrad(const Pass& ps, Space s, TS dy, int cout, int k, TS x, int cin, long long w_off, long long b_off,
                    const unsigned char* bias_mask, int flags = 0, GemmCtx* cx = nullptr, hipStream_t st = nullptr,
                    TS dy2 = TS{nullptr, 0}, TS x2 = TS{nullptr, 0}) {
        const Plan& p = *ps.pl;
        GemmCtx& gcx = cx ? *cx : gx;
        const hipStream_t gst = cx ? st : stream;
        GemmArgs g = rowgemm(p, s, GEMM_TN);
        const int pad = k / 2;
        g.A = dy.p; g.a_gs = dy.ts; g.lda = cout;
        g.B = x.p - (long long)pad * cin; g.b_gs = x.ts; g.ldb = cin;
        const double nsrc = dy2.p ? 2.0 : 1.0;
        if (dy2.p) { g.A2 = dy2.p; g.a2_gs = dy2.ts; g.B2 = x2.p - (long long)pad * cin; g.b2_gs = x2.ts; }   // dW = dy^T x + dy2^T x2 (the bias sum: dy only)
        TS gw = Gd(w_off);
        g.C = gw.p; g.c_gs = gw.ts; g.ldc = k * cin;
        g.M = cout; g.N = k * cin;
        g.K = maxM(p, s);  // upper bound of the per-task reduction length (the kernel reads the exact one through dimptr)
        g.flags = flags;
        // the bias gradient rides on the weight-gradient GEMM (GemmArgs::colsum: one extra n-tile against the mask's float image)
        // instead of two reduction launches per layer
        const float* mw = mask_w(p, bias_mask);
        const bool fused = b_off >= 0 && mw != nullptr;
        if (fused) {
            TS gb = Gd(b_off);
            g.colsum = gb.p; g.colsum_gs = gb.ts; g.colsum_w = mw; g.colsum_w_gs = 4 * row_ts(s);
        }
        gemm_launch(gcx, GEMM_TN, g, cout, k * cin, p.tasks, gst, 0, nsrc * 2.0 * alg_rows(p, s) * cout * k * cin, (long long)cout * p.tasks,
                    4.0 * (nsrc * alg_rows(p, s) * (cin + cout) + (double)p.tasks * cout * k * cin));
        if (b_off >= 0 && !fused) colsum(ps, s, dy, cout, bias_mask, TS{nullptr, 0}, Gd(b_off));   // (main stream: never reached on the deferred path)
    }
    // two-stage deterministic column reduction (rowops.h colpart/colfinal)
    void colreduce(const Plan& p, ColArgs a, float* out0, float* out1, long long out_ts, int maxM, bool on_side = false) {
        if (on_side) {   // the deferred path's reductions: side stream, its own partial buffer, plain two-stage form
            const int chunks_s = (maxM + kRC - 1) / kRC;
            MTTS_LAUNCH(colpart_kernel, dim3((a.C + 127) / 128, chunks_s, p.tasks), dim3(256), side, (const int*)p.meta, a, col_partial_side, col_max_chunks);
            MTTS_LAUNCH(colfinal_kernel, dim3(colfinal_blocks(a.C), 1, p.tasks), dim3(256), side, (const int*)p.meta, a.mfield, a.mode,
                        (const float*)col_partial_side, col_max_chunks, a.C, out0, out1, out_ts, 1e-5f, a.accumulate, (int)kRC);
            return;
        }
        // (single-launch variants were built and measured slower in rounds 2-3: a workgroup per 32-column stripe walking all rows is latency-
        // bound, +10 % on the 8-task step; a last-arriver fold needs an agent-scope release / acquire amid the GEMMs' dirty L2 lines, +10 %)
        const int chunks = (maxM + kRC - 1) / kRC;
        MTTS_LAUNCH(colpart_kernel, dim3((a.C + 127) / 128, chunks, p.tasks), dim3(256), stream, (const int*)p.meta, a, col_partial,
                    col_max_chunks);
        MTTS_LAUNCH(colfinal_kernel, dim3(colfinal_blocks(a.C), 1, p.tasks), dim3(256), stream, (const int*)p.meta, a.mfield, a.mode,
                    (const float*)col_partial, col_max_chunks, a.C, out0, out1, out_ts, 1e-5f, a.accumulate, (int)kRC);
    }
    void colsum(const Pass& ps, Space s, TS x, int C, const unsigned char* mask, TS roww, TS out, bool on_side = false) {
        const Plan& p = *ps.pl;
        ColArgs a;
        a.X = x.p; a.x_ts = x.ts; a.mask = mask; a.mask_ts = row_ts(s); a.roww = roww.p; a.roww_ts = roww.ts;
        a.C = C; a.mode = 0; a.mfield = mfield(s);
        colreduce(p, a, out.p, nullptr, out.ts, maxM(p, s), on_side);
    }
    // MEASUREMENT ONLY (results are wrong): MTTS_ABLATE_LN=1 drops every LayerNorm forward / backward launch of the FFT blocks and predictors, which
    // bounds from above what folding bias + dropout + residual + LayerNorm into the producing GEMM's epilogue (and the LayerNorm backward into
    // the consuming GEMM's prologue) could save — a timing run, never a result (profiles/r05_ab_log.md; bench.py refuses to gate parity on it)
    // The switch only exists in a library built with -DMTTS_ABLATE (a diagnostic build: `hipcc ... -DMTTS_ABLATE`); the shipped libmtts.so has no
    // way to skip a LayerNorm, whatever its environment holds.
#if defined(MTTS_ABLATE)
    static bool ablate_ln() {
        static const bool on = [] { const char* e = getenv("MTTS_ABLATE_LN"); return e && atoi(e) != 0; }();
        return on;
    }
#else
    static constexpr bool ablate_ln() { return false; }
#endif
    void ln_fwd(const Pass& ps, Space s, TS a, TS res, long long g_off, long long b_off, const unsigned char* mask,
                TS zout, TS y, TS st, int C, DropSpec din = DropSpec(), DropSpec dout = DropSpec(), bool y_twin = false) {
        // y_twin: bf16 mode — also write y's operand plane (H(y)): the next conv reads it instead of a conversion pass
        if (ablate_ln()) return;
        const Plan& p = *ps.pl;
        TS gm = W(ps, g_off), bt = W(ps, b_off);
        MTTS_LAUNCH_LN(layernorm_fwd_kernel, C, row2_grid(maxM(p, s), p.tasks), dim3(256), stream, (const int*)p.meta, mfield(s),
                    (const float*)a.p, a.ts, (const float*)res.p, res.ts, (const float*)gm.p, (const float*)bt.p, gm.ts, mask,
                    row_ts(s), zout.p, zout.ts, y.p, y.ts, st.p, st.ts, C, 1e-5f, din, dout, y_twin ? H(y.p) : (bf16_t*)nullptr);
    }
    // dz = LayerNorm backward (masked); parameter grads into the per-task grad buffer
    // dz_drop: second output = dropout(dz) with the forward site's mask; copy_always: written even when dropout is off (a plain copy)
    // din: dropout applied to dy on load (a dropout that sits BEHIND this LayerNorm in the forward, modules.py:222-235)
    // part: null = the gamma / beta reduction completes here (the kernel's 8-row partials in the shared scratch + one colfinal launch on this
    //       stream); set = the partials are left in `part` ([tasks][ln_chunks][3][C], a buffer of the site's own) and ln_param_grads_side
    //       folds them later on the side stream (deferred parameter gradients)
    void ln_bwd(const Pass& ps, Space s, TS dy, TS z, TS st, long long g_off, long long b_off, const unsigned char* mask,
                TS dz, int C, int relu_on_z, TS dz_drop = TS{nullptr, 0}, DropSpec dd = DropSpec(), bool copy_always = false,
                float* part = nullptr, DropSpec din = DropSpec(), int twin_sel = 0) {
        // twin_sel: bf16 mode — also write the operand plane of dz (1) or of dz_drop (2)
        if (ablate_ln()) return;
        const Plan& p = *ps.pl;
        TS gm = W(ps, g_off);
        const int chunks = ln_chunks(maxM(p, s));
        float* pbuf = part ? part : col_partial;
        bf16_t* twin = twin_sel == 1 ? H(dz.p) : (twin_sel == 2 ? H(dz_drop.p) : nullptr);
        if (!twin) twin_sel = 0;
        MTTS_LAUNCH_LN(layernorm_bwd_kernel, C, row2_grid(maxM(p, s), p.tasks), dim3(256), stream, (const int*)p.meta, mfield(s),
                    (const float*)dy.p, dy.ts, (const float*)z.p, z.ts, (const float*)st.p, st.ts, (const float*)gm.p, gm.ts,
                    mask, row_ts(s), dz.p, dz.ts, C, relu_on_z, (dd.thr16 || copy_always) ? dz_drop.p : nullptr, dz_drop.ts, dd,
                    din, pbuf, chunks, twin, twin_sel);
        if (!part) ln_fold(p, s, pbuf, chunks, g_off, b_off, C, stream);
    }
    // stage 2 of a LayerNorm's gamma / beta reduction: fold the backward kernel's partial rows
    void ln_fold(const Plan& p, Space s, const float* part, int chunks, long long g_off, long long b_off, int C, hipStream_t st) {
        TS gg = Gd(g_off), gb = Gd(b_off);
        MTTS_LAUNCH(colfinal_kernel, dim3(colfinal_blocks(C), 1, p.tasks), dim3(256), st, (const int*)p.meta, mfield(s), 1, part, chunks, C,
                    gg.p, gb.p, gg.ts, 1e-5f, 0, (int)kLnRows);
    }
    // the LayerNorm gamma / beta gradients of a deferred layer, from the partials its backward kernel left, on the side stream
    void ln_param_grads_side(const Pass& ps, Space s, const float* part, long long g_off, long long b_off, int C) {
        const Plan& p = *ps.pl;
        ln_fold(p, s, part, ln_chunks(maxM(p, s)), g_off, b_off, C, side);
    }
    void attn_gemm(const Pass& ps, Space s, int which, int form, const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, float alpha, int heads, int flags = 0, const float* A2 = nullptr, const float* B2 = nullptr) {
        const Plan& p = *ps.pl;
        GemmArgs g;
        g.table = (s == SP_P) ? p.enc_tab[which] : p.dec_tab[which];
        g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.flags = flags;
        g.A2 = A2; g.B2 = B2;   // dual source: the same per-(sequence, head) offsets into a second pair of buffers
        const double nsrc = A2 ? 2.0 : 1.0;
        const int L = (s == SP_P) ? p.enc_maxL : p.dec_maxL, dk = cfg.d_model / heads;
        const int groups = (s == SP_P) ? p.n_enc_groups : p.n_dec_groups;
        int mM = L, mN = L;
        if (which == TAB_PV || which == TAB_DV || which == TAB_DQ || which == TAB_DK) mN = dk;
        g.K = (which == TAB_QK || which == TAB_DP) ? dk : L;  // cost hint for launch batching (the table carries the real K)
        // per (sequence, head): two L x dk operands and one L x L matrix, each moved once
        const double sumL2 = (s == SP_P) ? p.sum_attn_p : p.sum_attn_f, sumL = (double)((s == SP_P) ? p.sumLp : p.sumLf);
        gemm_launch(gx, form, g, mM, mN, groups, stream, 0, nsrc * 2.0 * sumL2 * dk, (s == SP_P) ? p.sumLp : p.sumLf,
                    4.0 * nsrc * (sumL2 + 2.0 * sumL * dk));
    }

    // =================================================================================
    // FFT block (transformer/Layers.py:21-30)
    // =================================================================================
    void fft_fwd(const Pass& ps, Space s, int heads, const FFTP& P, LayerBuf& b, TS xin, TS S_unused) {
        TagScope tag_scope(*this, s == SP_P ? 1 : 2);
        (void)S_unused;
        const Plan& p = *ps.pl;
        const int d = cfg.d_model, dk = d / heads;
        const unsigned char* vm = valid_mask(p, s);
        const unsigned char* im = inrect_mask(p, s);
        conv_fwd(ps, s, xin, d, 1, W(ps, P.wqkv), W(ps, P.bqkv), 3 * d, b.qkv, 0, nullptr);
        const int groups = (s == SP_P) ? p.n_enc_groups : p.n_dec_groups;
        const int L = (s == SP_P) ? p.enc_maxL : p.dec_maxL;
        const AttnSeq* seqs = (s == SP_P) ? p.enc_seqs : p.dec_seqs;
        // scaled-dot-product attention (Modules.py:14-25): ONE fused launch (attention.h) — the score tile of 32 query rows lives in LDS
        // between Q K^T, the softmax and P V; the probabilities go to HBM once, for the backward.  MTTS_FUSED_ATTN=0 (A/B runs), the bf16
        // numerics mode (measured on C2: 7.28 ms with this fp32 kernel, 7.15 ms with bf16 grouped GEMMs around the softmax kernel) and
        // sequences beyond 1024 keys take the three-launch form: grouped GEMM, softmax kernel, grouped GEMM.
        static const bool fused_attn = [] { const char* e = getenv("MTTS_FUSED_ATTN"); return e ? atoi(e) != 0 : true; }();
        if (fused_attn && !gx.bf16 && attn_fused_ok(L, dk) && groups > 0) {
            AttnFwdArgs fa;
            fa.seqs = seqs;
            fa.tab_qk = (s == SP_P) ? p.enc_tab[TAB_QK] : p.dec_tab[TAB_QK];
            fa.tab_pv = (s == SP_P) ? p.enc_tab[TAB_PV] : p.dec_tab[TAB_PV];
            fa.Q = fa.K = fa.V = b.qkv.p; fa.ld_q = fa.ld_k = fa.ld_v = 3 * d;
            fa.P = b.P.p; fa.O = b.O.p; fa.ld_o = d;
            fa.scale = 1.f / sqrtf((float)dk); fa.dk = dk; fa.rot = attn_rot_default(); fa.prio = gx.wave_prio;
#if defined(MTTS_ATTN_DIAG)
            fa.diag = 0;
#endif
            GemmProfiler& prof = gx.prof;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prof.enabled) { e0 = prof.get(); e1 = prof.get(); hipEventRecord(e0, stream); }
            attn_fwd_launch(fa, L, groups, stream);
            if (prof.enabled) {
                hipEventRecord(e1, stream);
                const double sumL2 = (s == SP_P) ? p.sum_attn_p : p.sum_attn_f, sumL = (double)((s == SP_P) ? p.sumLp : p.sumLf);
                GemmProfiler::Rec rec{GK_ATTN_FWD, 2.0 * 2.0 * sumL2 * dk, e0, e1};   // both products
                rec.form = 4; rec.tile = 32; rec.N = dk; rec.K = L; rec.groups = groups; rec.rows = sumL;
                rec.bytes = 4.0 * (sumL2 + 4.0 * sumL * dk);                          // Q, K, V read, O and P written once
                rec.tag = prof.tag;
                prof.recs.push_back(rec);
            }
        } else {
            attn_gemm(ps, s, TAB_QK, GEMM_NT, b.qkv.p, 3 * d, b.qkv.p, 3 * d, b.P.p, 0, 1.f / sqrtf((float)dk), heads);
            if (groups > 0 && L > 0)
                MTTS_LAUNCH(softmax_fwd_kernel, dim3((L + 3) / 4, 1, groups), dim3(256), stream, seqs, b.P.p);
            attn_gemm(ps, s, TAB_PV, GEMM_NN, b.P.p, 0, b.qkv.p, 3 * d, b.O.p, d, 1.f, heads);
        }
        // self.dropout(self.fc(output)) + residual -> LayerNorm (SubLayers.py:54-55): in the fc GEMM's launch (row-complete epilogue), or
        // the dropout riding in the LayerNorm kernel (bf16 mode: y1's and h's operand planes are written by their producers — the LayerNorm
        // kernel, conv1's epilogue)
        ln_fused_fwd(ps, s, b.O, d, 1, W(ps, P.wfc), W(ps, P.bfc), b.z1, xin, P.ln1g, P.ln1b, vm, b.y1, b.st1, d,
                     drop_spec(ps, block_dropout(s), site_base), false, true);
        conv_fwd(ps, s, b.y1, d, cfg.k1, W(ps, P.w1), W(ps, P.b1), cfg.d_ff, b.h, GEMM_RELU, im, TS{nullptr, 0}, TS{nullptr, 0}, TS{nullptr, 0}, true, true);
        // self.dropout(output) + residual -> LayerNorm (SubLayers.py:90-91)
        ln_fused_fwd(ps, s, b.h, cfg.d_ff, cfg.k2, W(ps, P.w2), W(ps, P.b2), b.z2, b.y1, P.ln2g, P.ln2b, vm, b.y2, b.st2, d,
                     drop_spec(ps, block_dropout(s), site_base + 1), true, false);
    }

    // g0_in holds dL/dy2; K.g0 receives dL/dx, K.gh / K.dy1 / K.dO / K.gqkv the gradients in between (LayerKeep: by default aliases of
    // the shared scratch, with g0_in == K.g0 and K.dy1 == K.dO); dS is scratch
    // lg != null: deferred weight gradients (see LayerGrad) — the four gradients the layer's weight-gradient GEMMs read go to
    // lg's buffers, the GEMMs themselves to the side stream
    // fold_part != null (and lg == null): the two LayerNorms' gamma / beta partials stay in fold_part[0] (LN2) / fold_part[1] (LN1) and are
    // folded on the side stream behind this layer's backward instead of by two launches inside the critical chain
    void fft_bwd(const Pass& ps, Space s, int heads, const FFTP& P, LayerBuf& b, TS xin, TS g0_in, const LayerKeep& K,
                 TS dS, LayerGrad* lg = nullptr, float* const* fold_part = nullptr) {
        TagScope tag_scope(*this, s == SP_P ? 1 : 2);
        const Plan& p = *ps.pl;
        const int d = cfg.d_model, dk = d / heads, ff = cfg.d_ff;
        const unsigned char* vm = valid_mask(p, s);
        const unsigned char* im = inrect_mask(p, s);
        const bool df = lg != nullptr;
        TS g0 = K.g0, g1 = K.dy1, gqkv = K.gqkv, gh = K.gh;
        if (df && K.dy1.p == K.dO.p) { gh = lg->gh; gqkv = lg->gqkv; }   // (a set of its own already has a buffer per layer)
        // LN2 (+ row mask) backward -> g1 = dz2
        TS gm = (s == SP_P) ? gPm : gFm;                                   // masked copy feeds the conv branch, g1 the residual
        const DropSpec dd2 = drop_spec(ps, block_dropout(s), site_base + 1);
        // (bf16 mode: the plane of dc — whichever of the kernel's two outputs that is — and, below, gh's by conv2's epilogue)
        float* const fp2 = (!df && fold_part && side) ? fold_part[0] : nullptr;
        float* const fp1 = (!df && fold_part && side) ? fold_part[1] : nullptr;
        ln_bwd(ps, s, g0_in, b.z2, b.st2, P.ln2g, P.ln2b, vm, g1, d, 0, df ? lg->dc : gm, dd2, df, df ? lg->part2 : fp2, DropSpec(),
               (df || dd2.thr16) ? 2 : 1);
        TS dc = df ? lg->dc : (dd2.thr16 ? gm : g1);
        // conv2
        {   // wgrad and dgrad of a layer are independent: one multi-problem launch (gemm.h: gemm_f32_multi_kernel)
            GemmBatchScope pair(gx, stream);
            if (!df) conv_wgrad(ps, s, dc, d, cfg.k2, b.h, ff, P.w2, P.b2, vm);
            conv_dgrad(ps, s, dc, d, cfg.k2, W(ps, P.w2), ff, gh, 0, im, b.h, TS{nullptr, 0}, TS{nullptr, 0}, true, true);
        }
        // conv1: g1 += dgrad -> dy1
        {
            GemmBatchScope pair(gx, stream);
            if (!df) conv_wgrad(ps, s, gh, ff, cfg.k1, b.y1, d, P.w1, P.b1, im);
            conv_dgrad(ps, s, gh, ff, cfg.k1, W(ps, P.w1), d, g1, GEMM_ACCUM, im, TS{nullptr, 0}, TS{nullptr, 0}, TS{nullptr, 0}, true, false);
        }
        // LN1 backward -> g0 = dz1
        const DropSpec dd1 = drop_spec(ps, block_dropout(s), site_base);
        ln_bwd(ps, s, g1, b.z1, b.st1, P.ln1g, P.ln1b, vm, g0, d, 0, df ? lg->da : gm, dd1, df, df ? lg->part1 : fp1);
        TS da = df ? lg->da : (dd1.thr16 ? gm : g0);
        // fc
        {
            GemmBatchScope pair(gx, stream);
            if (!df) conv_wgrad(ps, s, da, d, 1, b.O, d, P.wfc, P.bfc, vm);
            conv_dgrad(ps, s, da, d, 1, W(ps, P.wfc), d, K.dO, 0, nullptr);
        }
        g1 = K.dO;
        // attention
        const int groups = (s == SP_P) ? p.n_enc_groups : p.n_dec_groups;
        const int L = (s == SP_P) ? p.enc_maxL : p.dec_maxL;
        const AttnSeq* seqs = (s == SP_P) ? p.enc_seqs : p.dec_seqs;
        {
            GemmBatchScope pair(gx, stream);
            attn_gemm(ps, s, TAB_DP, GEMM_NT, g1.p, d, b.qkv.p, 3 * d, dS.p, 0, 1.f, heads);
            attn_gemm(ps, s, TAB_DV, GEMM_TN, b.P.p, 0, g1.p, d, gqkv.p, 3 * d, 1.f, heads);
        }
        if (groups > 0 && L > 0)
            MTTS_LAUNCH(softmax_bwd_kernel, dim3((L + 3) / 4, 1, groups), dim3(256), stream, seqs, (const float*)b.P.p, dS.p,
                        1.f / sqrtf((float)dk));
        {
            GemmBatchScope pair(gx, stream);
            attn_gemm(ps, s, TAB_DQ, GEMM_NN, dS.p, 0, b.qkv.p, 3 * d, gqkv.p, 3 * d, 1.f, heads);
            attn_gemm(ps, s, TAB_DK, GEMM_TN, dS.p, 0, b.qkv.p, 3 * d, gqkv.p, 3 * d, 1.f, heads);
        }
        // fused q/k/v projection
        {
            GemmBatchScope pair(gx, stream);
            if (!df) conv_wgrad(ps, s, gqkv, 3 * d, 1, xin, d, P.wqkv, P.bqkv, vm);
            conv_dgrad(ps, s, gqkv, 3 * d, 1, W(ps, P.wqkv), d, g0, GEMM_ACCUM, nullptr);
        }
        if (df) {
            // the layer's four weight (+ bias) gradients: one multi-problem launch on the side stream, after everything above
            fork_side();
            {
                GemmBatchScope batch(gx_side, side);
                conv_wgrad(ps, s, gh, ff, cfg.k1, b.y1, d, P.w1, P.b1, im, 0, &gx_side, side);
                conv_wgrad(ps, s, dc, d, cfg.k2, b.h, ff, P.w2, P.b2, vm, 0, &gx_side, side);
                conv_wgrad(ps, s, gqkv, 3 * d, 1, xin, d, P.wqkv, P.bqkv, vm, 0, &gx_side, side);
                conv_wgrad(ps, s, da, d, 1, b.O, d, P.wfc, P.bfc, vm, 0, &gx_side, side);
            }
            ln_param_grads_side(ps, s, lg->part2, P.ln2g, P.ln2b, d);
            ln_param_grads_side(ps, s, lg->part1, P.ln1g, P.ln1b, d);
            defer_live = true;
        } else if (fp2) {
            fork_side();
            ln_param_grads_side(ps, s, fp2, P.ln2g, P.ln2b, d);
            ln_param_grads_side(ps, s, fp1, P.ln1g, P.ln1b, d);
        }
    }
    // weight gradients may be deferred to the side stream for this plan (under-filled launches; the bias
    // gradient rides on the GEMM — the separate column reduction would run on the main stream)
    bool defer_ok(const Plan& p) const {
        static const int on = [] { const char* e = getenv("MTTS_DEFER_WGRAD"); return e ? atoi(e) : 1; }();
        static const long long max_rows = [] { const char* e = getenv("MTTS_DEFER_MAX_ROWS"); return e ? atoll(e) : 16000LL; }();   // beyond this the launches fill the chip and the wgrad + dgrad pairing wins (measured: 4 / 8 tasks per rank neutral / -1 %)
        return on && defer_tasks > 0 && p.tasks <= defer_tasks && p.sumMf <= max_rows && side != nullptr;
    }
    // call-site class of the GEMM launches issued from here on (the profiler's per-launch records carry it: GemmProfiler::tag)
    void set_tag(int t) { gx.prof.tag = gx_side.prof.tag = gx_side2.prof.tag = t; }
    struct TagScope {
        Engine& e; int prev;
        TagScope(Engine& e_, int t) : e(e_), prev(e_.gx.prof.tag) { e.set_tag(t); }
        ~TagScope() { e.set_tag(prev); }
    };
    // the variance predictors on the side stream (forward: beside the decoder; backward: under the PostNet / decoder backward) — needs no
    // deferred-gradient buffer, so it also serves launches of more tasks than the deferred regime takes
    bool side_pred_ok(const Plan& p) const {
        static const int all = [] { const char* e = getenv("MTTS_SIDE_PRED_ALL"); return e ? atoi(e) : 1; }();
        return side != nullptr && arena_pred != nullptr && p.tasks <= cap_tasks && (all || defer_ok(p));
    }
    // ---- bucketed exchange, overlapped with the backward that produces the outer gradient (main.py:30-38: DDP's bucketed gradient
    // all-reduce) ------------------------------------------------------------------------------------------------
    // The flat buffer is cut at module boundaries into buckets in backward-COMPLETION order — PostNet, decoder L-1 (+ mel_linear) ... decoder 0,
    // variance adaptor, speaker table, encoder L-1 ... encoder 0 (+ word embedding) — and the moment a module's parameter gradients are
    // complete (ar_ready, called from backward_impl / backward_t) its bucket is summed over the rank's tasks and all-reduced on `comm_stream`
    // behind events of the main and the weight-gradient side stream, while the main stream carries on with the next module's backward.
    // The exchange tail (loss scalars, BatchNorm buffers) is final before the backward starts and goes first.  mtts_allreduce_outer then only
    // joins the comm stream.  Armed per gradient call by the host (mtts_arm_allreduce_overlap); results are those of the one-shot exchange
    // (the same floats are summed by the same collective, in pieces).
    struct ArHook { void* ctx = nullptr; int (*sum)(void*, float*, size_t, hipStream_t) = nullptr; int rank = 0, world = 1; };
    ArHook ar;
    hipStream_t comm_stream = nullptr;
    static constexpr int kArEvents = 64;   // more than one exchange records (2 per bucket + the tail: 27 at base.yaml): no event is re-recorded while a wait on it can be pending
    hipEvent_t ev_ar[kArEvents] = {};
    hipEvent_t ev_ar_done = nullptr;
    int ev_ar_next = 0;
    std::vector<std::pair<long long, long long>> ar_buckets;   // [lo, hi) float ranges of outer[], completion order
    bool ar_armed = false, ar_active = false, ar_issued = false, ar_failed = false;
    int ar_next = 0, ar_nt = 0;
    float ar_axpy = 0.f;            // second order: grad[range] += ar_axpy * hv[range] in front of the task sum (the last reverse step's update)
    int ar_launches = 0;            // collectives issued by the last overlapped exchange (tests, bench line)
    int ar_bucket_agreement = -1;   // mtts_comm_init: 1 = every rank holds the same bucket table, 0 = they disagreed (overlap off everywhere), -1 = no communicator yet
    int ar_idx_postnet() const { return 0; }
    int ar_idx_dec(int l) const { return 1 + (cfg.dec_layers - 1 - l); }
    int ar_idx_va() const { return 1 + cfg.dec_layers; }
    int ar_idx_spk() const { return 2 + cfg.dec_layers; }
    int ar_idx_enc(int l) const { return 3 + cfg.dec_layers + (cfg.enc_layers - 1 - l); }
    int ar_idx_last() const { return 2 + cfg.dec_layers + cfg.enc_layers; }
    // streams / events and the bucket table; 0 when the overlapped exchange is available
    int ar_setup() {
        if (!comm_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
            for (auto& e : ev_ar) HIP_CHECK(hipEventCreate(&e));
            HIP_CHECK(hipEventCreate(&ev_ar_done));
        }
        return build_buckets(ar_buckets);
    }
    // the module buckets of the flat parameter buffer, in backward-completion order; 0 when the architecture has them all
    int build_buckets(std::vector<std::pair<long long, long long>>& out) const {
        out.clear();
        if (cfg.enc_layers < 1) return 1;
        std::vector<const ParamEntry*> order;
        for (const ParamEntry& e : entries) order.push_back(&e);
        std::sort(order.begin(), order.end(), [](const ParamEntry* a, const ParamEntry* b) { return a->off < b->off; });
        auto layer_of = [](const std::string& n, const char* pre) { return atoi(n.c_str() + strlen(pre)); };
        auto bucket_of = [&](const std::string& n) -> int {
            if (n.rfind("postnet.", 0) == 0) return ar_idx_postnet();
            if (n.rfind("mel_linear.", 0) == 0) return cfg.dec_layers > 0 ? ar_idx_dec(cfg.dec_layers - 1) : ar_idx_postnet();
            if (n.rfind("decoder.layer_stack.", 0) == 0) return ar_idx_dec(layer_of(n, "decoder.layer_stack."));
            if (n.rfind("variance_adaptor.", 0) == 0) return ar_idx_va();
            if (n.rfind("speaker_emb.", 0) == 0) return ar_idx_spk();
            if (n.rfind("encoder.layer_stack.", 0) == 0) return ar_idx_enc(layer_of(n, "encoder.layer_stack."));
            if (n.rfind("encoder.", 0) == 0) return ar_idx_last();
            return -1;
        };
        const int nb = ar_idx_last() + 1;
        std::vector<std::pair<long long, long long>> rng((size_t)nb, {-1, -1});
        int prev = -2;
        for (size_t i = 0; i < order.size(); ++i) {
            const int b = bucket_of(order[i]->name);
            if (b < 0 || b >= nb) return 1;
            if (b != prev) {
                if (rng[(size_t)b].first >= 0) return 1;          // a module's tensors are not one contiguous run of the flat buffer
                rng[(size_t)b].first = order[i]->off;
                if (prev >= 0) rng[(size_t)prev].second = order[i]->off;
                prev = b;
            }
        }
        if (prev >= 0) rng[(size_t)prev].second = n_total;
        for (auto& r : rng) if (r.first >= 0 && r.second > r.first && ((r.second - r.first) % 4) == 0 && (r.first % 4) == 0) out.push_back(r); else if (r.first >= 0) { out.clear(); return 1; }
        // (a module without parameters has no bucket: the completion-order indices above then no longer match, so require them all)
        if ((int)out.size() != nb) { out.clear(); return 1; }
        long long covered = 0;
        for (const auto& r : out) covered += r.second - r.first;
        if (covered != n_total) { out.clear(); return 1; }   // (the buckets must tile the whole outer gradient)
        return 0;
    }
    void ar_destroy() {
        if (comm_stream) { hipStreamSynchronize(comm_stream); hipStreamDestroy(comm_stream); comm_stream = nullptr; }
        for (auto& e : ev_ar) if (e) { hipEventDestroy(e); e = nullptr; }
        if (ev_ar_done) { hipEventDestroy(ev_ar_done); ev_ar_done = nullptr; }
    }
    // start of an overlapped exchange (the gradient call that fills outer[]): true when armed and possible
    bool ar_begin(int nt, float axpy = 0.f) {
        const bool go = ar_armed && ar.sum != nullptr && comm_stream != nullptr && !ar_buckets.empty();
        ar_armed = false;
        if (!go) return false;
        ar_active = true; ar_failed = false; ar_next = 0; ar_nt = nt; ar_axpy = axpy; ar_launches = 0;
        return true;
    }
    void ar_wait_for(hipStream_t producer) {
        hipEvent_t ev = ev_ar[ev_ar_next];
        ev_ar_next = (ev_ar_next + 1) % kArEvents;
        hipEventRecord(ev, producer);
        hipStreamWaitEvent(comm_stream, ev, 0);
    }
    // the exchange tail (final once the loss of the pass is known): packed on the main stream, reduced on the comm stream
    void ar_tail() {
        const float w = bn_sync_mode == 1 ? 1.f / (float)ar.world : (ar.rank == 0 ? 1.f : 0.f);
        if (sync_pack(w)) { ar_failed = true; return; }
        ar_wait_for(stream);
        if (ar.sum(ar.ctx, outer + n_total, (size_t)sync_tail, comm_stream)) ar_failed = true;
        ++ar_launches;
    }
    // buckets 0 .. upto are complete (everything that writes them has been enqueued on the main / side stream): reduce those not yet sent
    void ar_ready(int upto) {
        if (!ar_active) return;
        if (upto >= (int)ar_buckets.size()) upto = (int)ar_buckets.size() - 1;
        if (ar_next > upto) return;
        ar_wait_for(stream);
        if (defer_live && side) ar_wait_for(side);
        for (; ar_next <= upto; ++ar_next) {
            const long long lo = ar_buckets[(size_t)ar_next].first, n = ar_buckets[(size_t)ar_next].second - lo;
            if (ar_axpy != 0.f)
                MTTS_LAUNCH(axpy_kernel, dim3(blocks_for(n / 4), 1, ar_nt), dim3(256), comm_stream, grad + lo, n_total, (const float*)(hv + lo), n_total, ar_axpy, n / 4);
            MTTS_LAUNCH(sum_tasks_kernel, dim3(blocks_for(n / 4)), dim3(256), comm_stream, (const float*)(grad + lo), n_total, ar_nt, 1.f, outer + lo, n / 4,
                        (int)outer_accumulate);
            if (ar.sum(ar.ctx, outer + lo, (size_t)n, comm_stream)) ar_failed = true;
            ++ar_launches;
        }
    }
    // end of the gradient call: whatever is left goes out, the exchange is "issued" (mtts_allreduce_outer joins it)
    int ar_end() {
        ar_ready((int)ar_buckets.size() - 1);
        ar_active = false; ar_axpy = 0.f;
        ar_issued = true;
        if (ar_failed) { set_error("overlapped all-reduce: a collective could not be issued"); return -1; }
        return 0;
    }
    // the main stream waits for the comm stream's collectives (before the clip + Adam reads outer[])
    void ar_join() {
        hipEventRecord(ev_ar_done, comm_stream);
        hipStreamWaitEvent(stream, ev_ar_done, 0);
        ar_issued = false;
    }

    // ---- inner SGD step, module by module behind the backward that produces its gradients (systems/utils.py:39-47 through learn2learn's
    // maml_update: p <- p - lr * g per parameter; no step of the reference orders the parameters against each other) --------------------
    // The monolithic update is an HBM-bound pass over every task's fast weights and gradients (3.4 GB at 8 tasks) that nothing overlaps
    // with when it sits between the backward and the next forward.  Cut at the same module boundaries as the exchange buckets, each piece
    // runs on `upd_stream` the moment its module's parameter gradients are complete (and its weights have been read for the last time in
    // this backward: a module's input-gradient GEMMs come before its hook), under the matrix-core-bound backward of the modules below it.
    // Same kernel, same floats: bit-identical to the monolithic update (tests/test_deferred_paths.py).  MTTS_UPD_OVERLAP=0: one launch.
    hipStream_t upd_stream = nullptr;
    static constexpr int kUpdEvents = 64;
    hipEvent_t ev_upd[kUpdEvents] = {};
    hipEvent_t ev_upd_done = nullptr;
    int ev_upd_next = 0;
    std::vector<std::pair<long long, long long>> upd_buckets;
    bool upd_active = false;
    int upd_next = 0, upd_nt = 0, upd_launches = 0;
    float upd_lr = 0.f;
    int upd_setup() {
        static const int on = [] { const char* e = getenv("MTTS_UPD_OVERLAP"); return e ? atoi(e) : 1; }();
        upd_buckets.clear();
        if (!on || n_adapt <= 0 || (adapt_start % 4) != 0 || (n_adapt % 4) != 0) return 1;
        if (build_buckets(upd_buckets)) return 1;
        if (!upd_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&upd_stream, hipStreamNonBlocking));
            for (auto& e : ev_upd) HIP_CHECK(hipEventCreate(&e));
            HIP_CHECK(hipEventCreate(&ev_upd_done));
        }
        return 0;
    }
    void upd_destroy() {
        if (upd_stream) { hipStreamSynchronize(upd_stream); hipStreamDestroy(upd_stream); upd_stream = nullptr; }
        for (auto& e : ev_upd) if (e) { hipEventDestroy(e); e = nullptr; }
        if (ev_upd_done) { hipEventDestroy(ev_upd_done); ev_upd_done = nullptr; }
    }
    bool upd_begin(int nt, float lr) {
        if (upd_stream == nullptr || upd_buckets.empty() || n_adapt <= 0 || inner_prox > 0.f) return false;
        upd_active = true; upd_next = 0; upd_nt = nt; upd_lr = lr; upd_launches = 0;
        return true;
    }
    void upd_wait_for(hipStream_t producer) {
        hipEvent_t ev = ev_upd[ev_upd_next];
        ev_upd_next = (ev_upd_next + 1) % kUpdEvents;
        hipEventRecord(ev, producer);
        hipStreamWaitEvent(upd_stream, ev, 0);
    }
    // modules 0 .. upto are done: their slices of the fast weights take the step
    void upd_ready(int upto) {
        if (!upd_active) return;
        if (upto >= (int)upd_buckets.size()) upto = (int)upd_buckets.size() - 1;
        bool waited = false;
        for (; upd_next <= upto; ++upd_next) {
            const long long lo = std::max(upd_buckets[(size_t)upd_next].first, adapt_start);
            const long long hi = std::min(upd_buckets[(size_t)upd_next].second, adapt_start + n_adapt);
            if (hi <= lo) continue;
            if (!waited) {
                upd_wait_for(stream);
                if (defer_live && side) upd_wait_for(side);
                waited = true;
            }
            MTTS_LAUNCH(sgd_update_kernel, dim3(blocks_for((hi - lo) / 4), 1, upd_nt), dim3(256), upd_stream, fast + (lo - adapt_start),
                        (const float*)(grad + lo), (hi - lo) / 4, upd_lr, n_adapt, n_total);
            ++upd_launches;
        }
    }
    // after the backward: the modules it did not reach (an encoder that is not adapted has no slice), then the main stream waits
    void upd_end() {
        upd_ready((int)upd_buckets.size() - 1);
        upd_active = false;
        hipEventRecord(ev_upd_done, upd_stream);
        hipStreamWaitEvent(stream, ev_upd_done, 0);
    }
    void module_done(int idx) { ar_ready(idx); upd_ready(idx); }

    // kernel-family choice of the pass's main-stream GEMM launches (gemm.h: gemm_glds_mode): the LDS-DMA family only when MTTS_GLDS=1 asks for it
    // per pass: the LDS-DMA family switch, and the critical stream's wavefront priority — in the deferred regime (weight gradients, run-ahead and
    // predictors on side streams beside an under-filled critical chain) the main stream's GEMM wavefronts issue at priority 3 (s_setprio), the
    // side streams' at the default 0: single-task rank 30.89 -> 30.39 ms, two tasks per rank 47.2 -> 46.8 ms; the 8-task step is unchanged with or without,
    // and a deferred launch of 6 750 frame rows (C2, batch 16) LOSES 1.1 % (fp32) / 2.4 % (bf16) — there the side stream's weight gradients are the
    // tail the step waits for — so the priority is raised up to kPrioMaxRows frame rows only (profiles/r06_ab_log.md).  MTTS_MAIN_PRIO=0: never; 2: every regime.
    static constexpr long long kPrioMaxRows = 5200;
    void set_regime(const Plan& p) {
        gx.no_glds = gemm_glds_mode() == 0;
        static const int mp = [] { const char* e = getenv("MTTS_MAIN_PRIO"); return e ? atoi(e) : 1; }();
        gx.wave_prio = (mp == 2 || (mp == 1 && defer_ok(p) && p.sumMf <= kPrioMaxRows)) ? 1 : 0;
    }

    // everything enqueued on the main stream so far happens before what is enqueued on the side stream next
    void fork_side() {
        hipEvent_t ev = ev_side[ev_next];
        ev_next = (ev_next + 1) % kSideEvents;
        hipEventRecord(ev, stream);
        hipStreamWaitEvent(side, ev, 0);
        defer_live = true;
    }
    // the main stream waits for the side stream's weight gradients (before anything reads or overwrites what they touch)
    void join_side() {
        if (!defer_live) return;
        hipEventRecord(ev_join, side);
        hipStreamWaitEvent(stream, ev_join, 0);
        defer_live = false;
    }

    // =================================================================================
    // variance predictor (lightning/model/modules.py:242-250)
    // =================================================================================
    void pred_fwd(const Pass& ps, const PredP& P, PredBuf& b, TS xin, Space s = SP_P) {
        TagScope tag_scope(*this, 4);
        const Plan& p = *ps.pl;
        const int d = cfg.d_model, f = cfg.vp_filter, k = cfg.vp_kernel;
        const unsigned char* im = inrect_mask(p, s);   // conv outputs / LayerNorm live on every position of the rectangle
        TS none{nullptr, 0};
        conv_fwd(ps, s, xin, d, k, W(ps, P.c1w), W(ps, P.c1b), f, b.r1, GEMM_RELU, im);
        ln_fwd(ps, s, b.r1, none, P.l1g, P.l1b, im, none, b.n1, b.st1, f, DropSpec(), drop_spec(ps, cfg.vp_dropout, site_base));
        conv_fwd(ps, s, b.n1, f, k, W(ps, P.c2w), W(ps, P.c2b), f, b.r2, GEMM_RELU, im);
        ln_fwd(ps, s, b.r2, none, P.l2g, P.l2b, im, none, b.n2, b.st2, f, DropSpec(), drop_spec(ps, cfg.vp_dropout, site_base + 1));
        TS w = W(ps, P.lw), bb = W(ps, P.lb);
        MTTS_LAUNCH(rowdot_kernel, row_grid(maxM(p, s), p.tasks), dim3(256), stream, (const int*)p.meta, mfield(s),
                    (const float*)b.n2.p, b.n2.ts, (const float*)w.p, (const float*)bb.p, w.ts, valid_mask(p, s),
                    row_ts(s), b.out.p, b.out.ts, f);
    }
    // The three predictors of a teacher-forced pass are independent (their inputs come from the TARGET embeddings): run them stage by
    // stage so that the three conv1 (then the three conv2) GEMMs — 28 workgroups each on a single-task rank — go out as ONE
    // multi-problem launch each.  Same arithmetic and dropout sites as three pred_fwd calls.
    void pred_fwd3(const Pass& ps, const PredP* const P[3], PredBuf* const b[3], const TS xin[3], const int sites[3]) {
        TagScope tag_scope(*this, 4);
        const Plan& p = *ps.pl;
        const int d = cfg.d_model, f = cfg.vp_filter, k = cfg.vp_kernel;
        const unsigned char* im = inrect_mask(p, SP_P);
        TS none{nullptr, 0};
        {
            GemmBatchScope batch(gx, stream);
            for (int i = 0; i < 3; ++i) conv_fwd(ps, SP_P, xin[i], d, k, W(ps, P[i]->c1w), W(ps, P[i]->c1b), f, b[i]->r1, GEMM_RELU, im);
        }
        for (int i = 0; i < 3; ++i)
            ln_fwd(ps, SP_P, b[i]->r1, none, P[i]->l1g, P[i]->l1b, im, none, b[i]->n1, b[i]->st1, f, DropSpec(), drop_spec(ps, cfg.vp_dropout, sites[i]));
        {
            GemmBatchScope batch(gx, stream);
            for (int i = 0; i < 3; ++i) conv_fwd(ps, SP_P, b[i]->n1, f, k, W(ps, P[i]->c2w), W(ps, P[i]->c2b), f, b[i]->r2, GEMM_RELU, im);
        }
        for (int i = 0; i < 3; ++i)
            ln_fwd(ps, SP_P, b[i]->r2, none, P[i]->l2g, P[i]->l2b, im, none, b[i]->n2, b[i]->st2, f, DropSpec(), drop_spec(ps, cfg.vp_dropout, sites[i] + 1));
        for (int i = 0; i < 3; ++i) {
            TS w = W(ps, P[i]->lw), bb = W(ps, P[i]->lb);
            MTTS_LAUNCH(rowdot_kernel, row_grid(p.maxMp, p.tasks), dim3(256), stream, (const int*)p.meta, (int)META_MP, (const float*)b[i]->n2.p,
                        b[i]->n2.ts, (const float*)w.p, (const float*)bb.p, w.ts, valid_mask(p, SP_P), row_ts_p, b[i]->out.p, b[i]->out.ts, f);
        }
    }
    // dout: [Mp] gradient of the prediction (0 on masked rows); dx accumulates the input gradient
    // pg != null (phoneme space only): deferred parameter gradients — the two weight-gradient GEMMs, the LayerNorm reductions and the
    // output layer's column sums of this predictor run on the side stream from buffers of its own (see LayerGrad)
    void pred_bwd(const Pass& ps, const PredP& P, PredBuf& b, TS xin, TS dout, TS dx, Space s = SP_P, PredGrad* pg = nullptr,
                  int dx_flags = GEMM_ACCUM) {
        TagScope tag_scope(*this, 4);
        const Plan& p = *ps.pl;
        const int d = cfg.d_model, f = cfg.vp_filter, k = cfg.vp_kernel;
        const unsigned char* im = inrect_mask(p, s);
        TS g1 = (s == SP_P) ? gPf1 : gRf1, g2 = (s == SP_P) ? gPf2 : gRf2;
        TS none{nullptr, 0};
        if (pg) {
            TS w = W(ps, P.lw);
            MTTS_LAUNCH(rowdot_bwd_kernel, row_grid(maxM(p, s), p.tasks), dim3(256), stream, (const int*)p.meta, mfield(s),
                        (const float*)dout.p, dout.ts, (const float*)w.p, w.ts, g1.p, g1.ts, f);
            // (the backward of the dropout behind each LayerNorm rides in the LayerNorm backward's load of its incoming gradient)
            ln_bwd(ps, s, g1, b.r2, b.st2, P.l2g, P.l2b, im, pg->g2a, f, 1, none, DropSpec(), false, pg->part2, drop_spec(ps, cfg.vp_dropout, site_base + 1));
            conv_dgrad(ps, s, pg->g2a, f, k, W(ps, P.c2w), f, g1, 0, im);
            ln_bwd(ps, s, g1, b.r1, b.st1, P.l1g, P.l1b, im, pg->g2b, f, 1, none, DropSpec(), false, pg->part1, drop_spec(ps, cfg.vp_dropout, site_base));
            conv_dgrad(ps, s, pg->g2b, f, k, W(ps, P.c1w), d, dx, GEMM_ACCUM, im);
            fork_side();
            {
                GemmBatchScope batch(gx_side, side);
                conv_wgrad(ps, s, pg->g2a, f, k, b.n1, f, P.c2w, P.c2b, im, 0, &gx_side, side);
                conv_wgrad(ps, s, pg->g2b, f, k, xin, d, P.c1w, P.c1b, im, 0, &gx_side, side);
            }
            ln_param_grads_side(ps, s, pg->part2, P.l2g, P.l2b, f);
            ln_param_grads_side(ps, s, pg->part1, P.l1g, P.l1b, f);
            colsum(ps, s, dout, 1, nullptr, none, Gd(P.lb), true);
            colsum(ps, s, b.n2, f, nullptr, dout, Gd(P.lw), true);
            defer_live = true;
            return;
        }
        colsum(ps, s, dout, 1, nullptr, none, Gd(P.lb));
        colsum(ps, s, b.n2, f, nullptr, dout, Gd(P.lw));
        TS w = W(ps, P.lw);
        MTTS_LAUNCH(rowdot_bwd_kernel, row_grid(maxM(p, s), p.tasks), dim3(256), stream, (const int*)p.meta, mfield(s),
                    (const float*)dout.p, dout.ts, (const float*)w.p, w.ts, g1.p, g1.ts, f);
        ln_bwd(ps, s, g1, b.r2, b.st2, P.l2g, P.l2b, im, g2, f, 1, none, DropSpec(), false, nullptr, drop_spec(ps, cfg.vp_dropout, site_base + 1));       // g2 = d conv2 out
        {
            GemmBatchScope pair(gx, stream);
            conv_wgrad(ps, s, g2, f, k, b.n1, f, P.c2w, P.c2b, im);
            conv_dgrad(ps, s, g2, f, k, W(ps, P.c2w), f, g1, 0, im);     // g1 = d n1
        }
        ln_bwd(ps, s, g1, b.r1, b.st1, P.l1g, P.l1b, im, g2, f, 1, none, DropSpec(), false, nullptr, drop_spec(ps, cfg.vp_dropout, site_base));       // g2 = d conv1 out
        {
            GemmBatchScope pair(gx, stream);
            conv_wgrad(ps, s, g2, f, k, xin, d, P.c1w, P.c1b, im);
            conv_dgrad(ps, s, g2, f, k, W(ps, P.c1w), d, dx, dx_flags, im);
        }
    }
    // The phoneme-level predictors' backward needs only the loss gradient and the forward's activations, and nothing reads its input
    // gradients before the length regulator's backward at the far end of the decoder: the whole chain (3 predictors x 11 small launches)
    // runs on the side stream under the PostNet / decoder backward, each predictor's input gradient into a buffer of its own (gPxE / gPxP / gPxD).
    bool pred_bwd_early(const Pass& ps) {
        const Plan& p = *ps.pl;
        static const int on = [] { const char* e = getenv("MTTS_PRED_EARLY"); return e ? atoi(e) : 1; }();
        if (!on || !side_pred_ok(p) || any_frame_level()) return false;
        fork_side();
        std::swap(stream, side);
        std::swap(gx, gx_side);
        std::swap(col_partial, col_partial_side);
        site_base = 136; pred_bwd(ps, eneP, eneB, x1, dpred[2], gPxE, SP_P, nullptr, 0);
        site_base = 132; pred_bwd(ps, pitP, pitB, x0, dpred[1], gPxP, SP_P, nullptr, 0);
        site_base = 128; pred_bwd(ps, durP, durB, x0, dpred[0], gPxD, SP_P, nullptr, 0);
        std::swap(col_partial, col_partial_side);
        std::swap(gx, gx_side);
        std::swap(stream, side);
        hipEventRecord(ev_pred, side);
        return true;
    }

    // =================================================================================
    // full forward (fastspeech2.py:40-112 / base_adaptor.py:41-95), teacher-forced
    // =================================================================================
    unsigned next_drop_seed() {
        unsigned sd = ((drop_base + 0x9E3779B9u) * 0x85EBCA6Bu) ^ (0x632BE5ABu * ++drop_counter);
        sd ^= sd >> 15;
        return sd ? sd : 1u;
    }
    // embedding + positions + encoder blocks (ps.pl->drop_seed already set); returns the encoder output
    TS encoder_fwd(const Pass& ps) {
        const Plan& p = *ps.pl;
        TS none{nullptr, 0};
        TS we = W(ps, word_emb);
        MTTS_LAUNCH(embed_pos_kernel, row_grid(p.maxMp, p.tasks), dim3(256), stream, (const int*)p.meta, (int)META_MP, emb_out.p,
                    emb_out.ts, (const float*)we.p, we.ts, (const float*)pos_table, (const int*)p.p_tok, (const int*)p.p_row_t,
                    (const unsigned char*)p.p_valid, row_ts_p, cfg.d_model);
        TS x = emb_out;
        for (int l = 0; l < cfg.enc_layers; ++l) { site_base = 2 * l; fft_fwd(ps, SP_P, cfg.enc_heads, encP[l], encB[l], x, none); x = encB[l].y2; }
        return x;
    }
    // Run-ahead (see kAhead): enqueue the encoder forwards of `steps` train-mode passes over plan `pl` on side2; seeds[s] / enc_ahead[s] /
    // ev_enc[s] belong to step s.  Returns false when the regime does not qualify (the caller then runs forward() as usual).
    // per_step_sets: step s writes the encoder's activations into activation set s + 1 (second-order MAML keeps them for its reverse sweep)
    // query (optional): the query pass's encoder forward runs ahead too, after the inner steps' — into the arena's own activation set,
    // where its backward will find the activations; *query_seed receives its dropout seed, ev_enc[steps] its completion
    bool run_encoder_ahead(Plan& pl, int steps, unsigned* seeds, bool per_step_sets = false, Plan* query = nullptr, unsigned* query_seed = nullptr) {
        static const int on = [] { const char* e = getenv("MTTS_ENC_AHEAD"); return e ? atoi(e) : 1; }();
        static const int all = [] { const char* e = getenv("MTTS_ENC_AHEAD_ALL"); return e ? atoi(e) : 1; }();   // also launches beyond the deferred regime
        auto ahead_ok = [&](const Plan& q) { return arena_pred != nullptr && q.tasks <= cap_tasks && (all || defer_ok(q)); };
        if (query && steps + 1 > kAhead) { query = nullptr; if (query_seed) *query_seed = 0; }   // no slot left for the query pass: the inner steps still run ahead
        if (!on || steps < 1 || steps > kAhead || encoder_adapted() || !ahead_ok(pl) || side2 == nullptr) return false;
        for (int s = 0; s < steps; ++s) seeds[s] = next_drop_seed();
        static const int q_on = [] { const char* e = getenv("MTTS_ENC_AHEAD_QUERY"); return e ? atoi(e) : 1; }();
        if (query && (!q_on || !ahead_ok(*query) || cfg.enc_layers < 1)) query = nullptr;
        if (query) *query_seed = next_drop_seed();   // (the seed forward() would draw for the query pass: after the inner steps')
        refresh_shadows(Pass{&pl, true, true});   // (bf16 mode: the encoder's weight shadows, before the fork)
        hipEvent_t ev = ev_side[ev_next];
        ev_next = (ev_next + 1) % kSideEvents;
        hipEventRecord(ev, stream);              // the batch image / plan kernels of this plan are on the main stream
        hipStreamWaitEvent(side2, ev, 0);
        std::swap(stream, side2);
        std::swap(gx, gx_side2);
        for (int s = 0; s < steps; ++s) {
            Pass pe{&pl, true, true};
            pl.drop_seed = seeds[s];
            if (per_step_sets) bind_act(s + 1);
            TS x = encoder_fwd(pe);
            MTTS_LAUNCH(copy_tasks_kernel, dim3((unsigned)std::min<long long>(((long long)pl.maxMp * cfg.d_model / 4 + 255) / 256, 1024), 1, pl.tasks), dim3(256),
                        stream, (const float*)x.p, x.ts, enc_ahead[s].p, enc_ahead[s].ts, (long long)pl.maxMp * cfg.d_model / 4);
            hipEventRecord(ev_enc[s], stream);
        }
        if (per_step_sets) bind_act(0);
        if (query) {
            Pass pe{query, true, true};
            query->drop_seed = *query_seed;
            encoder_fwd(pe);                         // (output = encB.back().y2 of activation set 0; nothing overwrites it before the query pass)
            hipEventRecord(ev_enc[steps], stream);
        } else if (query_seed) *query_seed = 0;
        std::swap(gx, gx_side2);
        std::swap(stream, side2);
        return true;
    }
    int forward(const Pass& ps) {
        const Plan& p = *ps.pl;
        set_regime(p);
        const int d = cfg.d_model, nt = p.tasks;
        TS none{nullptr, 0};
        if (ps.train) ps.pl->drop_seed = ps.seed_override ? ps.seed_override : next_drop_seed();
        refresh_shadows(ps, !ps.enc_out.p);   // (bf16 mode: the weight shadows this pass and its backward read)
        // encoder
        TS x = ps.enc_out.p ? ps.enc_out : encoder_fwd(ps);
        // speaker vector, added on every position of the phoneme rectangle
        TS tb = W(ps, spk_table);
        if (p.ext_spk)   // the batch's own embeddings (speaker_encoder.py:71-76 computed by the d-vector encoder): rows of the image
            MTTS_LAUNCH(copy_tasks_kernel, dim3((unsigned)(((long long)p.maxB * d / 4 + 255) / 256), 1, nt), dim3(256), stream,
                        (const float*)(p.img_dev + img.spk_emb), (long long)cap_B * d, spk.p, spk.ts, (long long)p.maxB * d / 4);
        else
        MTTS_LAUNCH(speaker_vec_kernel, dim3(p.maxB, 1, nt), dim3(64), stream, (const int*)p.meta, (const float*)tb.p, tb.ts,
                    (const int*)p.spk_ids, (long long)cap_B + 1, cap_B, p.average_spk, spk.p, spk.ts, d);
        MTTS_LAUNCH(add_rowvec_kernel, row_grid(p.maxMp, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP,
                    (const float*)x.p, x.ts, (const float*)spk.p, spk.ts, (const int*)p.p_row_b, (const unsigned char*)p.p_inrect,
                    row_ts_p, x0.p, x0.ts, d);
        // variance adaptor: targets select the embeddings when given, else the (controlled) predictions.  A phoneme-level feature
        // (modules.py:118-127) is handled on the phoneme rectangle before the length regulator, a frame-level one (:139-148) on the
        // frame rectangle after it.
        TS pe = W(ps, pitch_emb), ee = W(ps, energy_emb);
        const bool tf = p.has_targets;
        TS xp = x0;
        static const bool pred_batch = [] { const char* e = getenv("MTTS_PRED_BATCH"); return e ? atoi(e) != 0 : true; }();
        if (tf && pred_batch && !any_frame_level() && p.sumMp <= 2048) {   // (measured neutral once the launches fill the chip: 8-task meta-batches)
            // teacher-forced: both embeddings come from the targets, so x1 / x2 do not wait for a predictor — embed first, then the three
            // predictors side by side
            MTTS_LAUNCH(bucket_embed_add_kernel, row_grid(p.maxMp, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP, (const float*)x0.p, x0.ts,
                        (const float*)p.p_pitch_t, row_ts_p, 1.f, (const float*)pitch_bins, cfg.n_bins - 1, (const float*)pe.p, pe.ts,
                        (const unsigned char*)p.p_inrect, row_ts_p, pidx, x1.p, x1.ts, d);
            MTTS_LAUNCH(bucket_embed_add_kernel, row_grid(p.maxMp, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP, (const float*)x1.p, x1.ts,
                        (const float*)p.p_energy_t, row_ts_p, 1.f, (const float*)energy_bins, cfg.n_bins - 1, (const float*)ee.p, ee.ts,
                        (const unsigned char*)p.p_inrect, row_ts_p, eidx, x2.p, x2.ts, d);
            const PredP* const PP[3] = {&durP, &pitP, &eneP};
            PredBuf* const BB[3] = {&durB, &pitB, &eneB};
            const TS XX[3] = {x0, x0, x1};
            const int SS[3] = {128, 132, 136};
            // teacher-forced: nothing downstream in the forward reads the predictions (the decoder input uses the TARGET embeddings); with
            // an idle side stream the predictors overlap the decoder and are joined at the end of forward()
            static const bool pred_side = [] { const char* e = getenv("MTTS_PRED_SIDE"); return e ? atoi(e) != 0 : true; }();
            if (pred_side && side_pred_ok(p) && !defer_live) {
                fork_side();
                std::swap(stream, side);
                std::swap(gx, gx_side);
                pred_fwd3(ps, PP, BB, XX, SS);
                std::swap(gx, gx_side);
                std::swap(stream, side);
            } else {
                pred_fwd3(ps, PP, BB, XX, SS);
            }
            xp = x2;
        } else {
        site_base = 128; pred_fwd(ps, durP, durB, x0);
        if (!cfg.pitch_frame) {
            site_base = 132; pred_fwd(ps, pitP, pitB, xp);
            MTTS_LAUNCH(bucket_embed_add_kernel, row_grid(p.maxMp, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP,
                        (const float*)xp.p, xp.ts, tf ? (const float*)p.p_pitch_t : (const float*)pitB.out.p, tf ? row_ts_p : pitB.out.ts,
                        tf ? 1.f : ps.p_control, (const float*)pitch_bins, cfg.n_bins - 1, (const float*)pe.p, pe.ts,
                        (const unsigned char*)p.p_inrect, row_ts_p, pidx, x1.p, x1.ts, d);
            xp = x1;
        }
        if (!cfg.energy_frame) {
            site_base = 136; pred_fwd(ps, eneP, eneB, xp);
            MTTS_LAUNCH(bucket_embed_add_kernel, row_grid(p.maxMp, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP,
                        (const float*)xp.p, xp.ts, tf ? (const float*)p.p_energy_t : (const float*)eneB.out.p, tf ? row_ts_p : eneB.out.ts,
                        tf ? 1.f : ps.e_control, (const float*)energy_bins, cfg.n_bins - 1, (const float*)ee.p, ee.ts,
                        (const unsigned char*)p.p_inrect, row_ts_p, eidx, x2.p, x2.ts, d);
            xp = x2;
        }
        }
        va_out = xp;  // what the length regulator expands (backward needs to know which buffer it was)
        if (!tf && frames_from_predictions_impl(ps)) return -1;
        if (!any_frame_level()) {
            // length regulator + speaker + decoder positions
            MTTS_LAUNCH(length_regulate_fwd_kernel, row_grid(p.maxMf, nt), dim3(256), stream, (const int*)p.meta, (const float*)xp.p,
                        xp.ts, (const int*)p.f_src, (const int*)p.f_row_b, (const int*)p.f_row_t, row_ts_f, (const float*)spk.p, spk.ts,
                        (const float*)pos_table, dec_in.p, dec_in.ts, d);
        } else {
            MTTS_LAUNCH(length_regulate_rect_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, (const float*)xp.p, xp.ts,
                        (const int*)p.f_src, row_ts_f, (const int*)p.r2f, row_ts_r, xr0.p, xr0.ts, d);
            TS xr = xr0;
            if (cfg.pitch_frame) {
                site_base = 132; pred_fwd(ps, pitP, pitR, xr, SP_R);
                MTTS_LAUNCH(bucket_embed_add_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, (int)META_MR,
                            (const float*)xr.p, xr.ts, tf ? (const float*)p.r_pitch_t : (const float*)pitR.out.p, tf ? row_ts_r : pitR.out.ts,
                            tf ? 1.f : ps.p_control, (const float*)pitch_bins, cfg.n_bins - 1, (const float*)pe.p, pe.ts,
                            (const unsigned char*)p.r_inrect, row_ts_r, pidx_r, xr1.p, xr1.ts, d);
                xr = xr1;
            }
            if (cfg.energy_frame) {
                site_base = 136; pred_fwd(ps, eneP, eneR, xr, SP_R);
                MTTS_LAUNCH(bucket_embed_add_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, (int)META_MR,
                            (const float*)xr.p, xr.ts, tf ? (const float*)p.r_energy_t : (const float*)eneR.out.p, tf ? row_ts_r : eneR.out.ts,
                            tf ? 1.f : ps.e_control, (const float*)energy_bins, cfg.n_bins - 1, (const float*)ee.p, ee.ts,
                            (const unsigned char*)p.r_inrect, row_ts_r, eidx_r, xr2.p, xr2.ts, d);
                xr = xr2;
            }
            // packed decoder input = rectangle rows of the valid frames + speaker + positions
            MTTS_LAUNCH(length_regulate_fwd_kernel, row_grid(p.maxMf, nt), dim3(256), stream, (const int*)p.meta, (const float*)xr.p,
                        xr.ts, (const int*)p.f2r, (const int*)p.f_row_b, (const int*)p.f_row_t, row_ts_f, (const float*)spk.p, spk.ts,
                        (const float*)pos_table, dec_in.p, dec_in.ts, d);
        }
        x = dec_in;
        for (int l = 0; l < cfg.dec_layers; ++l) { site_base = 64 + 2 * l; fft_fwd(ps, SP_F, cfg.dec_heads, decP[l], decB[l], x, none); x = decB[l].y2; }
        // mel_linear: packed frames -> mel rectangle; padded frames carry the bias
        {
            GemmArgs g = rowgemm(p, SP_F, GEMM_NT);
            TS w = W(ps, mel_w), b = W(ps, mel_b);
            g.A = x.p; g.a_gs = x.ts; g.lda = d;
            g.B = w.p; g.b_gs = w.ts; g.ldb = d;
            g.C = mel.p; g.c_gs = mel.ts; g.ldc = cfg.n_mel;
            g.N = cfg.n_mel; g.K = d;
            g.bias = b.p; g.bias_gs = b.ts;
            g.c_rowmap = p.f2r; g.c_rowmap_gs = row_ts_f;
            gemm_launch(gx, GEMM_NT, g, p.maxMf, cfg.n_mel, nt, stream, 0, 2.0 * p.sum_nF * cfg.n_mel * d, p.sumMf,
                        4.0 * (p.sum_nF * (d + cfg.n_mel) + (double)nt * cfg.n_mel * d));
            MTTS_LAUNCH(fill_padded_rows_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, mel.p, mel.ts,
                        (const float*)b.p, b.ts, (const unsigned char*)p.r_inrect, (const unsigned char*)p.r_valid, row_ts_r,
                        cfg.n_mel);
        }
        // PostNet
        TS cur = mel;
        set_tag(3);
        for (int i = 0; i < cfg.postnet_layers; ++i) {
            const PostP& P = postP[i];
            PostBuf& b = postB[i];
            conv_fwd(ps, SP_R, cur, P.cin, cfg.postnet_kernel, W(ps, P.w), W(ps, P.b), P.cout, b.c, 0, p.r_inrect, TS{nullptr, 0}, TS{nullptr, 0},
                     TS{nullptr, 0}, i > 0, false);   // (bf16 mode: the previous layer's bn_apply wrote its output's operand plane)
            if (ps.train) {
                ColArgs ca;
                ca.X = b.c.p; ca.x_ts = b.c.ts; ca.mask = p.r_inrect; ca.mask_ts = row_ts_r; ca.C = P.cout; ca.mode = 2;
                ca.mfield = META_MR;
                colreduce(p, ca, b.stats.p, nullptr, b.stats.ts, p.maxMr);
                if (ps.update_bn && cfg.postnet_layers > kBnRunMax) {
                    MTTS_LAUNCH(bn_running_update_kernel, dim3((P.cout + 63) / 64), dim3(64), stream, (const float*)b.stats.p,
                                b.stats.ts, nt, bn_rm[i], bn_rv[i], P.cout, 0.1f);
                    bn_tracked[i] += nt;
                }
            } else {
                MTTS_LAUNCH(bn_eval_stats_kernel, dim3((P.cout + 63) / 64), dim3(64), stream, (const float*)bn_rm[i],
                            (const float*)bn_rv[i], b.stats.p, b.stats.ts, nt, P.cout, 1e-5f);
            }
            TS gm = W(ps, P.g), bt = W(ps, P.beta);
            MTTS_LAUNCH(bn_apply_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, (const float*)b.c.p, b.c.ts,
                        (const float*)b.stats.p, b.stats.ts, (const float*)gm.p, (const float*)bt.p, gm.ts,
                        (const unsigned char*)p.r_inrect, row_ts_r, (int)(i < cfg.postnet_layers - 1), b.a.p, b.a.ts, P.cout,
                        drop_spec(ps, cfg.postnet_dropout, 192 + i),   // F.dropout(..., 0.5, self.training), Layers.py:133-134, in passing
                        i + 1 < cfg.postnet_layers ? H(b.a.p) : (bf16_t*)nullptr);
            cur = b.a;
        }
        set_tag(0);
        if (ps.train && ps.update_bn && cfg.postnet_layers >= 1 && cfg.postnet_layers <= kBnRunMax) {   // the running buffers of every layer, one launch
            BnRunArgs ra;
            int maxC = 0;
            for (int i = 0; i < cfg.postnet_layers; ++i) {
                ra.stats[i] = postB[i].stats.p; ra.st_ts[i] = postB[i].stats.ts; ra.rm[i] = bn_rm[i]; ra.rv[i] = bn_rv[i]; ra.C[i] = postP[i].cout;
                maxC = std::max(maxC, postP[i].cout);
                bn_tracked[i] += nt;
            }
            MTTS_LAUNCH(bn_running_update_multi_kernel, dim3((maxC + 63) / 64, cfg.postnet_layers), dim3(64), stream, ra, nt, 0.1f);
        }
        // mel_post = postnet(mel) + mel   (whole [tasks][rows][n_mel] slab incl. guard rows: all zero there)
        {
            const long long n4 = (mel.ts * nt) / 4;
            const float* base_a = cur.p - (long long)G * cfg.n_mel;
            const float* base_b = mel.p - (long long)G * cfg.n_mel;
            float* base_o = mel_post.p - (long long)G * cfg.n_mel;
            MTTS_LAUNCH(add2_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), stream, base_a, base_b,
                        base_o, n4);
        }
        join_side();   // (the predictors of a teacher-forced pass may have run on the side stream)
        return 0;
    }

    // free-running (modules.py:132-137,167-190): durations = clamp(round(exp(logd) - 1) * d_control, 0) on the device, ONE
    // device -> host copy of the compact [task][B * S] image for ALL tasks (the reference reads every duration with .item()),
    // then the frame spaces are sized on the host (O(B) per task) and built on the device like any other plan
    int frames_from_predictions_impl(const Pass& ps) {
        Plan& p = *ps.pl;
        const int nt = p.tasks;
        MTTS_LAUNCH(duration_round_kernel, dim3(4, 1, nt), dim3(256), stream, (const int*)p.meta, (const float*)durB.out.p, durB.out.ts,
                    ps.d_control, (const unsigned char*)p.p_valid, row_ts_p, d_rounded.p);
        const long long bs = (long long)cap_B * cap_S;
        MTTS_LAUNCH(plan_gather_durations_kernel, dim3(2, 1, nt), dim3(256), stream, (const int*)p.meta, (const float*)d_rounded.p, d_rounded.ts,
                    p.dur_readback, cap_B, cap_S);
        fr_host.resize((size_t)nt * bs);
        HIP_CHECK(hipMemcpyAsync(fr_host.data(), p.dur_readback, (size_t)nt * bs * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        const int slot = (int)(&p - &plans[0]);
        for (int t = 0; t < nt; ++t) {
            TaskIn& in = p.in[t];
            const float* dr = fr_host.data() + (size_t)t * bs;
            in.durations.assign((size_t)in.B * in.S, 0);
            in.d_rounded.assign((size_t)in.B * in.S, 0.f);
            in.mel_lens.assign(in.B, 0);
            long long tmax = 0;
            for (int i = 0; i < in.B; ++i) {
                long long tot = 0;
                for (int s2 = 0; s2 < in.S; ++s2) {
                    const float v = dr[(size_t)i * in.S + s2];
                    const long long dd = std::max<long long>((long long)v, 0);  // max(int(expand_size), 0)
                    in.durations[(size_t)i * in.S + s2] = dd;
                    in.d_rounded[(size_t)i * in.S + s2] = v;
                    tot += dd;
                }
                in.mel_lens[i] = tot;
                tmax = std::max(tmax, tot);
            }
            if (tmax < 1) { set_error("free-running synthesis produced no frames (all predicted durations are 0)"); return -1; }
            if (tmax > cap_T) { set_error("predicted mel length exceeds engine capacity (T_max)"); return -1; }
            in.T_max = (int)tmax;
        }
        return build_plan(slot, true, ps.train);
    }
    std::vector<float> fr_host;

    LossArgs loss_args(const Plan& p) const {
        LossArgs a;
        a.mel = mel.p; a.mel_post = mel_post.p; a.mel_tgt = p.mel_tgt;
        a.rvalid = p.r_valid;
        a.pp = pitB.out.p; a.ep = eneB.out.p; a.logd = durB.out.p;
        a.p_tgt = p.p_pitch_t; a.e_tgt = p.p_energy_t; a.dur = p.p_dur; a.pvalid = p.p_valid;
        a.mel_ts = mel.ts; a.rrow_ts = row_ts_r; a.prow_ts = row_ts_p; a.pred_ts = pitB.out.ts;
        a.n_mel = cfg.n_mel;
        a.pitch_frame = cfg.pitch_frame; a.energy_frame = cfg.energy_frame;
        if (any_frame_level()) {
            a.pp_r = pitR.out.p; a.ep_r = eneR.out.p; a.pred_r_ts = pitR.out.ts;
            a.p_tgt_r = p.r_pitch_t; a.e_tgt_r = p.r_energy_t;
        }
        return a;
    }

    // losses_out (device, [tasks][6]) of the last forward
    int loss(const Pass& ps, float* losses_out) {
        const Plan& p = *ps.pl;
        if (!p.has_targets) { set_error("loss needs a teacher-forced batch (targets)"); return -1; }
        if (durB.out.ts != row_ts_p + 2 * G || pitB.out.ts != durB.out.ts) { set_error("internal: prediction stride"); return -1; }
        LossArgs a = loss_args(p);
        MTTS_LAUNCH(loss_partial_kernel, dim3(kLossBlocks, 1, p.tasks), dim3(256), stream, (const int*)p.meta, a, loss_partial);
        MTTS_LAUNCH(loss_final_kernel, dim3(p.tasks), dim3(64), stream, (const int*)p.meta, (const float*)loss_partial,
                    (int)kLossBlocks, cfg.n_mel, losses_out, cfg.pitch_frame, cfg.energy_frame);
        return 0;
    }

    // =================================================================================
    // full backward of (scale * total loss); writes every parameter gradient of the touched
    // modules into grad[task][...] (fully overwritten, no accumulation across calls)
    // =================================================================================
    int backward(const Pass& ps, float scale, bool need_encoder) {
        const int rc = backward_impl(ps, scale, need_encoder);
        join_side();
        return rc;
    }
    int backward_impl(const Pass& ps, float scale, bool need_encoder) {
        const Plan& p = *ps.pl;
        set_regime(p);
        const int d = cfg.d_model, nt = p.tasks, nm = cfg.n_mel;
        if (!ps.train) { set_error("backward needs a train-mode forward (batch statistics)"); return -1; }
        if (!p.has_targets) { set_error("backward needs a teacher-forced batch (targets)"); return -1; }
        LossArgs a = loss_args(p);
        GradSet& K = GK();   // where the gradients a second-order reverse sweep reads again go (default: the shared scratch)
        const TS gRm = K.gRm;
        // prediction strides in the phoneme space: the [Mp] vectors were allocated as rows(capMp, 1)
        MTTS_LAUNCH(loss_grad_kernel, dim3(kLossBlocks, 1, nt), dim3(256), stream, (const int*)p.meta, a, scale, gRm.p, gRp.p,
                    dpred[1].p, dpred[2].p, dpred[0].p, dpred_r[0].p, dpred_r[1].p);
        const bool pred_early = pred_bwd_early(ps);
        // ---- PostNet: cur = dL/d(a_i), starts as dL/d(mel_post); dc -> gR0, layer-input grad -> gR1
        TS cur = gRp;
        set_tag(3);
        for (int i = cfg.postnet_layers - 1; i >= 0; --i) {
            const PostP& P = postP[i];
            PostBuf& b = postB[i];
            const int act = (i < cfg.postnet_layers - 1);
            const float ysc = drop_active(ps) ? 1.f - cfg.postnet_dropout : 1.f;
            const DropSpec pdrop = drop_spec(ps, cfg.postnet_dropout, 192 + i);   // gradient through the mask: applied where dY is read
            TS dgm = Gd(P.g), dbt = Gd(P.beta);
            ColArgs ca;
            ca.yscale = ysc; ca.xdrop = pdrop;
            ca.X = cur.p; ca.x_ts = cur.ts; ca.Y = b.a.p; ca.y_ts = b.a.ts; ca.Z = b.c.p; ca.z_ts = b.c.ts;
            ca.stats = b.stats.p; ca.st_ts = b.stats.ts; ca.mask = p.r_inrect; ca.mask_ts = row_ts_r; ca.C = P.cout;
            ca.mode = 3; ca.do_tanh = act; ca.mfield = META_MR;
            colreduce(p, ca, dgm.p, dbt.p, dgm.ts, p.maxMr);
            TS gm = W(ps, P.g);
            const bool dfp = defer_ok(p);
            TS dc = dfp ? postG[i] : gR0;  // [rows][Cout] inside a scratch sized for max(postnet_dim, n_mel) channels (deferred: a buffer of the layer's own)
            MTTS_LAUNCH(bn_bwd_apply_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, (const float*)cur.p,
                        cur.ts, (const float*)b.a.p, b.a.ts, (const float*)b.c.p, b.c.ts, (const float*)b.stats.p, b.stats.ts,
                        (const float*)gm.p, gm.ts, (const float*)dgm.p, (const float*)dbt.p, dgm.ts,
                        (const unsigned char*)p.r_inrect, row_ts_r, act, dc.p, dc.ts, P.cout, ysc, pdrop, H(dc.p));
            TS xin = (i == 0) ? mel : postB[i - 1].a;
            if (dfp) {   // this layer's weight gradient on the side stream, overlapping the rest of the backward chain
                fork_side();
                GemmBatchScope batch(gx_side, side);
                conv_wgrad(ps, SP_R, dc, P.cout, cfg.postnet_kernel, xin, P.cin, P.w, P.b, p.r_inrect, 0, &gx_side, side);
                defer_live = true;
            }
            GemmBatchScope pair(gx, stream);
            if (!dfp) conv_wgrad(ps, SP_R, dc, P.cout, cfg.postnet_kernel, xin, P.cin, P.w, P.b, p.r_inrect);
            if (i > 0) {
                conv_dgrad(ps, SP_R, dc, P.cout, cfg.postnet_kernel, W(ps, P.w), P.cin, K.post_cur[i - 1], 0, p.r_inrect, TS{nullptr, 0}, TS{nullptr, 0},
                           TS{nullptr, 0}, true, false);
                cur = K.post_cur[i - 1];
            } else {
                // dL/d(mel) total = direct L1 term + residual path + PostNet input gradient
                const long long n4 = (gRm.ts * nt) / 4;
                MTTS_LAUNCH(add2_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), stream,
                            (const float*)(gRm.p - (long long)G * nm), (const float*)(gRp.p - (long long)G * nm),
                            gRm.p - (long long)G * nm, n4);
                conv_dgrad(ps, SP_R, dc, P.cout, cfg.postnet_kernel, W(ps, P.w), P.cin, gRm, GEMM_ACCUM, p.r_inrect, TS{nullptr, 0}, TS{nullptr, 0},
                           TS{nullptr, 0}, true, false);
            }
        }
        // ---- mel_linear -------------------------------------------------------------------
        set_tag(0);
        TS none{nullptr, 0};
        module_done(ar_idx_postnet());
        const bool dfm = defer_ok(p);   // gRm / gMelF are final from here on: their parameter gradients can run on the side stream
        if (!dfm) colsum(ps, SP_R, gRm, nm, nullptr, none, Gd(mel_b));
        MTTS_LAUNCH(gather_rows_kernel, row_grid(p.maxMf, nt), dim3(256), stream, (const int*)p.meta, (int)META_MF,
                    (const float*)gRm.p, gRm.ts, (const int*)p.f2r, row_ts_f, gMelF.p, gMelF.ts, nm);
        TS dec_out = cfg.dec_layers ? decB[cfg.dec_layers - 1].y2 : dec_in;
        if (dfm) {
            fork_side();
            {
                GemmBatchScope batch(gx_side, side);
                conv_wgrad(ps, SP_F, gMelF, nm, 1, dec_out, d, mel_w, -1, nullptr, 0, &gx_side, side);
            }
            colsum(ps, SP_R, gRm, nm, nullptr, none, Gd(mel_b), true);
            defer_live = true;
        }
        {
            GemmBatchScope pair(gx, stream);
            if (!dfm) conv_wgrad(ps, SP_F, gMelF, nm, 1, dec_out, d, mel_w, -1, nullptr);
            conv_dgrad(ps, SP_F, gMelF, nm, 1, W(ps, mel_w), d, K.dec_top, 0, nullptr);
        }
        // ---- decoder ----------------------------------------------------------------------
        for (int l = cfg.dec_layers - 1; l >= 0; --l) {
            TS xin = l == 0 ? dec_in : decB[l - 1].y2;
            site_base = 64 + 2 * l;
            fft_bwd(ps, SP_F, cfg.dec_heads, decP[l], decB[l], xin, l == cfg.dec_layers - 1 ? K.dec_top : K.dec[l + 1].g0, K.dec[l], dSf,
                    defer_ok(p) ? &decG[l] : nullptr, decLnPart.empty() ? nullptr : &decLnPart[2 * (size_t)l]);
            module_done(ar_idx_dec(l));   // (overlapped exchange: PostNet, mel_linear and the decoder layers down to l are complete)
        }
        const TS gF0 = cfg.dec_layers ? K.dec[0].g0 : K.dec_top;   // gradient of the decoder input
        // speaker vector gradient, part 1: every valid frame
        // (spk_side: parameter-gradient work of a pass whose phoneme-side gradient buffers stay untouched until the next pass — the
        // segment sums, the speaker table's and the bucket tables' gradients go to the side stream)
        const bool spk_side = pred_early && !need_encoder;
        if (!spk_side)
        MTTS_LAUNCH(segsum_rows_kernel, dim3((d + 63) / 64, p.maxB, nt), dim3(256), stream, (const int*)p.meta, (const float*)gF0.p,
                    gF0.ts, (const int*)p.f_seg_start, (const int*)p.f_seg_len, (long long)cap_B, dspk.p, dspk.ts, d, 0);
        // ---- frame-level half of the variance adaptor (frame rectangle), then the length regulator -> gP0 = dL/d(va_out) --------
        TS gLR = gF0;
        if (any_frame_level()) {
            MTTS_LAUNCH(gather_rows_kernel, row_grid(p.maxMr, nt), dim3(256), stream, (const int*)p.meta, (int)META_MR, (const float*)gF0.p,
                        gF0.ts, (const int*)p.r2f, row_ts_r, gRx.p, gRx.ts, d);          // dL/d(xr_last): 0 on padded frames
            if (cfg.energy_frame) {
                MTTS_LAUNCH(table_grad_kernel, dim3(cfg.n_bins, 1, nt), dim3(256), stream, (const int*)p.meta, (int)META_MR,
                            (const float*)gRx.p, gRx.ts, (const int*)eidx_r, row_ts_r, -1, Gd(energy_emb).p, n_total, d);
                site_base = 136; pred_bwd(ps, eneP, eneR, cfg.pitch_frame ? xr1 : xr0, dpred_r[1], gRx, SP_R);
            }
            if (cfg.pitch_frame) {
                MTTS_LAUNCH(table_grad_kernel, dim3(cfg.n_bins, 1, nt), dim3(256), stream, (const int*)p.meta, (int)META_MR,
                            (const float*)gRx.p, gRx.ts, (const int*)pidx_r, row_ts_r, -1, Gd(pitch_emb).p, n_total, d);
                site_base = 132; pred_bwd(ps, pitP, pitR, xr0, dpred_r[0], gRx, SP_R);
            }
            MTTS_LAUNCH(gather_rows_kernel, row_grid(p.maxMf, nt), dim3(256), stream, (const int*)p.meta, (int)META_MF, (const float*)gRx.p,
                        gRx.ts, (const int*)p.f2r, row_ts_f, gFx.p, gFx.ts, d);
            gLR = gFx;
        }
        const TS gLRo = pred_early ? gPx2 : gP0;
        MTTS_LAUNCH(length_regulate_bwd_kernel, row_grid(p.maxMp, nt), dim3(256), stream, (const int*)p.meta, (const float*)gLR.p,
                    gLR.ts, (const int*)p.p_first, (const int*)p.p_count, row_ts_p, gLRo.p, gLRo.ts, d, 0);
        // ---- phoneme-level half of the variance adaptor ---------------------------------------------------------------------
        if (pred_early) {
            // the predictors' input gradients are waiting in gPxE / gPxP / gPxD (pred_bwd_early): dL/d(x2) = gPx2 (just written),
            // dL/d(x1) = gPx2 + gPxE (into gPxE), dL/d(x0) = (dL/d(x1) + gPxP) + gPxD (into gP0) — the order in which three accumulating GEMM
            // epilogues would have added them; the bucket tables' gradients read gPx2 / gPxE on the side stream
            const long long n4 = (gP0.ts * nt) / 4, o = (long long)G * d;
            const dim3 ag((unsigned)std::min<long long>((n4 + 255) / 256, 2048));
            hipStreamWaitEvent(stream, ev_pred, 0);
            MTTS_LAUNCH(add2_kernel, ag, dim3(256), stream, (const float*)(gPx2.p - o), (const float*)(gPxE.p - o), gPxE.p - o, n4);
            MTTS_LAUNCH(add3_kernel, ag, dim3(256), stream, (const float*)(gPxE.p - o), (const float*)(gPxP.p - o), (const float*)(gPxD.p - o),
                        gP0.p - o, n4);
            fork_side();
            MTTS_LAUNCH(table_grad_kernel, dim3(cfg.n_bins, 1, nt), dim3(256), side, (const int*)p.meta, (int)META_MP,
                        (const float*)gPx2.p, gPx2.ts, (const int*)eidx, row_ts_p, -1, Gd(energy_emb).p, n_total, d);
            MTTS_LAUNCH(table_grad_kernel, dim3(cfg.n_bins, 1, nt), dim3(256), side, (const int*)p.meta, (int)META_MP,
                        (const float*)gPxE.p, gPxE.ts, (const int*)pidx, row_ts_p, -1, Gd(pitch_emb).p, n_total, d);
        } else {
        if (!cfg.energy_frame) {
            MTTS_LAUNCH(table_grad_kernel, dim3(cfg.n_bins, 1, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP,
                        (const float*)gP0.p, gP0.ts, (const int*)eidx, row_ts_p, -1, Gd(energy_emb).p, n_total, d);
            site_base = 136; pred_bwd(ps, eneP, eneB, cfg.pitch_frame ? x0 : x1, dpred[2], gP0, SP_P, defer_ok(p) ? &predG[2] : nullptr);
        }
        if (!cfg.pitch_frame) {
            MTTS_LAUNCH(table_grad_kernel, dim3(cfg.n_bins, 1, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP,
                        (const float*)gP0.p, gP0.ts, (const int*)pidx, row_ts_p, -1, Gd(pitch_emb).p, n_total, d);
            site_base = 132; pred_bwd(ps, pitP, pitB, x0, dpred[1], gP0, SP_P, defer_ok(p) ? &predG[1] : nullptr);
        }
        site_base = 128; pred_bwd(ps, durP, durB, x0, dpred[0], gP0, SP_P, defer_ok(p) ? &predG[0] : nullptr);
        }
        // speaker vector gradient, part 2: every position of the phoneme rectangle
        const hipStream_t sst = spk_side ? side : stream;
        if (spk_side)   // (after the fork above: gF0 and gP0 are final)
        MTTS_LAUNCH(segsum_rows_kernel, dim3((d + 63) / 64, p.maxB, nt), dim3(256), sst, (const int*)p.meta, (const float*)gF0.p,
                    gF0.ts, (const int*)p.f_seg_start, (const int*)p.f_seg_len, (long long)cap_B, dspk.p, dspk.ts, d, 0);
        MTTS_LAUNCH(segsum_rows_kernel, dim3((d + 63) / 64, p.maxB, nt), dim3(256), sst, (const int*)p.meta, (const float*)gP0.p,
                    gP0.ts, (const int*)p.p_seg_start, (const int*)p.p_seg_len, (long long)cap_B, dspk.p, dspk.ts, d, 1);
        if (!p.ext_spk)
        MTTS_LAUNCH(speaker_table_grad_kernel, dim3(cfg.n_speaker, 1, nt), dim3(64), sst, (const int*)p.meta,
                    (const float*)dspk.p, dspk.ts, (const int*)p.spk_ids, (long long)cap_B + 1, cap_B, p.average_spk,
                    Gd(spk_table).p, n_total, d);
        module_done(ar_idx_spk());        // variance adaptor + speaker table
        if (!need_encoder) return 0;
        // ---- encoder ------------------------------------------------------------------------
        for (int l = cfg.enc_layers - 1; l >= 0; --l) {
            TS xin = l == 0 ? emb_out : encB[l - 1].y2;
            site_base = 2 * l;
            fft_bwd(ps, SP_P, cfg.enc_heads, encP[l], encB[l], xin, gP0, LayerKeep{gP0, gPh, gP1, gP1, gPqkv}, dSp, defer_ok(p) ? &encG[l] : nullptr,
                    encLnPart.empty() ? nullptr : &encLnPart[2 * (size_t)l]);
            if (l > 0) module_done(ar_idx_enc(l));   // (layer 0's bucket also holds the word embedding, below)
        }
        // word embedding (padding row 0 keeps a zero gradient); p_tok is 0 on invalid rows, and
        // gP0 is only meaningful on valid rows -> scan with the token ids masked by validity
        MTTS_LAUNCH(table_grad_kernel, dim3(cfg.vocab, 1, nt), dim3(256), stream, (const int*)p.meta, (int)META_MP,
                    (const float*)gP0.p, gP0.ts, (const int*)p.p_tok, row_ts_p, 0, Gd(word_emb).p, n_total, d);
        return 0;
    }

    // =================================================================================
    // MAML (base_adaptor.py:98-124) and the outer update
    // =================================================================================
    // inner SGD step on the per-task fast weights; with inner_prox > 0 the proximal term of iMAML joins the gradient
    void inner_update(int nt, float inner_lr) {
        if (n_adapt <= 0) return;
        if (inner_prox > 0.f)
            MTTS_LAUNCH(sgd_prox_kernel, dim3(blocks_for(n_adapt / 4), 1, nt), dim3(256), stream, fast, (const float*)(grad + adapt_start),
                        (const float*)(theta + adapt_start), n_adapt / 4, inner_lr, inner_prox, n_adapt, n_total);
        else
            MTTS_LAUNCH(sgd_update_kernel, dim3(blocks_for(n_adapt / 4), 1, nt), dim3(256), stream, fast, (const float*)(grad + adapt_start),
                        n_adapt / 4, inner_lr, n_adapt, n_total);
    }
    // backward of an inner step + its SGD step (the step overlapped module by module when upd_setup made that possible)
    int inner_backward_update(const Pass& ps, int nt, float inner_lr) {
        const bool overlapped = upd_begin(nt, inner_lr);
        const int rc = backward(ps, 1.f, encoder_adapted());
        if (overlapped) upd_end();
        else if (!rc) { inner_update(nt, inner_lr); upd_launches = 0; }
        return rc;
    }
    // the inner-loop backward must reach the encoder when adapt.modules lists it (config/algorithm/dev.yaml does)
    bool encoder_adapted() const { return (cfg.adapt_mask >> MOD_ENCODER) & 1; }
    static unsigned blocks_for(long long n4) { return (unsigned)std::min<long long>(std::max<long long>((n4 + 255) / 256, 1), 4096); }

    // One meta-gradient: `steps` inner SGD steps on plan 0 (support), query pass on plan 1 with the
    // support speaker ids averaged.  First-order (the reference trains second-order: see DESIGN.md).
    // grad_scale = 1 / (total tasks of the meta-batch across all ranks).  Result: outer[] = sum over
    // local tasks of grad_scale * dL_query/dtheta; query losses in losses_out [tasks][6];
    // support losses per step in sup_losses_out [steps][tasks][6] (optional).
    int meta_grad(int steps, float inner_lr, float grad_scale, float* losses_out, float* sup_losses_out) {
        Plan& sp = plans[0];
        Plan& qp = plans[1];
        if (sp.tasks != qp.tasks || sp.tasks < 1) { set_error("support/query plans not set"); return -1; }
        const int nt = sp.tasks;
        if (n_adapt > 0)
            MTTS_LAUNCH(broadcast_kernel, dim3(blocks_for(n_adapt / 4), 1, nt), dim3(256), stream,
                        (const float*)(theta + adapt_start), fast, n_adapt / 4, n_adapt);
        Pass ps{&sp, true, true};
        unsigned seeds[kAhead], qseed = 0;
        const bool ahead = run_encoder_ahead(sp, steps, seeds, false, &qp, &qseed);
        for (int s = 0; s < steps; ++s) {
            if (ahead) { hipStreamWaitEvent(stream, ev_enc[s], 0); ps.seed_override = seeds[s]; ps.enc_out = enc_ahead[s]; }
            if (forward(ps)) return -1;
            if (sup_losses_out && loss(ps, sup_losses_out + (long long)s * nt * 6)) return -1;
            if (inner_backward_update(ps, nt, inner_lr)) return -1;
        }
        Pass pq{&qp, true, true};
        if (ahead && qseed) {   // the query pass's encoder ran ahead on the second side stream
            hipStreamWaitEvent(stream, ev_enc[steps], 0);
            pq.seed_override = qseed;
            pq.enc_out = encB[cfg.enc_layers - 1].y2;
        }
        if (forward(pq)) return -1;
        if (loss(pq, losses_out ? losses_out : losses)) return -1;
        if (!losses_out || losses_out == losses) { sync_nt = nt; sync_scale = grad_scale; }
        else ar_armed = false;               // (the exchange tail is packed from `losses`: a caller-supplied buffer takes the one-shot exchange)
        const bool overlap = ar_begin(nt);   // (mtts_arm_allreduce_overlap: the buckets leave as the query backward completes them)
        if (overlap) ar_tail();
        const int rc = backward(pq, grad_scale, true);
        if (overlap) { if (ar_end() || rc) return -1; return 0; }
        if (rc) return -1;
        MTTS_LAUNCH(sum_tasks_kernel, dim3(blocks_for(n_total / 4)), dim3(256), stream, (const float*)grad, n_total, nt, 1.f, outer,
                    n_total / 4, (int)outer_accumulate);
        return 0;
    }

    // Inner loop only (few-shot test loop, base_adaptor.py:170-173: cumulative first-order adaptation on the
    // same clone).  reset: start from theta (learner = self.learner.clone()), else continue on the fast weights.
    int adapt(int steps, float inner_lr, bool reset, float* sup_losses_out) {
        Plan& sp = plans[0];
        if (sp.tasks < 1 || !sp.has_targets) { set_error("support plan not set"); return -1; }
        const int nt = sp.tasks;
        if (reset && n_adapt > 0)
            MTTS_LAUNCH(broadcast_kernel, dim3(blocks_for(n_adapt / 4), 1, nt), dim3(256), stream,
                        (const float*)(theta + adapt_start), fast, n_adapt / 4, n_adapt);
        Pass ps{&sp, true, true};
        unsigned seeds[kAhead];
        const bool ahead = run_encoder_ahead(sp, steps, seeds);
        for (int s2 = 0; s2 < steps; ++s2) {
            if (ahead) { hipStreamWaitEvent(stream, ev_enc[s2], 0); ps.seed_override = seeds[s2]; ps.enc_out = enc_ahead[s2]; }
            if (forward(ps)) return -1;
            if (sup_losses_out && loss(ps, sup_losses_out + (long long)s2 * nt * 6)) return -1;
            if (inner_backward_update(ps, nt, inner_lr)) return -1;
        }
        return 0;
    }

    // Plain multi-task step gradient (baseline.py:25-36 -> system.py:53-56) on plan `slot`
    int plain_grad(int slot, float grad_scale, float* losses_out) {
        Plan& p = plans[slot];
        if (p.tasks < 1) { set_error("plan not set"); return -1; }
        Pass ps{&p, false, true};
        if (forward(ps)) return -1;
        if (loss(ps, losses_out ? losses_out : losses)) return -1;
        if (!losses_out || losses_out == losses) { sync_nt = p.tasks; sync_scale = grad_scale; }
        else ar_armed = false;
        const bool overlap = ar_begin(p.tasks);
        if (overlap) ar_tail();
        const int rc = backward(ps, grad_scale, true);
        if (overlap) { if (ar_end() || rc) return -1; return 0; }
        if (rc) return -1;
        MTTS_LAUNCH(sum_tasks_kernel, dim3(blocks_for(n_total / 4)), dim3(256), stream, (const float*)grad, n_total, p.tasks, 1.f,
                    outer, n_total / 4, (int)outer_accumulate);
        return 0;
    }

    // clip_grad_norm_(max_norm) + Adam on theta from `g` (device, n_total floats; usually outer[])
    const float* extra_sumsq = nullptr;   // device scalar added to the squared gradient norm before the clip (see sumsq_final_kernel)
    // gradient of the last backward's loss w.r.t. the per-utterance speaker vectors of `task` ([B][d_model]): what a speaker
    // encoder in front of the model back-propagates (speaker_emb: encoder / scratch_encoder)
    int get_speaker_grad(int task, int B, float* out_host) {
        if (task < 0 || task >= cap_tasks || B < 1 || B > cap_B || !out_host) { set_error("bad arguments"); return -1; }
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipMemcpy(out_host, dspk.p + (long long)task * dspk.ts, (size_t)B * cfg.d_model * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }
    // ---- the exchange step's tail: replicated side state that travels with the outer gradient ------------------------------------
    // outer[n_total ..): [0..5] the six losses of the last gradient call, summed over this rank's tasks and scaled by its grad_scale
    // (= 1 / total tasks: the SUM over ranks is the mean `log_dict(sync_dist=True)` reports, meta.py:78-79); [8 ..) the PostNet
    // BatchNorm running buffers weighted by bn_weight — 1 on rank 0 and 0 elsewhere reproduces DDP's broadcast_buffers (every rank
    // continues with rank 0's buffers, main.py:32), 1 / world their mean.  sync_unpack installs the reduced buffers, so replicas stay
    // bit-identical INCLUDING buffers.
    long long sync_tail = 0;
    int sync_nt = 0;
    float sync_scale = 1.f;
    int bn_sync_mode = 0;   // 0: rank 0's buffers (reference semantics), 1: mean over ranks
    SyncBn sync_bn() {
        SyncBn b;
        b.n = std::min(cfg.postnet_layers, 8);
        int off = 0;
        for (int i = 0; i < b.n; ++i) { b.rm[i] = bn_rm[i]; b.rv[i] = bn_rv[i]; b.c[i] = postP[i].cout; b.off[i] = off; off += 2 * postP[i].cout; }
        return b;
    }
    int sync_pack(float bn_weight) {
        if (cfg.postnet_layers > 8) { set_error("more than 8 PostNet layers"); return -1; }
        MTTS_LAUNCH(sync_pack_kernel, dim3(1 + cfg.postnet_layers), dim3(256), stream, (const float*)losses, sync_nt, sync_scale, outer + n_total,
                    sync_bn(), bn_weight);
        return 0;
    }
    int sync_unpack() {
        if (cfg.postnet_layers > 0) MTTS_LAUNCH(sync_unpack_kernel, dim3(cfg.postnet_layers), dim3(256), stream, (const float*)(outer + n_total), sync_bn());
        return 0;
    }
    int get_synced_losses(float* out6) {
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipMemcpy(out6, outer + n_total, 6 * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }

    int outer_update(const float* g, float lr, float b1, float b2, float eps, float weight_decay, float max_norm,
                     float* norm_out_host) {
        const int nb = 512;
        if (ar_issued) ar_join();  // (an overlapped exchange nobody joined with mtts_allreduce_outer: the clip must still see the reduced buffer)
        shadows_current = false;   // theta is about to move
        MTTS_LAUNCH(sumsq_partial_kernel, dim3(nb), dim3(256), stream, g, n_total / 4, norm_partial);
        MTTS_LAUNCH(sumsq_final_kernel, dim3(1), dim3(64), stream, (const float*)norm_partial, nb, norm_out, extra_sumsq);
        ++adam_step_count;
        const float bc1 = 1.f - (float)std::pow((double)b1, (double)adam_step_count);
        const float bc2 = 1.f - (float)std::pow((double)b2, (double)adam_step_count);
        MTTS_LAUNCH(adam_clip_kernel, dim3(blocks_for(n_total / 4)), dim3(256), stream, theta, g, adam_m, adam_v, n_total / 4,
                    (const float*)norm_out, max_norm, lr, b1, b2, eps, bc1, bc2, weight_decay);
        if (norm_out_host) {
            HIP_CHECK(hipStreamSynchronize(stream));
            HIP_CHECK(hipMemcpy(norm_out_host, norm_out, sizeof(float), hipMemcpyDeviceToHost));
        }
        return 0;
    }

#include "engine_so.inc"
#include "engine_imaml.inc"
};

}  // namespace mtts
