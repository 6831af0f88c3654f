// libmtts translation unit: C ABI (include/mtts.h) over the engine.  Built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC mtts.hip -o libmtts.so
// (tests/emu builds the same file for the host with -DMTTS_EMU; see compat.h).
#include "../../include/mtts.h"

#include "comm.h"
#include "engine.h"
#include "vocoder.h"
#include "dvector.h"
#include "melfront.h"

using namespace mtts;

static int copy_losses_impl(Engine& e, const float* dev, float* host, int n);
static int copy_losses(Engine& e, const float* dev, float* host, int n) { return copy_losses_impl(e, dev, host, n); }

struct mtts_handle {
    Engine eng;
    Comm comm;
    float* sup_losses_dev = nullptr;
    int sup_losses_cap = 0;
};

static std::string g_create_error;

// Launches are asynchronous: a rejected launch (bad configuration, out of resources) only shows up in hipGetLastError().
// Every compute entry point ends here so such a failure is reported as an error, never as silently stale results.
static int launched(Engine& e, int rc) {
    if (rc) return rc;
    for (GemmCtx* cx : {&e.gx, &e.gx_side, &e.gx_side2})
        if (cx->error) { e.set_error(std::string("GEMM launcher: ") + cx->error); cx->error = nullptr; return -1; }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { e.set_error(std::string("kernel launch failed: ") + hipGetErrorString(err)); return -1; }
    return 0;
}

struct mtts_vocoder {
    Vocoder v;
};
struct mtts_dvector {
    DVector d;
};
struct mtts_stft {
    MelFront m;
};

extern "C" {

int mtts_create(const mtts_model_cfg* c, int device, int max_tasks, int max_B, int max_S, int max_T, mtts_handle** out) {
    if (!c || !out || max_tasks < 1 || max_B < 1 || max_S < 1 || max_T < 1) { g_create_error = "bad arguments"; return -1; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed (no MI355X visible?)"; return -1; }
    mtts_handle* h = new mtts_handle();
    ModelCfg m;
    m.d_model = c->d_model; m.enc_layers = c->enc_layers; m.dec_layers = c->dec_layers; m.enc_heads = c->enc_heads;
    m.dec_heads = c->dec_heads; m.d_ff = c->d_ff; m.k1 = c->k1; m.k2 = c->k2; m.vp_filter = c->vp_filter;
    m.vp_kernel = c->vp_kernel; m.n_bins = c->n_bins; m.max_seq_len = c->max_seq_len; m.n_mel = c->n_mel;
    m.vocab = c->vocab; m.n_speaker = c->n_speaker; m.postnet_dim = c->postnet_dim; m.postnet_kernel = c->postnet_kernel;
    m.postnet_layers = c->postnet_layers; m.pitch_min = c->pitch_min; m.pitch_max = c->pitch_max;
    m.energy_min = c->energy_min; m.energy_max = c->energy_max; m.adapt_mask = c->adapt_mask;
    m.enc_dropout = c->enc_dropout; m.dec_dropout = c->dec_dropout; m.vp_dropout = c->vp_dropout;
    m.pitch_frame = c->pitch_frame_level != 0; m.energy_frame = c->energy_frame_level != 0;
    if (h->eng.init(m, max_tasks, max_B, max_S, max_T) != 0) {
        g_create_error = h->eng.last_error;
        delete h;
        return -1;
    }
    h->sup_losses_cap = 128;
    if (hipMalloc((void**)&h->sup_losses_dev, (size_t)h->sup_losses_cap * max_tasks * 6 * sizeof(float)) != hipSuccess) {
        g_create_error = "hipMalloc failed";
        delete h;
        return -1;
    }
    *out = h;
    return 0;
}

void mtts_destroy(mtts_handle* h) {
    if (!h) return;
    hipDeviceSynchronize();
    h->comm.release();
    h->eng.destroy();
    if (h->sup_losses_dev) hipFree(h->sup_losses_dev);
    delete h;
}

const char* mtts_last_error(mtts_handle* h) { return h ? h->eng.last_error.c_str() : g_create_error.c_str(); }

int mtts_set_stream(mtts_handle* h, void* s) { h->eng.stream = (hipStream_t)s; return 0; }
int mtts_set_grad_accumulation(mtts_handle* h, int accumulate) {
    if (!h) return -1;
    h->eng.outer_accumulate = accumulate != 0;
    return 0;
}

int mtts_set_numerics(mtts_handle* h, int mode) {
    if (!h) return -1;
    if (mode < 0 || mode > 2) { h->eng.set_error("numerics mode must be 0 (fp32), 1 (bf16 operands, fp32 accumulate) or 2 (1 without operand planes)"); return -1; }
    if (mode == 1 && h->eng.enable_planes()) return -1;   // (operand planes + weight shadows: allocated the first time the mode is selected)
    h->eng.planes_wanted = (mode == 1);
    h->eng.shadows_current = false;   // a forward in another mode did not refresh the weight shadows: planes again from the next forward on
    h->eng.gx.bf16 = h->eng.gx_side.bf16 = h->eng.gx_side2.bf16 = (mode != 0);
    return 0;
}
int mtts_get_numerics(mtts_handle* h) { return !h || !h->eng.gx.bf16 ? 0 : (h->eng.planes_wanted ? 1 : 2); }

int mtts_set_dropout(mtts_handle* h, int enable, unsigned seed) {
    Engine& e = h->eng;
    for (float p : {e.cfg.enc_dropout, e.cfg.dec_dropout, e.cfg.vp_dropout})
        if (p < 0.f || p >= 1.f) { e.set_error("dropout probability out of range"); return -1; }
    e.dropout_on = enable != 0;
    e.drop_base = seed;
    e.drop_counter = 0;
    return 0;
}
int mtts_synchronize(mtts_handle* h) { return hipStreamSynchronize(h->eng.stream) == hipSuccess ? 0 : -1; }

int mtts_param_count(mtts_handle* h) { return (int)h->eng.entries.size(); }
int mtts_param_info(mtts_handle* h, int i, const char** name, int* ndim, int shape[4], int64_t* off, int* adapted) {
    if (i < 0 || i >= (int)h->eng.entries.size()) { h->eng.set_error("param index out of range"); return -1; }
    const ParamEntry& e = h->eng.entries[i];
    if (name) *name = e.name.c_str();
    if (ndim) *ndim = (int)e.shape.size();
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = k < (int)e.shape.size() ? e.shape[k] : 1;
    if (off) *off = e.off;
    if (adapted) *adapted = e.off >= h->eng.adapt_start;
    return 0;
}
int64_t mtts_param_total(mtts_handle* h) { return h->eng.n_total; }
int64_t mtts_adapt_start(mtts_handle* h) { return h->eng.adapt_start; }
int mtts_load_param(mtts_handle* h, const char* name, const float* host, int64_t numel) { h->eng.shadows_current = false; return h->eng.load_param(name, host, numel); }
int mtts_import_state(mtts_handle* h, const char* name, int which, const float* host, int64_t numel) {
    h->eng.shadows_current = false;
    return h->eng.load_param(name, host, numel, which);
}
int mtts_set_optimizer_step(mtts_handle* h, int64_t step) { h->eng.adam_step_count = step; return 0; }
int mtts_export_param(mtts_handle* h, const char* name, int which, int task, float* host, int64_t numel) {
    if (task < 0 || task >= h->eng.cap_tasks) { h->eng.set_error("task out of range"); return -1; }
    return h->eng.export_param(name, which, task, host, numel);
}
int mtts_set_bn_buffers(mtts_handle* h, int layer, const float* mean, const float* var, int64_t tracked) {
    Engine& e = h->eng;
    if (layer < 0 || layer >= e.cfg.postnet_layers) { e.set_error("bn layer out of range"); return -1; }
    const int c = e.postP[layer].cout;
    if (hipMemcpy(e.bn_rm[layer], mean, c * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (hipMemcpy(e.bn_rv[layer], var, c * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
    e.bn_tracked[layer] = tracked;
    return 0;
}
int mtts_get_bn_buffers(mtts_handle* h, int layer, float* mean, float* var, int64_t* tracked) {
    Engine& e = h->eng;
    if (layer < 0 || layer >= e.cfg.postnet_layers) { e.set_error("bn layer out of range"); return -1; }
    const int c = e.postP[layer].cout;
    hipStreamSynchronize(e.stream);
    if (mean && hipMemcpy(mean, e.bn_rm[layer], c * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (var && hipMemcpy(var, e.bn_rv[layer], c * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (tracked) *tracked = e.bn_tracked[layer];
    return 0;
}

int mtts_set_bins(mtts_handle* h, const float* pb, const float* eb, int n) {
    Engine& e = h->eng;
    if (n != e.cfg.n_bins - 1) { e.set_error("bin count must be n_bins - 1"); return -1; }
    for (const float* b : {pb, eb})
        if (b) for (int i = 1; i < n; ++i) if (!(b[i] >= b[i - 1])) { e.set_error("bin boundaries must be non-decreasing"); return -1; }
    if (hipStreamSynchronize(e.stream) != hipSuccess) return -1;
    if (pb && hipMemcpy(e.pitch_bins, pb, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (eb && hipMemcpy(e.energy_bins, eb, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
}
int mtts_get_bins(mtts_handle* h, float* pb, float* eb, int n) {
    Engine& e = h->eng;
    if (n != e.cfg.n_bins - 1) { e.set_error("bin count must be n_bins - 1"); return -1; }
    if (hipStreamSynchronize(e.stream) != hipSuccess) return -1;
    if (pb && hipMemcpy(pb, e.pitch_bins, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (eb && hipMemcpy(eb, e.energy_bins, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

static HostBatch to_host_batch(const mtts_batch& b) {
    HostBatch o;
    o.B = b.B; o.S_max = b.S_max; o.T_max = b.T_max;
    o.speakers = (const long long*)b.speakers; o.texts = (const long long*)b.texts; o.src_lens = (const long long*)b.src_lens;
    o.mels = b.mels; o.mel_lens = (const long long*)b.mel_lens; o.pitches = b.pitches; o.energies = b.energies;
    o.durations = (const long long*)b.durations;
    o.spk_emb = b.spk_emb;
    return o;
}

int mtts_set_batches(mtts_handle* h, int slot, int n_tasks, const mtts_batch* batches, const mtts_batch* spk_from, int average_spk) {
    if (slot < 0 || slot > 1 || !batches || n_tasks < 1) { h->eng.set_error("bad arguments"); return -1; }
    std::vector<HostBatch> hb(n_tasks), sf(n_tasks);
    for (int t = 0; t < n_tasks; ++t) { hb[t] = to_host_batch(batches[t]); if (spk_from) sf[t] = to_host_batch(spk_from[t]); }
    return h->eng.set_batches(slot, n_tasks, hb.data(), spk_from ? sf.data() : nullptr, average_spk);
}

int mtts_forward(mtts_handle* h, int slot, int use_fast, int train) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || e.plans[slot].tasks < 1) { e.set_error("plan not set"); return -1; }
    if (e.plans[slot].has_targets && e.retarget(slot, train != 0)) return -1;
    Engine::Pass ps{&e.plans[slot], use_fast != 0, train != 0};
    return launched(e, e.forward(ps));
}

int mtts_synthesize(mtts_handle* h, int slot, int use_fast, int train, float p_control, float e_control, float d_control) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || e.plans[slot].tasks < 1) { e.set_error("plan not set"); return -1; }
    Engine::Pass ps{&e.plans[slot], use_fast != 0, train != 0, p_control, e_control, d_control};
    return launched(e, e.forward(ps));
}

int mtts_get_durations(mtts_handle* h, int slot, int task, float* d_rounded, int64_t* mel_lens, int* t_cap) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || task < 0 || task >= e.plans[slot].tasks) { e.set_error("bad slot/task"); return -1; }
    const Engine::Plan& pl = e.plans[slot];
    if (!pl.frames_ready) { e.set_error("durations are not known yet (run mtts_forward / mtts_synthesize first)"); return -1; }
    const Engine::TaskIn& in = pl.in[task];
    if (d_rounded) for (size_t i = 0; i < in.durations.size(); ++i) d_rounded[i] = in.d_rounded.size() == in.durations.size() ? in.d_rounded[i] : (float)in.durations[i];
    if (mel_lens) for (int i = 0; i < in.B; ++i) mel_lens[i] = std::min<long long>(in.mel_lens[i], pl.hTcap[task]);
    if (t_cap) *t_cap = pl.hTcap[task];
    return 0;
}

int mtts_adapt(mtts_handle* h, int steps, float inner_lr, int reset, float* sup_losses_host) {
    Engine& e = h->eng;
    if (steps < 0 || steps > h->sup_losses_cap) { e.set_error("too many inner steps"); return -1; }
    if (e.plans[0].tasks > 0 && e.retarget(0, true)) return -1;
    if (launched(e, e.adapt(steps, inner_lr, reset != 0, h->sup_losses_dev))) return -1;
    return copy_losses(e, h->sup_losses_dev, sup_losses_host, steps * e.plans[0].tasks * 6);
}

int mtts_get_outputs(mtts_handle* h, int slot, int task, float* mel, float* mel_post, float* p, float* en, float* logd) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || task < 0 || task >= e.plans[slot].tasks) { e.set_error("bad slot/task"); return -1; }
    const Engine::Plan& pl = e.plans[slot];
    const int B = pl.hB[task], S = pl.hSmax[task], T = pl.hTcap[task], nm = e.cfg.n_mel;
    if (hipStreamSynchronize(e.stream) != hipSuccess) return -1;
    for (int b = 0; b < B; ++b) {
        const long long r0 = G + (long long)b * (T + G);
        if (mel && hipMemcpy(mel + (size_t)b * T * nm, e.mel.p + task * e.mel.ts + r0 * nm, (size_t)T * nm * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (mel_post && hipMemcpy(mel_post + (size_t)b * T * nm, e.mel_post.p + task * e.mel_post.ts + r0 * nm, (size_t)T * nm * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        const long long p0 = G + (long long)b * (S + G);
        // a frame-level feature's prediction lives on the mel rows: [B][T_cap]
        if (p && !e.cfg.pitch_frame && hipMemcpy(p + (size_t)b * S, e.pitB.out.p + task * e.pitB.out.ts + p0, S * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (p && e.cfg.pitch_frame && hipMemcpy(p + (size_t)b * T, e.pitR.out.p + task * e.pitR.out.ts + r0, T * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (en && !e.cfg.energy_frame && hipMemcpy(en + (size_t)b * S, e.eneB.out.p + task * e.eneB.out.ts + p0, S * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (en && e.cfg.energy_frame && hipMemcpy(en + (size_t)b * T, e.eneR.out.p + task * e.eneR.out.ts + r0, T * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (logd && hipMemcpy(logd + (size_t)b * S, e.durB.out.p + task * e.durB.out.ts + p0, S * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return 0;
}

int mtts_get_mel_device(mtts_handle* h, int slot, int task, int postnet, const float** mel_dev, int* t_cap, int64_t* utt_stride) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || task < 0 || task >= e.plans[slot].tasks || !mel_dev) { e.set_error("bad slot/task"); return -1; }
    const Engine::Plan& pl = e.plans[slot];
    if (!pl.frames_ready) { e.set_error("no mel yet (run mtts_forward / mtts_synthesize first)"); return -1; }
    const int T = pl.hTcap[task], nm = e.cfg.n_mel;
    const auto& buf = postnet ? e.mel_post : e.mel;
    *mel_dev = buf.p + task * buf.ts + (long long)G * nm;      // row(b, t) = G + b * (T + G) + t
    if (t_cap) *t_cap = T;
    if (utt_stride) *utt_stride = (int64_t)(T + G) * nm;
    return 0;
}

static int copy_losses_impl(Engine& e, const float* dev, float* host, int n) {
    if (!host) return 0;
    if (hipStreamSynchronize(e.stream) != hipSuccess) return -1;
    return hipMemcpy(host, dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

int mtts_loss(mtts_handle* h, int slot, float* losses_host) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || e.plans[slot].tasks < 1) { e.set_error("plan not set"); return -1; }
    Engine::Pass ps{&e.plans[slot], false, true};
    if (launched(e, e.loss(ps, e.losses))) return -1;
    return copy_losses(e, e.losses, losses_host, e.plans[slot].tasks * 6);
}

int mtts_backward(mtts_handle* h, int slot, int use_fast, float scale, int need_encoder) {
    Engine& e = h->eng;
    if (slot < 0 || slot > 1 || e.plans[slot].tasks < 1) { e.set_error("plan not set"); return -1; }
    Engine::Pass ps{&e.plans[slot], use_fast != 0, true};
    return launched(e, e.backward(ps, scale, need_encoder != 0));
}

// mtts_arm_allreduce_overlap is one-shot: the flag must not outlive the gradient call it was set for.  A call that fails before it reaches
// ar_begin (forward / loss error, a failing reverse step) would otherwise leave it set, and the NEXT unrelated gradient call of this rank — a
// validation pass, a retry — would issue collectives its peers do not.
struct ArmGuard {
    Engine& e;
    ~ArmGuard() { e.ar_armed = false; }
};

int mtts_meta_grad(mtts_handle* h, int steps, float inner_lr, float grad_scale, int second_order, float* qry_losses_host,
                   float* sup_losses_host) {
    Engine& e = h->eng;
    ArmGuard disarm{e};   // an armed overlapped exchange is for THIS call only, however it ends
    if (steps < 0 || steps > h->sup_losses_cap) { e.set_error("too many inner steps"); return -1; }
    for (int sl = 0; sl < 2; ++sl) if (e.plans[sl].tasks > 0 && e.retarget(sl, true)) return -1;
    if (launched(e, second_order ? e.meta_grad_so(steps, inner_lr, grad_scale, e.losses, h->sup_losses_dev)
                                 : e.meta_grad(steps, inner_lr, grad_scale, e.losses, h->sup_losses_dev))) return -1;
    if (copy_losses(e, e.losses, qry_losses_host, e.plans[1].tasks * 6)) return -1;
    return copy_losses(e, h->sup_losses_dev, sup_losses_host, steps * e.plans[0].tasks * 6);
}

int mtts_hvp_support(mtts_handle* h) { return launched(h->eng, h->eng.hvp_support()); }
int mtts_reserve_second_order(mtts_handle* h, int steps) {
    Engine& e = h->eng;
    if (steps < 1 || steps > h->sup_losses_cap) { e.set_error("bad step count"); return -1; }
    if (e.enable_second_order(steps)) return -1;
    if (e.ensure_act_sets(steps)) (void)e.ensure_grad_sets(steps);   // optional: the engine falls back to recomputation without them
    return 0;
}

int mtts_set_inner_prox(mtts_handle* h, float reg_param) {
    if (!(reg_param >= 0.f)) { h->eng.set_error("reg_param must be >= 0"); return -1; }
    h->eng.inner_prox = reg_param;
    return 0;
}
int mtts_imaml_begin(mtts_handle* h, float* qry_losses_host) {
    Engine& e = h->eng;
    for (int sl = 0; sl < 2; ++sl) if (e.plans[sl].tasks > 0 && e.retarget(sl, true)) return -1;
    if (launched(e, e.imaml_begin(e.losses))) return -1;
    return copy_losses(e, e.losses, qry_losses_host, e.plans[1].tasks * 6);
}
int mtts_imaml_cg_step(mtts_handle* h, float inner_lr, float reg_param, float tol) {
    Engine& e = h->eng;
    if (e.plans[0].tasks > 0 && e.retarget(0, true)) return -1;
    return launched(e, e.imaml_cg_step(inner_lr, reg_param, tol));
}
int mtts_imaml_finish(mtts_handle* h, float inner_lr, float reg_param, float grad_scale, float max_norm, float* task_norms_host) {
    return launched(h->eng, h->eng.imaml_finish(inner_lr, reg_param, grad_scale, max_norm, task_norms_host));
}

int mtts_plain_grad(mtts_handle* h, int slot, float grad_scale, float* losses_host) {
    Engine& e = h->eng;
    ArmGuard disarm{e};
    if (slot < 0 || slot > 1) { e.set_error("bad slot"); return -1; }
    if (e.plans[slot].tasks > 0 && e.retarget(slot, true)) return -1;
    if (launched(e, e.plain_grad(slot, grad_scale, e.losses))) return -1;
    return copy_losses(e, e.losses, losses_host, e.plans[slot].tasks * 6);
}

float* mtts_outer_grad_ptr(mtts_handle* h) { return h->eng.outer; }

int mtts_comm_available(mtts_handle* h) {
    if (!h) return -1;
    if (h->comm.load()) { h->eng.set_error(h->comm.err); return -1; }
    return 0;
}
int mtts_comm_unique_id(mtts_handle* h, void* id128) {
    if (!id128) { h->eng.set_error("null id buffer"); return -1; }
    if (h->comm.unique_id((NcclUniqueId*)id128)) { h->eng.set_error(h->comm.err); return -1; }
    return 0;
}
int mtts_comm_init(mtts_handle* h, const void* id128, int rank, int world_size) {
    if (!id128) { h->eng.set_error("null id buffer"); return -1; }
    NcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (h->comm.init(id, rank, world_size)) { h->eng.set_error(h->comm.err); return -1; }
    // the overlapped, bucketed exchange (mtts_arm_allreduce_overlap): the engine issues its collectives through this hook
    Engine& e = h->eng;
    e.ar.ctx = &h->comm;
    e.ar.sum = [](void* c, float* buf, size_t n, hipStream_t st) { return ((Comm*)c)->sum(buf, n, st); };
    e.ar.rank = rank; e.ar.world = world_size;
    auto rollback = [&](const std::string& why) {   // no half-initialised communicator: the caller may fall back to its own all-reduce
        h->comm.release();
        e.ar.ctx = nullptr; e.ar.sum = nullptr; e.ar.rank = 0; e.ar.world = 1;
        e.ar_buckets.clear();
        e.set_error(why);
        return -1;
    };
    if (e.ar_setup() < 0) return rollback("overlapped exchange: stream / event creation failed");   // (> 0: no bucket table for this architecture — the one-shot exchange stays available)
    // every rank decides LOCALLY whether it has a bucket table; a rank that does not would issue 1 collective against its peers' N + 1.  Agree on
    // it here, collectively: SUM of (n, n^2) over the ranks — all n equal  <=>  world * sum(n^2) == (sum n)^2.  Any disagreement switches the
    // overlap off on EVERY rank (the one-shot exchange needs no table).
    {
        float probe[4] = {(float)e.ar_buckets.size(), (float)(e.ar_buckets.size() * e.ar_buckets.size()), 0.f, 0.f};
        float* dev = e.outer + e.n_total;            // the exchange tail: scratch until the first sync_pack
        if (hipMemcpyAsync(dev, probe, sizeof(probe), hipMemcpyHostToDevice, e.stream) != hipSuccess) return rollback("bucket agreement: copy failed");
        if (h->comm.sum(dev, 4, e.stream)) return rollback(h->comm.err);
        if (hipMemcpyAsync(probe, dev, sizeof(probe), hipMemcpyDeviceToHost, e.stream) != hipSuccess || hipStreamSynchronize(e.stream) != hipSuccess)
            return rollback("bucket agreement: copy back failed");
        const double sn = probe[0], sq = probe[1];
        e.ar_bucket_agreement = (fabs((double)world_size * sq - sn * sn) < 0.5) ? 1 : 0;
        if (!e.ar_bucket_agreement) e.ar_buckets.clear();
    }
    return 0;
}
int mtts_disarm_allreduce_overlap(mtts_handle* h) {
    if (!h) return -1;
    h->eng.ar_armed = false;
    return 0;
}
int mtts_arm_allreduce_overlap(mtts_handle* h) {
    if (!h) return -1;
    Engine& e = h->eng;
    static const int on = [] { const char* v = getenv("MTTS_AR_OVERLAP"); return v ? atoi(v) : 1; }();
    if (!on || !h->comm.comm || e.ar.sum == nullptr || e.comm_stream == nullptr || e.ar_buckets.empty()) return 1;
    e.ar_armed = true;
    return 0;
}
int mtts_allreduce_launches(mtts_handle* h) { return h ? h->eng.ar_launches : -1; }
int mtts_allreduce_bucket_agreement(mtts_handle* h) { return h ? h->eng.ar_bucket_agreement : -1; }
int mtts_inner_update_launches(mtts_handle* h) { return h ? h->eng.upd_launches : -1; }
int mtts_allreduce_outer(mtts_handle* h) {
    Engine& e = h->eng;
    if (!h->comm.comm) { e.set_error("communicator not initialised (mtts_comm_init)"); return -1; }
    e.ar_armed = false;
    if (e.ar_issued) {   // the gradient call already sent every bucket and the tail (overlapped exchange): wait for them, install the buffers
        e.ar_join();
        return launched(e, e.sync_unpack());
    }
    const float w = e.bn_sync_mode == 1 ? 1.f / (float)h->comm.world : (h->comm.rank == 0 ? 1.f : 0.f);
    if (e.sync_pack(w)) return -1;
    if (h->comm.sum(e.outer, (size_t)(e.n_total + e.sync_tail), e.stream)) { e.set_error(h->comm.err); return -1; }
    return launched(e, e.sync_unpack());
}
int64_t mtts_outer_sync_floats(mtts_handle* h) { return h->eng.n_total + h->eng.sync_tail; }
int mtts_sync_pack(mtts_handle* h, float bn_weight) { return launched(h->eng, h->eng.sync_pack(bn_weight)); }
int mtts_sync_unpack(mtts_handle* h) { return launched(h->eng, h->eng.sync_unpack()); }
int mtts_get_synced_losses(mtts_handle* h, float* out6) {
    if (!out6) { h->eng.set_error("null output"); return -1; }
    return h->eng.get_synced_losses(out6);
}
int mtts_set_bn_sync(mtts_handle* h, int mode) {
    if (mode != 0 && mode != 1) { h->eng.set_error("bn sync mode must be 0 (rank 0's buffers) or 1 (mean over ranks)"); return -1; }
    h->eng.bn_sync_mode = mode;
    return 0;
}

int mtts_outer_update(mtts_handle* h, const float* grad_dev, float lr, float b1, float b2, float eps, float wd, float max_norm,
                      float* norm_host) {
    return launched(h->eng, h->eng.outer_update(grad_dev ? grad_dev : h->eng.outer, lr, b1, b2, eps, wd, max_norm, norm_host));
}

int mtts_get_speaker_grad(mtts_handle* h, int task, int B, float* out) { return h->eng.get_speaker_grad(task, B, out); }
int mtts_set_extra_grad_sumsq(mtts_handle* h, const float* sumsq_dev) { h->eng.extra_sumsq = sumsq_dev; return 0; }
const float* mtts_grad_norm_dev(mtts_handle* h) { return h->eng.norm_out; }

int mtts_reset_optimizer(mtts_handle* h) {
    Engine& e = h->eng;
    e.adam_step_count = 0;
    if (hipMemsetAsync(e.adam_m, 0, e.n_total * sizeof(float), e.stream) != hipSuccess) return -1;
    if (hipMemsetAsync(e.adam_v, 0, e.n_total * sizeof(float), e.stream) != hipSuccess) return -1;
    return 0;
}

int mtts_profile_gemm(mtts_handle* h, int enable) {
    if (!h) return -1;
    int id = 0;
    for (GemmCtx* cx : {&h->eng.gx, &h->eng.gx_side, &h->eng.gx_side2}) {   // the side streams' launches (deferred parameter gradients) count too
        cx->prof.reset();
        cx->prof.enabled = enable != 0;
        cx->prof.ctx = id++;
    }
    return 0;
}

int mtts_profile_kinds(void) { return GK_COUNT; }
const char* mtts_profile_kernel_name(int kind) { return gemm_kind_name(kind); }
int mtts_profile_report(mtts_handle* h, double* out, int kinds) {
    if (!h || !out || kinds < GK_COUNT) return -1;
    for (int k = 0; k < GK_COUNT * 4; ++k) out[k] = 0.0;
    for (GemmCtx* cx : {&h->eng.gx, &h->eng.gx_side, &h->eng.gx_side2}) {
        double r[GK_COUNT][4];
        cx->prof.report(r);
        for (int k = 0; k < GK_COUNT; ++k) for (int j = 0; j < 4; ++j) out[k * 4 + j] += r[k][j];
    }
    return 0;
}

// launcher state of the handle-less kernel-level entry points: one context per host thread, WITHOUT a split-K workspace — nothing is
// allocated behind the caller's back, nothing is tied to the device that happened to be current on a thread's first call, and two
// streams driven from one thread share no counters or slabs (the launcher keeps S = 1 when a context has no workspace; split-K is a
// handle's business: its workspace is allocated by mtts_create on the handle's device and used on the handle's streams only)
static GemmCtx& kernel_ctx() {
    static thread_local GemmCtx cx;
    return cx;
}
// result of a kernel-level GEMM entry: a launch the launcher refused, or a launch error
static int kernel_launch_rc() {
    GemmCtx& cx = kernel_ctx();
    if (cx.error) { cx.error = nullptr; (void)hipGetLastError(); return -1; }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
static bool tile_code_ok(int tile) {
    return tile == 0 || tile == 4064 || ((tile % 1000 == 64 || tile % 1000 == 128) && tile % 10000 < 4000);
}

int mtts_gemm_f32(int form, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, float alpha, int flags, int tile, void* stream) {
    if (form < 0 || form > 2 || !tile_code_ok(tile)) return -1;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.alpha = alpha; g.flags = flags & 0xff;
    if (flags & ~0xff) return -1;
    gemm_launch(kernel_ctx(), form, g, M, N, 1, (hipStream_t)stream, tile);
    return kernel_launch_rc();
}

int mtts_xcd_schedule_check(const int* dims_in, int groups, int cls, int tn, int units_per_group, int* max_load_permille) {
    if (!dims_in || (groups != 8 && groups != 4 && groups != 2) || (cls != 1 && cls != 2) || tn < 1) return -1;
    int dims[8] = {0};
    for (int z = 0; z < groups; ++z) dims[z] = dims_in[z];
    XcdSched s;
    xcd_sched_build(s, dims, cls, tn, units_per_group, 64, groups);
    std::vector<std::vector<int>> seen(groups);
    long long total = 0, work[8] = {0};
    for (int z = 0; z < groups; ++z) {
        const int tiles = dims[z] <= 0 ? 0 : (cls == 1 ? ((dims[z] + 63) / 64) * tn : units_per_group);
        seen[z].assign(tiles, 0);
        total += (long long)tiles * (cls == 1 ? 1 : std::max(dims[z], 1));
    }
    const long slots = gemm_xcd_sched_slots(s);
    for (long lin = 0; lin < slots; ++lin) {
        int z = -1, tile = -1;
        if (!xcd_sched_locate(s, (int)lin, z, tile)) continue;
        if (z < 0 || z >= groups || tile < 0 || tile >= (int)seen[z].size() || seen[z][tile]++) return -1;
        work[lin & 7] += cls == 1 ? 1 : std::max(dims[z], 1);
    }
    for (int z = 0; z < groups; ++z)
        for (int v : seen[z]) if (v != 1) return -1;
    if (max_load_permille) {
        long long mx = 0;
        for (int x = 0; x < 8; ++x) mx = std::max(mx, work[x]);
        *max_load_permille = total > 0 ? (int)(mx * 8000 / total) : 0;
    }
    return (int)slots;
}

int mtts_gemm_f32_dual(int form, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* A2, const float* B2,
                       float* C, int ldc, const float* bias, float alpha, int flags, int tile, void* stream) {
    if (form < 0 || form > 2 || !tile_code_ok(tile) || !A2 || !B2 || (flags & ~0xff)) return -1;
    GemmArgs g;
    g.A = A; g.B = B; g.A2 = A2; g.B2 = B2; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.alpha = alpha; g.flags = flags & 0xff;
    gemm_launch(kernel_ctx(), form, g, M, N, 1, (hipStream_t)stream, tile);
    return kernel_launch_rc();
}

int mtts_gemm_bf16(int form, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* A2, const float* B2,
                   float* C, int ldc, const float* bias, float alpha, int flags, int tile, void* stream) {
    if (form < 0 || form > 2 || (tile != 0 && ((tile % 1000 != 64 && tile % 1000 != 128) || tile / 1000 > 4)) || (flags & ~0xff) || ((A2 == nullptr) != (B2 == nullptr))) return -1;
    GemmArgs g;
    g.A = A; g.B = B; g.A2 = A2; g.B2 = B2; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.alpha = alpha; g.flags = flags & 0xff;
    GemmCtx& cx = kernel_ctx();
    cx.bf16 = true;
    gemm_launch(cx, form, g, M, N, 1, (hipStream_t)stream, tile);
    cx.bf16 = false;
    return kernel_launch_rc();
}

int mtts_to_bf16(const float* src, unsigned short* dst, long long n, void* stream) {
    if (!src || !dst || n < 0 || n % 8 != 0) return -1;
    if (n == 0) return 0;
    MTTS_LAUNCH(to_bf16_kernel, dim3((unsigned)std::min<long long>((n / 8 + 255) / 256, 4096)), dim3(256), (hipStream_t)stream, src, (bf16_t*)dst, n / 8);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int mtts_gemm_bf16_planes(int M, int N, int K, const unsigned short* Ah, int lda, const unsigned short* Bh, int ldb, float* C, unsigned short* Ch,
                          int ldc, const float* bias, float alpha, int flags, int tile, void* stream) {
    if (!Ah || !Bh || !C || (tile != 0 && tile != 64 && tile != 128) || (flags & ~0xff) || K % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0) return -1;
    GemmArgs g;
    static const float anchor = 0.f;   // (plane-only problem: A / B carry no data, the planes are addressed relative to them)
    g.A = &anchor; g.B = nullptr; g.Ah = (const bf16_t*)Ah; g.Bh = (const bf16_t*)Bh; g.plane_only = true;
    g.C = C; g.Ch = (bf16_t*)Ch; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.alpha = alpha; g.flags = flags & 0xff;
    GemmCtx& cx = kernel_ctx();
    cx.bf16 = true;
    gemm_launch(cx, GEMM_NT, g, M, N, 1, (hipStream_t)stream, tile);
    cx.bf16 = false;
    return kernel_launch_rc();
}

long long mtts_plane_problems(mtts_handle* h) {
    return h ? h->eng.gx.plane_problems + h->eng.gx_side.plane_problems + h->eng.gx_side2.plane_problems : -1;
}

int mtts_conv1d_f32(int mode, int L, int Cin, int Cout, int k, const float* a, const float* b, float* out, const float* bias,
                    int tile, void* stream) {
    if (mode < 0 || mode > 2 || (k & 1) == 0 || k / 2 > G || !tile_code_ok(tile)) return -1;
    const int pad = k / 2;
    GemmArgs g;
    if (mode == 0) {         // y[L][Cout] = conv(x) + bias
        g.A = a - (long long)pad * Cin; g.lda = Cin; g.B = b; g.ldb = k * Cin; g.C = out; g.ldc = Cout;
        g.M = L; g.N = Cout; g.K = k * Cin; g.bias = bias;
        gemm_launch(kernel_ctx(), GEMM_NT, g, L, Cout, 1, (hipStream_t)stream, tile);
    } else if (mode == 1) {  // dx[L][Cin] = dgrad(dy [L][Cout], w)
        g.A = a - (long long)pad * Cout; g.lda = Cout; g.B = b; g.ldb = k * Cin; g.C = out; g.ldc = Cin;
        g.M = L; g.N = Cin; g.K = k * Cout; g.taps = k; g.tap_k = Cout; g.tap_bstride = Cin;
        gemm_launch(kernel_ctx(), GEMM_NN, g, L, Cin, 1, (hipStream_t)stream, tile);
    } else {                 // dw[Cout][k*Cin] = dy^T conv-rows(x)
        g.A = a; g.lda = Cout; g.B = b - (long long)pad * Cin; g.ldb = Cin; g.C = out; g.ldc = k * Cin;
        g.M = Cout; g.N = k * Cin; g.K = L;
        gemm_launch(kernel_ctx(), GEMM_TN, g, Cout, k * Cin, 1, (hipStream_t)stream, tile);
    }
    return kernel_launch_rc();
}

#include "kernel_api.inc"

// ---- MelGAN generator (vocoder.h; reference call site lightning/utils.py:8-30) ----------------------------
int mtts_vocoder_create(int n_mel, int ngf, int n_res, const int* ratios, int n_ratios, int device, int max_B, int max_T,
                        mtts_vocoder** out) {
    if (!out || !ratios || n_ratios < 1 || n_ratios > 8) { g_create_error = "bad arguments"; return -1; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed (no MI355X visible?)"; return -1; }
    mtts_vocoder* h = new mtts_vocoder();
    VocoderCfg c;
    c.n_mel = n_mel; c.ngf = ngf; c.n_res = n_res; c.n_ratios = n_ratios;
    for (int i = 0; i < n_ratios; ++i) c.ratios[i] = ratios[i];
    if (h->v.init(c, max_B, max_T) != 0) { g_create_error = h->v.last_error; h->v.destroy(); delete h; return -1; }
    *out = h;
    return 0;
}
void mtts_vocoder_destroy(mtts_vocoder* h) {
    if (!h) return;
    hipDeviceSynchronize();
    h->v.destroy();
    delete h;
}
const char* mtts_vocoder_last_error(mtts_vocoder* h) { return h ? h->v.last_error.c_str() : g_create_error.c_str(); }
int mtts_vocoder_set_stream(mtts_vocoder* h, void* s) { h->v.stream = (hipStream_t)s; return 0; }
int mtts_vocoder_hop(mtts_vocoder* h) { return h->v.hop; }
int mtts_vocoder_param_count(mtts_vocoder* h) { return (int)h->v.tensors.size(); }
int mtts_vocoder_param_info(mtts_vocoder* h, int i, char* name, int cap, int64_t* numel) {
    if (i < 0 || i >= (int)h->v.tensors.size()) return -1;
    snprintf(name, cap, "%s", h->v.tensors[i].name.c_str());
    *numel = h->v.tensors[i].numel;
    return 0;
}
int mtts_vocoder_load(mtts_vocoder* h, const char* name, const float* data, int64_t numel) { return h->v.load(name, data, numel); }
int mtts_vocoder_infer(mtts_vocoder* h, const float* mel, int B, int T_max, const int* mel_lens, float mel_scale, float* wav) {
    return h->v.infer_host(mel, B, T_max, mel_lens, mel_scale, wav);
}
int mtts_vocoder_infer_device(mtts_vocoder* h, const float* mel_dev, int64_t mel_utt_stride, int B, int T_max, const int* mel_lens,
                              float mel_scale, float* wav_dev) {
    return h->v.run(mel_dev, B, T_max, mel_lens, mel_scale, wav_dev, (long long)T_max * h->v.hop, (long long)mel_utt_stride);
}

// ---- d-vector speaker encoder (dvector.h; reference lightning/model/speaker_encoder.py:11-31,54-60,71-76) ----------------
int mtts_dvector_create(int n_mels, int hidden, int layers, int emb, int max_partials, int frames, int max_utts, int device, mtts_dvector** out) {
    if (!out) { g_create_error = "bad arguments"; return -1; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed (no MI355X visible?)"; return -1; }
    mtts_dvector* h = new mtts_dvector();
    if (h->d.init(n_mels, hidden, layers, emb, max_partials, frames, max_utts) != 0) { g_create_error = h->d.last_error; h->d.destroy(); delete h; return -1; }
    *out = h;
    return 0;
}
int mtts_dvector_set_stream(mtts_dvector* h, void* s) { if (!h) return -1; h->d.stream = (hipStream_t)s; return 0; }
void mtts_dvector_destroy(mtts_dvector* h) {
    if (!h) return;
    hipDeviceSynchronize();
    h->d.destroy();
    delete h;
}
const char* mtts_dvector_last_error(mtts_dvector* h) { return h ? h->d.last_error.c_str() : g_create_error.c_str(); }
int mtts_dvector_load(mtts_dvector* h, const char* name, const float* data, int64_t numel) { return h->d.load(name, data, numel); }
int mtts_dvector_embed(mtts_dvector* h, const float* mels, int n_partials, const int* utt_offsets, int n_utts, float* out, float* partial_out) {
    return h->d.embed(mels, n_partials, utt_offsets, n_utts, out, partial_out);
}
int mtts_dvector_enable_training(mtts_dvector* h) { return h->d.enable_training(); }
int mtts_dvector_embed_train(mtts_dvector* h, const float* mels, int n_partials, const int* utt_offsets, int n_utts, float* out) {
    return h->d.embed(mels, n_partials, utt_offsets, n_utts, out, nullptr, true);
}
int mtts_dvector_backward(mtts_dvector* h, const float* dout) { return h->d.backward(dout); }
const float* mtts_dvector_grad_sumsq(mtts_dvector* h) { return h->d.train_ready ? h->d.grad_sumsq() : nullptr; }
int mtts_dvector_adam_step(mtts_dvector* h, const float* norm_dev, float max_norm, float lr, float b1, float b2, float eps, float wd) {
    return h->d.adam_step(norm_dev, max_norm, lr, b1, b2, eps, wd);
}
int mtts_dvector_set_optimizer_step(mtts_dvector* h, int step) { h->d.adam_steps = step; return 0; }
int mtts_dvector_export(mtts_dvector* h, const char* name, int which, float* out, int64_t numel) { return h->d.export_state(name, which, out, numel); }
int mtts_dvector_import(mtts_dvector* h, const char* name, int which, const float* data, int64_t numel) { return h->d.import_state(name, which, data, numel); }

// ---- waveform -> log-mel + energy (melfront.h; reference audio/stft.py:128-178, audio/tools.py:8-15) ----------------------
int mtts_stft_create(int filter_length, int hop_length, int n_mel, int max_samples, int device, mtts_stft** out) {
    if (!out) { g_create_error = "bad arguments"; return -1; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed (no MI355X visible?)"; return -1; }
    mtts_stft* h = new mtts_stft();
    if (h->m.init(filter_length, hop_length, n_mel, max_samples) != 0) { g_create_error = h->m.last_error; h->m.destroy(); delete h; return -1; }
    *out = h;
    return 0;
}
int mtts_stft_set_stream(mtts_stft* h, void* s) { if (!h) return -1; h->m.stream = (hipStream_t)s; return 0; }
void mtts_stft_destroy(mtts_stft* h) {
    if (!h) return;
    hipDeviceSynchronize();
    h->m.destroy();
    delete h;
}
const char* mtts_stft_last_error(mtts_stft* h) { return h ? h->m.last_error.c_str() : g_create_error.c_str(); }
int mtts_stft_load(mtts_stft* h, const float* forward_basis, const float* mel_basis) { return h->m.load(forward_basis, mel_basis); }
int mtts_stft_mel_spectrogram(mtts_stft* h, const float* wav, int n_samples, float* mel, float* energy) {
    return h->m.mel_spectrogram(wav, n_samples, mel, energy);
}

}  // extern "C"
