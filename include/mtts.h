/* libmtts — C ABI of the MI355X-native Meta-TTS hot path (FastSpeech2 forward/backward inside the
 * MAML inner/outer loop).  Plain C, raw pointers and sizes only; no torch / C++ types cross this
 * boundary.  The reference (SungFeng-Huang/Meta-TTS) is pure Python with no FFI layer, so every
 * entry point below names the Python interface it sits beneath (file:line under /root/reference);
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success, non-zero on error (mtts_last_error() gives the
 * message); nothing throws across the ABI.  A handle owns all device memory it needs (allocated in
 * mtts_create for the stated capacities; no allocation afterwards — with ONE exception: second-order MAML,
 * mtts_meta_grad(second_order = 1) / mtts_hvp_support, allocates its tangent arena, its per-step activation and gradient
 * sets and its deferred-gradient buffers on FIRST use, so first-order / baseline / inference users do not pay for them;
 * mtts_reserve_second_order(h, steps) makes that allocation explicit and up-front).  All work is enqueued on the
 * handle's HIP stream (mtts_set_stream) and is asynchronous w.r.t. the host unless a function
 * copies results to a host pointer, in which case it synchronises that stream.  For small plans the
 * handle additionally uses two private non-blocking streams (parameter gradients, encoder run-ahead);
 * they are forked from and joined back into the handle's stream with events inside the same call, so
 * ordering against the caller's stream is exactly as if everything ran on it.  A handle must not
 * be used from two host threads at once, but DIFFERENT handles share no mutable state (launch queue,
 * profiler, split-K workspace and numerics mode are per handle) and may be driven from different host
 * threads concurrently.  "host" pointers are host memory, "dev" pointers device.
 */
#ifndef MTTS_H
#define MTTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtts_handle mtts_handle;

/* Sizes of config/model/base.yaml:1-30 plus what the reference reads from disk:
 * vocab = len(text.symbols)+1 (transformer/Models.py:40), n_speaker = len(speakers.json)
 * (lightning/model/speaker_encoder.py:49-50), pitch/energy min/max = stats.json
 * (lightning/model/modules.py:41-46), PostNet sizes (transformer/Layers.py:72-78). */
typedef struct mtts_model_cfg {
    int d_model, enc_layers, dec_layers, enc_heads, dec_heads, d_ff, k1, k2;
    int vp_filter, vp_kernel, n_bins, max_seq_len, n_mel, vocab, n_speaker;
    int postnet_dim, postnet_kernel, postnet_layers;
    float pitch_min, pitch_max, energy_min, energy_max;
    /* adapt.modules of config/algorithm/<name>.yaml as a bit mask over
     * {0 encoder, 1 variance_adaptor, 2 decoder, 3 mel_linear, 4 postnet, 5 speaker_emb}
     * (lightning/systems/base_adaptor.py:31-35) */
    int adapt_mask;
    /* transformer.{encoder,decoder}_dropout, variance_predictor.dropout (config/model/base.yaml:10-11,16); the PostNet's
     * 0.5 is hard-coded in the reference (transformer/Layers.py:133-134).  Only used after mtts_set_dropout(h, 1, seed). */
    float enc_dropout, dec_dropout, vp_dropout;
    /* preprocess_config["preprocessing"]["pitch" | "energy"]["feature"] == "frame_level" (lightning/model/modules.py:28-33,139-148,
     * loss.py:54-63): that feature's targets are [B][T_max] (one value per mel frame) and its predictor / embedding / loss run on
     * the frame rectangle after the length regulator.  0 = phoneme_level (config/preprocess/LibriTTS.yaml).  First- and second-order. */
    int pitch_frame_level, energy_frame_level;
} mtts_model_cfg;

/* One padded batch: elements [2:] of the reference 12-tuple (lightning/collate.py:47-60), host memory,
 * int64 where the reference uses LongTensor. */
typedef struct mtts_batch {
    int B, S_max, T_max;
    const int64_t* speakers;  /* [B]                 batch[2]  */
    const int64_t* texts;     /* [B][S_max]          batch[3]  */
    const int64_t* src_lens;  /* [B]                 batch[4]  */
    const float* mels;        /* [B][T_max][n_mel]   batch[6]  */
    const int64_t* mel_lens;  /* [B]                 batch[7]  */
    const float* pitches;     /* [B][S_max]          batch[9]   ([B][T_max] when pitch is frame-level)  */
    const float* energies;    /* [B][S_max]          batch[10]  ([B][T_max] when energy is frame-level) */
    const int64_t* durations; /* [B][S_max]          batch[11] */
    /* NULL, or [B][d_model] speaker embeddings used INSTEAD of the table lookup — `speaker_emb: dvec`, where batch[2] is
     * (ref_mels, ref_slices) and the embedding is the d-vector encoder's output (speaker_encoder.py:71-76; mtts_dvector_embed).
     * `speakers` is ignored then; no speaker-table gradient (the embeddings are inputs: no tangent in second-order passes either). */
    const float* spk_emb;
} mtts_batch;

/* ---- lifecycle (System.__init__, lightning/systems/system.py:30-48) ---------------------------- */
int mtts_create(const mtts_model_cfg* cfg, int device, int max_tasks, int max_B, int max_S, int max_T,
                mtts_handle** out);
void mtts_destroy(mtts_handle* h);
const char* mtts_last_error(mtts_handle* h); /* h may be NULL: error of the last failed mtts_create */
int mtts_set_stream(mtts_handle* h, void* hip_stream);
/* ---- gradient accumulation (main.py:62 `accumulate_grad_batches=grad_acc_step`): with accumulate != 0 the next mtts_meta_grad /
 * mtts_plain_grad calls ADD their (already grad_scale-d) result to the outer-gradient buffer instead of overwriting it; the caller
 * passes grad_scale / grad_acc_step, all-reduces and steps the optimizer once per grad_acc_step batches (systems.Trainer). */
int mtts_set_grad_accumulation(mtts_handle* h, int accumulate);

/* Numerics mode of the handle's contractions (every Linear / Conv1d / attention product, forward and backward; csrc/gemm_bf16.h).
 * 0 (default): fp32 operands on the fp32-input MFMA — the reference's own arithmetic (main.py:110-112 passes no `precision=`), the
 *    mode every parity gate (mel L1 <= 1e-4) is stated in.
 * 1: bf16 operands (rounded to nearest even once, on the way into LDS), exact products, fp32 accumulation — BASELINE.json configs[1]
 *    ("multi-task baseline bf16 on 1xMI355X"; what `Trainer(precision="bf16")` / torch.autocast(bfloat16) would make of the Linear /
 *    Conv1d / bmm ops).  Parameters, optimizer state, LayerNorm / BatchNorm / softmax / losses and every tensor in HBM stay fp32.
 *    The few problems whose conv taps do not cover whole 32-element K-slices (the PostNet's 80-channel output layer in its
 *    input-gradient form) keep the fp32 kernels — unless planes serve them, see below.  The long NT problems (the FFT blocks' and the
 *    PostNet's convolutions, forward and input gradient) read their operands from bf16 PLANES: a twin of the activation slab and a
 *    shadow of the weight (refreshed from the fp32 master once per pass; the input gradient as an NT problem over the transposed
 *    shadow) — the same rounded values, half the bytes through L2.  Allocated the first time the mode is selected (about half the
 *    activation arena + 2 bytes per shadowed weight element, twice).
 * 2: mode 1 without the planes (every operand is read as fp32 and rounded on its way into LDS): the A/B and test arm of mode 1 — the
 *    forward results are bit-identical to mode 1's.
 * May be switched between calls; applies to the handle's three streams. */
int mtts_set_numerics(mtts_handle* h, int mode);
int mtts_get_numerics(mtts_handle* h);

/* Train-mode dropout (nn.Dropout / F.dropout sites of SubLayers.py:54,90, modules.py:223,235, Layers.py:133-134).
 * Off by default — the parity configuration (SURVEY.md Appendix B.5 patches dropout to identity).  When on, masks are a
 * counter-based function of (seed, pass, site, element) regenerated in backward and in the second-order replay. */
int mtts_set_dropout(mtts_handle* h, int enable, unsigned seed);
int mtts_synchronize(mtts_handle* h);

/* ---- parameters: reference state_dict names without the "model." prefix,
 * torch layouts on the host side ((out,in) Linear, (Cout,Cin,k) Conv1d) -------------------------- */
int mtts_param_count(mtts_handle* h);
int mtts_param_info(mtts_handle* h, int index, const char** name, int* ndim, int shape[4], int64_t* flat_offset,
                    int* adapted);
int64_t mtts_param_total(mtts_handle* h);  /* floats in the flat parameter / gradient space */
int64_t mtts_adapt_start(mtts_handle* h);  /* first float of the adapted (fast-weight) slice */
int mtts_load_param(mtts_handle* h, const char* name, const float* host, int64_t numel);
/* which: 0 parameter, 1 outer gradient, 2 per-task gradient, 3 fast weight of `task`, 4 Adam m, 5 Adam v,
 * 6 Hessian-vector product of `task` (after mtts_hvp_support / a second-order mtts_meta_grad) */
int mtts_export_param(mtts_handle* h, const char* name, int which, int task, float* host, int64_t numel);
/* checkpoint resume: which = 0 parameter, 4 Adam exp_avg, 5 Adam exp_avg_sq; and the Adam step count
 * (PL ckpt["optimizer_states"], main.py:63 resume_from_checkpoint) */
int mtts_import_state(mtts_handle* h, const char* name, int which, const float* host, int64_t numel);
int mtts_set_optimizer_step(mtts_handle* h, int64_t step);
/* BatchNorm1d buffers of PostNet layer `layer` (running_mean, running_var, num_batches_tracked) */
int mtts_set_bn_buffers(mtts_handle* h, int layer, const float* mean_host, const float* var_host, int64_t tracked);
int mtts_get_bn_buffers(mtts_handle* h, int layer, float* mean_host, float* var_host, int64_t* tracked);

/* variance_adaptor.pitch_bins / energy_bins (frozen nn.Parameters, modules.py:57-71; n = n_bins - 1 boundaries each).  mtts_create
 * builds them from the cfg's stats.json min/max; a checkpoint trained on another corpus carries its own (system.py corpus-mismatch
 * branch), which the host layer installs here.  Either pointer may be NULL. */
int mtts_set_bins(mtts_handle* h, const float* pitch_bins_host, const float* energy_bins_host, int n);
int mtts_get_bins(mtts_handle* h, float* pitch_bins_host, float* energy_bins_host, int n);

/* ---- batches: slot 0 = support (or a plain batch), slot 1 = query.  `spk_from`/`average_spk`
 * reproduce forward_learner(..., sup_batch[2], *qry_batch[3:], average_spk_emb=True)
 * (lightning/systems/base_adaptor.py:64-70,122) --------------------------------------------------- */
int mtts_set_batches(mtts_handle* h, int slot, int n_tasks, const mtts_batch* batches, const mtts_batch* spk_from,
                     int average_spk);

/* ---- FastSpeech2.forward, teacher-forced (lightning/model/fastspeech2.py:40-112) and
 * FastSpeech2Loss.forward (lightning/model/loss.py:19-92) ---------------------------------------- */
int mtts_forward(mtts_handle* h, int slot, int use_fast_weights, int train_mode);
/* Same forward with the reference's p/e/d_control arguments.  A batch set without mels/durations runs
 * free-running (modules.py:132-137): durations = clamp(round(exp(logd)-1)*d_control, 0) on device, ONE
 * device->host copy per task sizes the frame spaces (the reference syncs once per phoneme), then the
 * decoder / PostNet run on the predicted length.  train_mode != 0 reproduces the reference's
 * post-adaptation synthesis (clone left in .train(): batch-stat BatchNorm, truncation at max_seq_len). */
int mtts_synthesize(mtts_handle* h, int slot, int use_fast_weights, int train_mode, float p_control, float e_control,
                    float d_control);
/* d_rounded [B][S_max] (duration_rounded of the 10-tuple), mel_lens [B], T_cap = width of mel / mel_post rows */
int mtts_get_durations(mtts_handle* h, int slot, int task, float* d_rounded, int64_t* mel_lens, int* t_cap);
/* copy the outputs of task `task` to host: mel, mel_post [B][T_cap][n_mel] (T_cap = min(T_max, max_seq_len));
 * p, e, logd [B][S_max] (p / e are [B][T_cap] for a frame-level feature).  Any pointer may be NULL. */
/* device view of the last forward's mel (postnet = 0) or mel_post (1) of one task: utterance b, frame t, channel c at
 * mel_dev[b * utt_stride + t * n_mel + c], t < t_cap.  Valid until the next set_batches / forward on that slot. */
int mtts_get_mel_device(mtts_handle* h, int slot, int task, int postnet, const float** mel_dev, int* t_cap, int64_t* utt_stride);
int mtts_get_outputs(mtts_handle* h, int slot, int task, float* mel, float* mel_post, float* p, float* e, float* logd);
int mtts_loss(mtts_handle* h, int slot, float* losses_host /* [n_tasks][6]: total, mel, postnet, pitch, energy, duration */);
/* gradient of scale * total loss w.r.t. every parameter of the touched modules -> per-task gradient */
int mtts_backward(mtts_handle* h, int slot, int use_fast_weights, float scale, int need_encoder);

/* ---- MAML: BaseAdaptorSystem.adapt + meta_learn (lightning/systems/base_adaptor.py:98-124),
 * learn2learn MAML.clone/adapt (lightning/systems/utils.py:17-77).  second_order = 1 is the reference's training
 * mode (`first_order = not train`, base_adaptor.py:107): the query gradient is propagated back through the inner
 * SGD steps by a Hessian-vector-product recursion (forward-over-reverse, csrc/engine_so.inc).  Produces the outer gradient
 * sum_t grad_scale * dL_query,t/dtheta in the handle's outer-gradient buffer.  Host loss pointers may
 * be NULL (then nothing synchronises). ------------------------------------------------------------ */
int mtts_meta_grad(mtts_handle* h, int steps, float inner_lr, float grad_scale, int second_order,
                   float* qry_losses_host /* [n_tasks][6] */, float* sup_losses_host /* [steps][n_tasks][6] */);
/* Pre-allocates everything a second-order mtts_meta_grad of `steps` inner steps would allocate on first use (the tangent arena, one
 * activation set and one gradient set per step — ~0.95 GB per full-size task and step —, the per-layer tangent-gradient buffers of the
 * deferred Hessian-vector weight gradients): after it, second-order calls allocate nothing.  Sets that do not fit are skipped (the
 * engine then recomputes: same results up to rounding, slower); returns 0 unless the mandatory tangent arena itself cannot be allocated. */
int mtts_reserve_second_order(mtts_handle* h, int steps);
/* Hessian-vector product of the support loss (slot 0) at the current fast weights in the direction currently held
 * in the per-task gradient buffer (e.g. after mtts_backward): the building block of the second-order sweep,
 * exposed for parity tests against torch.autograd (export with which = 6). */
int mtts_hvp_support(mtts_handle* h);
/* ---- iMAML (lightning/systems/imaml.py:22-150, lightning/systems/utils.py:120-189; `hypergrad` is an empty submodule of the
 * reference: its conjugate gradient is restated, parity unpinned there).  mtts_set_inner_prox(reg) adds the proximal term
 * 0.5 * reg * |theta_a - w|^2 (imaml.py:41-46,69) to the inner loss of every subsequent mtts_adapt / mtts_meta_grad step (0 = off).
 * Hypergradient of the query loss after mtts_adapt:  begin (query pass at the adapted weights; b = dL_q/dw, v = 0) ->
 * K x cg_step (one conjugate-gradient iteration on a * (H_support + reg * I), a = inner_lr; the Hessian-vector product runs on the
 * batch currently in slot 0, so a fresh support mini-batch may be set before each call: `imaml.stochastic`, imaml.py:88-91) ->
 * finish: outer gradient buffer := sum over local tasks of grad_scale * clip_t(a * reg * v_t) on the adapted parameters, 0 elsewhere
 * (imaml.py:125-131 clips every task's hypergradient to max_norm BEFORE the mean over ranks; max_norm <= 0: no clipping);
 * task_norms_host [n_tasks] (optional) receives the unclipped norms. */
int mtts_set_inner_prox(mtts_handle* h, float reg_param);
int mtts_imaml_begin(mtts_handle* h, float* qry_losses_host /* [n_tasks][6] */);
int mtts_imaml_cg_step(mtts_handle* h, float inner_lr, float reg_param, float tol);
int mtts_imaml_finish(mtts_handle* h, float inner_lr, float reg_param, float grad_scale, float max_norm, float* task_norms_host);
/* BaseAdaptorSystem.adapt alone (few-shot test loop, base_adaptor.py:155-189): `steps` first-order inner steps
 * on slot 0; reset != 0 starts from a fresh clone of theta, else continues on the current fast weights. */
int mtts_adapt(mtts_handle* h, int steps, float inner_lr, int reset, float* sup_losses_host /* [steps][n_tasks][6] */);
/* The inner SGD step (learn2learn maml_update through systems/utils.py:39-47: p <- p - lr * g per adapted parameter) runs module by module on
 * a stream of the handle behind the backward that produces the gradients, instead of as one pass between that backward and the next
 * forward (MTTS_UPD_OVERLAP=0, iMAML's proximal step and architectures without the module table keep the single pass; results are
 * bit-identical).  Returns the launches the LAST inner step's update took: > 1 module by module, 0 the single pass. */
int mtts_inner_update_launches(mtts_handle* h);
/* BaselineSystem.training_step (lightning/systems/baseline.py:25-36): plain gradient of slot's batches */
int mtts_plain_grad(mtts_handle* h, int slot, float grad_scale, float* losses_host);
/* device pointer of the outer gradient (mtts_param_total floats) — the buffer the host all-reduces
 * over RCCL between ranks (PL strategy="ddp", main.py:32) */
float* mtts_outer_grad_ptr(mtts_handle* h);
/* The exchange step over RCCL / xGMI without leaving the library (PL strategy="ddp", main.py:30-38: DDP's gradient all-reduce): one
 * communicator per handle = per rank.  Rank 0 calls mtts_comm_unique_id (128 bytes = NCCL_UNIQUE_ID_BYTES) and hands the id to the
 * other ranks over any out-of-band channel; every rank then calls mtts_comm_init (collective).  mtts_allreduce_outer enqueues
 * ncclAllReduce(SUM, fp32, in place) of the whole outer-gradient buffer on the handle's stream: ordered after the meta-gradient
 * kernels and before the following mtts_outer_update, no host synchronisation.  Each rank scales its contribution by
 * 1 / total_tasks through grad_scale, so the sum is the mean the reference takes.  librccl.so is resolved with dlopen on first use. */
int mtts_comm_available(mtts_handle* h);   /* 0 when librccl can be loaded in this process (every rank probes before rank 0 makes the id) */
int mtts_comm_unique_id(mtts_handle* h, void* id128);
int mtts_comm_init(mtts_handle* h, const void* id128, int rank, int world_size);
int mtts_allreduce_outer(mtts_handle* h);
/* Overlapped, bucketed exchange (DDP's bucketed gradient all-reduce overlapping the backward, main.py:30-38): call BEFORE the gradient call
 * that fills the outer gradient (mtts_meta_grad first or second order, mtts_plain_grad; with gradient accumulation: the window's LAST
 * call).  That call then cuts the flat buffer at module boundaries in backward-completion order (PostNet, decoder L-1 + mel_linear ...
 * decoder 0, variance adaptor, speaker table, encoder L-1 ... encoder 0 + word embedding) and issues one ncclAllReduce per bucket on a
 * communication stream of the handle the moment the backward has completed the module — behind events of the main and the
 * weight-gradient side stream — while the main stream continues; the exchange tail goes out first.  The following mtts_allreduce_outer
 * only makes the handle's stream wait for those collectives (its time is the EXPOSED part of the exchange).  One-shot: disarmed by
 * the gradient call.  Returns 0 when armed, 1 when the overlapped path is not available (no communicator, MTTS_AR_OVERLAP=0, an
 * architecture whose modules are not contiguous runs of the flat buffer) — mtts_allreduce_outer then reduces the whole buffer as before.
 * Results equal the one-shot exchange's (the same floats, the same collective, in pieces).  mtts_allreduce_launches: collectives the
 * last overlapped exchange issued (buckets + tail). */
int mtts_arm_allreduce_overlap(mtts_handle* h);
/* Take the arming back (a caller whose gradient call will not happen after all: an exception between arming and the call, an aborted
 * accumulation window).  The gradient calls also disarm on EVERY exit path, failures included, so a stale flag can never make a later,
 * unrelated gradient call of this rank issue collectives its peers do not.  mtts_comm_init makes the ranks agree on the bucket table
 * (a SUM of (n, n^2) over the ranks): if any rank has no table, the overlap is off on all of them — mtts_allreduce_bucket_agreement: 1 agreed,
 * 0 disagreed (overlap off everywhere), -1 no communicator. */
int mtts_disarm_allreduce_overlap(mtts_handle* h);
int mtts_allreduce_bucket_agreement(mtts_handle* h);
int mtts_allreduce_launches(mtts_handle* h);
/* What DDP moves between ranks BESIDES the gradient rides behind the flat outer gradient in the same buffer and the same collective:
 * [n_total .. n_total+6) the six losses of the last mtts_meta_grad / mtts_plain_grad call, summed over this rank's tasks and scaled by its
 * grad_scale — the SUM over ranks is the mean `self.log_dict(..., sync_dist=True)` reports (meta.py:78-79, baseline.py:35); behind them the
 * PostNet BatchNorm running_mean | running_var of every layer, weighted so that the SUM is rank 0's buffers (mode 0, default: DDP's
 * broadcast_buffers, main.py:32) or the mean over ranks (mode 1).  mtts_allreduce_outer packs, reduces mtts_outer_sync_floats() floats and
 * installs the reduced buffers itself; a caller that runs the collective on its own (torch.distributed on mtts_outer_grad_ptr) brackets
 * it with mtts_sync_pack(h, w) (w = this rank's BatchNorm weight: 1 / 0 for mode 0, 1 / world for mode 1) and mtts_sync_unpack.
 * mtts_get_synced_losses: the six reduced loss scalars (after the collective; with one rank, after mtts_sync_pack). */
int64_t mtts_outer_sync_floats(mtts_handle* h);
int mtts_sync_pack(mtts_handle* h, float bn_weight);
int mtts_sync_unpack(mtts_handle* h);
int mtts_get_synced_losses(mtts_handle* h, float* out6_host);
int mtts_set_bn_sync(mtts_handle* h, int mode);
/* clip_grad_norm_(max_norm) (main.py:61) + Adam (lightning/optimizer.py:9-15) with the learning rate of
 * lightning/scheduler.py:11-23 supplied by the caller.  grad_dev NULL = the internal outer gradient. */
int mtts_outer_update(mtts_handle* h, const float* grad_dev, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float max_norm, float* grad_norm_host);
int mtts_reset_optimizer(mtts_handle* h);
/* Hooks for a module trained in front of the engine (the speaker encoder): the gradient of the last backward pass w.r.t. the
 * per-utterance speaker vectors of `task` (out [B][d_model], host); a device scalar added to the squared gradient norm inside
 * mtts_outer_update (NULL: none); the device address of the norm mtts_outer_update computes. */
int mtts_get_speaker_grad(mtts_handle* h, int task, int B, float* out);
int mtts_set_extra_grad_sumsq(mtts_handle* h, const float* sumsq_dev);
const float* mtts_grad_norm_dev(mtts_handle* h);

/* ---- measurement: per-launch HIP-event timing of this handle's GEMM launches on its stream.
 * report: out[kind][4] = {launches, total ms, total algorithmic flops, total algorithmic bytes (every operand and the output
 * moved once, fp32)}; `kinds` = mtts_profile_kinds() rows, one per real kernel symbol — mtts_profile_kernel_name(kind) is the name
 * a rocprofv3 kernel trace prints for it (csrc/gemm.h: GemmKind).  (bench.py roofline leg) */
int mtts_profile_gemm(mtts_handle* h, int enable);
int mtts_profile_kinds(void);
const char* mtts_profile_kernel_name(int kind);
int mtts_profile_report(mtts_handle* h, double* out, int kinds);

/* ---- kernel-level entry points (parity tests; dev pointers; stream may be NULL) ----------------
 * form 0: C[M,N] = alpha*A[M,K]*B[N,K]^T + bias   1: C = A[M,K]*B[K,N]   2: C[M,N] = A[K,M]^T*B[K,N]
 * flags bit0 ReLU, bit1 accumulate;
 * tile 0 (auto: the launch queue — a plain 64x64 grid, the LDS-DMA kernels for under-filled launches) / 64 / 128 (+1000 software-pipelined
 * variant, +2000 BK = 32) / 4064 LDS-DMA kernel.  These handle-less entries never split K (no hidden workspace). */
int mtts_gemm_f32(int form, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, float alpha, int flags, int tile, void* hip_stream);
/* Dual-source product in one accumulator chain: C = alpha * (op(A, B) + op(A2, B2)) + bias, same form / sizes / leading dimensions for
 * both pairs — the shape of every tangent product of second-order MAML, t(X W) = tX W + X tW (csrc/gemm.h: GemmArgs::A2;
 * reference: the double backward that `higher` records for lightning/systems/base_adaptor.py:107). */
int mtts_gemm_f32_dual(int form, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* A2, const float* B2,
                       float* C, int ldc, const float* bias, float alpha, int flags, int tile, void* hip_stream);
/* The bf16 operand family on one problem (csrc/gemm_bf16.h): C = alpha * (op(A, B) [+ op(A2, B2)]) + bias with fp32 A / B in memory,
 * rounded to bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation.  A2 / B2 both NULL or both set.  tile 0 (auto) / 64 / 128
 * (+ 1000 * K-slices kept in flight per workgroup: micro-benchmarks). */
int mtts_gemm_bf16(int form, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* A2, const float* B2,
                   float* C, int ldc, const float* bias, float alpha, int flags, int tile, void* hip_stream);
/* bf16 operand planes at kernel level (numerics mode 1's long NT problems): mtts_to_bf16 rounds n (a multiple of 8) floats to bf16
 * (round to nearest even) — the plane of an activation slab; mtts_gemm_bf16_planes computes C[M][N] = alpha * Ah[M][K] . Bh[N][K]^T
 * (+ bias, flags as mtts_gemm_f32) from the two planes and, when Ch is set, writes C's own bf16 twin beside it.  K, lda, ldb in whole
 * groups of 8.  mtts_plane_problems: how many GEMM problems of this handle have run on the plane-staged K-loop so far. */
int mtts_to_bf16(const float* src, unsigned short* dst, long long n, void* stream);
int mtts_gemm_bf16_planes(int M, int N, int K, const unsigned short* Ah, int lda, const unsigned short* Bh, int ldb, float* C, unsigned short* Ch,
                          int ldc, const float* bias, float alpha, int flags, int tile, void* stream);
long long mtts_plane_problems(mtts_handle* h);
/* Host-only self check of the task-per-XCD workgroup schedule of 8- / 4- / 2-group launches (csrc/gemm.h: XcdSched): builds the schedule
 * for the `groups` group sizes `dims` (cls 1: M-ragged, units = m-tiles of 64 rows with tn tiles each; cls 2: K-ragged, units_per_group
 * tiles per group whose cost is dims[z]) and walks every workgroup slot.  Returns the number of slots when every (group, tile) is visited
 * exactly once, -1 otherwise; *max_load_permille (optional) = heaviest XCD's work in 1/1000 of a perfect eighth. */
int mtts_xcd_schedule_check(const int* dims, int groups, int cls, int tn, int units_per_group, int* max_load_permille);
/* Conv1d over one zero-guarded sequence, channels-last: x [L][Cin] with >= k/2 zero rows before and
 * after, w [Cout][k][Cin].  mode 0: y = conv(x) + bias; 1: dx = dgrad(dy); 2: dw = wgrad(dy, x) */
int mtts_conv1d_f32(int mode, int L, int Cin, int Cout, int k, const float* x_or_dy, const float* w_or_x, float* out,
                    const float* bias, int tile, void* hip_stream);

/* ---- kernel-level entry points of the HBM-bound (non-GEMM) kernels, so each can be parity-tested alone (dev pointers; rows are
 * dense [rows][C] fp32 matrices, C % 4 == 0, C <= 1024; `ws` = caller-allocated device scratch of mtts_kernel_ws_bytes(rows, n_mat)
 * bytes; nothing is allocated inside; asynchronous on `stream` apart from a <= 64-byte descriptor upload) -------------------
 * layernorm: transformer/SubLayers.py:55,91 + modules.py:222-235 (eps 1e-5): z = a + res (res may be NULL; z may be NULL),
 *   y = LN(z) * gamma + beta on rows with mask != 0 (mask NULL: all), 0 elsewhere; stats[row] = (mean, rstd).  bwd: dz, dgamma, dbeta.
 * softmax / sdpa: transformer/Modules.py:14-25 for n_mat independent (sequence, head) pairs of length L; S, P are
 *   [n_mat][L][ldS], ldS = (L + 3) & ~3; q, k, v, o are [n_mat][L][dk].  softmax_bwd turns dP into dS = alpha * P o (dP - rowsum(dP o P)) in place.
 * batchnorm: PostNet BatchNorm1d in training mode over the rows with inrect != 0 (transformer/Layers.py:129-137), optional tanh;
 *   stats [3C] = mean | rstd | unbiased var.  bwd takes n_in = number of such rows.
 * table_grad: nn.Embedding backward, dtable [V][C] fully written, rows summed in ascending order, skip_row (padding_idx) stays 0. */
int64_t mtts_kernel_ws_bytes(int rows, int n_mat);
int mtts_layernorm_fwd(int rows, int C, const float* a, const float* res, const float* gamma, const float* beta, const unsigned char* mask,
                       float* z, float* y, float* stats, void* ws, void* hip_stream);
int mtts_layernorm_bwd(int rows, int C, const float* dy, const float* z, const float* stats, const float* gamma, const unsigned char* mask,
                       float* dz, float* dgamma, float* dbeta, void* ws, void* hip_stream);
int mtts_softmax_fwd(int n_mat, int L, float* S, void* ws, void* hip_stream);
int mtts_softmax_bwd(int n_mat, int L, const float* P, float* dP, float alpha, void* ws, void* hip_stream);
int mtts_sdpa_fwd(int n_mat, int L, int dk, const float* q, const float* k, const float* v, float* P, float* o, void* ws, void* hip_stream);
int mtts_batchnorm_fwd(int rows, int C, const float* x, const unsigned char* inrect, const float* gamma, const float* beta, int do_tanh,
                       float* stats, float* y, void* ws, void* hip_stream);
int mtts_batchnorm_bwd(int rows, int n_in, int C, const float* dy, const float* y, const float* x, const float* stats, const unsigned char* inrect,
                       const float* gamma, int do_tanh, float* dx, float* dgamma, float* dbeta, void* ws, void* hip_stream);
int mtts_table_grad(int rows, int C, int V, const float* dx, const int* idx, int skip_row, float* dtable, void* ws, void* hip_stream);
/* length_regulate (replaces LengthRegulator.LR / expand, lightning/model/modules.py:167-190, and its autograd transpose):
 *   fwd: out[r][:] = x[src[r]][:] (+ spk[:]) (+ pos[row_t[r]][:]) for r < n_frames, zeros where src[r] < 0; spk [C], pos [*][C] and row_t
 *   may be NULL (plain gather; pos needs row_t).  bwd: dx[p][:] (+)= sum of dout over frame rows [first[p], first[p] + count[p]).
 * layernorm_jvp / softmax_jvp: the forward-mode (tangent) twins the second-order path is built from (csrc/tangent.h), on the z / stats
 *   written by mtts_layernorm_fwd resp. the probabilities written by mtts_softmax_fwd; tgamma / tbeta / tres / mask may be NULL. */
int mtts_length_regulate_fwd(int n_frames, int C, const float* x, const int* src, const float* spk, const float* pos, const int* row_t, float* out,
                             void* ws, void* hip_stream);
int mtts_length_regulate_bwd(int n_phonemes, int C, const float* dout, const int* first, const int* count, float* dx, int accumulate, void* ws,
                             void* hip_stream);
int mtts_layernorm_jvp(int rows, int C, const float* ta, const float* tres, const float* z, const float* stats, const float* gamma, const float* tgamma,
                       const float* tbeta, const unsigned char* mask, float* ty, void* ws, void* hip_stream);
int mtts_softmax_jvp(int n_mat, int L, const float* P, float* tS, void* ws, void* hip_stream);

/* ---- MelGAN generator: mel -> waveform --------------------------------------------
 * Replaces `LightningMelGAN.inverse / infer`, lightning/utils.py:8-30 (vocoder.mel2wav of torch.hub
 * "descriptinc/melgan-neurips"; the generator is an un-vendored dependency: architecture of that hub entry, weights
 * supplied by the caller).  Tensors (weight-norm already folded, see meta_tts_amd/vocoder.py):
 *   conv_in.w [C0][7][n_mel], conv_in.b [C0];  up{s}.w [r][C_out][2*C_in] (phase-major polyphase image of the
 *   ConvTranspose1d), up{s}.b;  res{s}.{j}.w1 [C][3][C], .b1, .w2 [C][C], .b2, .ws [C][C], .bs;  conv_out.w [7][C_last],
 *   conv_out.b [1];  C0 = ngf << n_ratios.
 * infer: mel [B][T_max][n_mel] (host), mel_lens[b] frames valid, every value multiplied by mel_scale (the reference
 * passes mel / ln 10) -> wav [B][T_max * hop] floats in (-1, 1), the first mel_lens[b] * hop samples of a row written.
 * infer_device: same with device pointers (no copies, no synchronisation; runs on the vocoder's stream); mel_utt_stride =
 * floats between consecutive utterances of mel_dev (0: T_max * n_mel) so that the engine's own mel buffer can be fed as it
 * is (mtts_get_mel_device). */
typedef struct mtts_vocoder mtts_vocoder;
int mtts_vocoder_create(int n_mel, int ngf, int n_res, const int* ratios, int n_ratios, int device, int max_B, int max_T,
                        mtts_vocoder** out);
void mtts_vocoder_destroy(mtts_vocoder* h);
const char* mtts_vocoder_last_error(mtts_vocoder* h);
int mtts_vocoder_set_stream(mtts_vocoder* h, void* hip_stream);
int mtts_vocoder_hop(mtts_vocoder* h);
int mtts_vocoder_param_count(mtts_vocoder* h);
int mtts_vocoder_param_info(mtts_vocoder* h, int index, char* name, int name_cap, int64_t* numel);
int mtts_vocoder_load(mtts_vocoder* h, const char* name, const float* data, int64_t numel);
int mtts_vocoder_infer(mtts_vocoder* h, const float* mel, int B, int T_max, const int* mel_lens, float mel_scale, float* wav);
int mtts_vocoder_infer_device(mtts_vocoder* h, const float* mel_dev, int64_t mel_utt_stride, int B, int T_max, const int* mel_lens,
                              float mel_scale, float* wav_dev);

/* ---- d-vector speaker encoder, forward only (`speaker_emb: dvec`) -------------------------
 * Replaces `SpeakerEncoder.forward` for emb_type "dvec" (lightning/model/speaker_encoder.py:54-60,71-76): the frozen
 * resemblyzer `VoiceEncoder` (un-vendored; architecture restated by the reference's own GE2E class, :11-31: LSTM(n_mels -> hidden,
 * `layers` layers, batch_first) + Linear(hidden -> emb) + ReLU), applied to the partial utterances of the batch
 * (`spk_ref_mel_slices`, lightning/collate.py:29-43): partial embedding = L2-normalised ReLU(Linear(final hidden state of the last
 * layer)); utterance embedding = F.normalize(mean of its partials).  The trained variants ("encoder", "scratch_encoder") are the
 * mtts_dvector_enable_training / embed_train / backward / adam_step entries further down.
 * Tensors (torch names and layouts): lstm.weight_ih_l{k} [4*hidden][in], lstm.weight_hh_l{k} [4*hidden][hidden],
 * lstm.bias_ih_l{k}, lstm.bias_hh_l{k} [4*hidden] (gate order i, f, g, o), linear.weight [emb][hidden], linear.bias [emb].
 * embed: mels [n_partials][frames][n_mels] (host), utt_offsets [n_utts + 1] (partials of utterance b = [off[b], off[b+1]),
 * off[0] = 0, off[n_utts] = n_partials) -> out [n_utts][emb] (host; feed it to mtts_batch.spk_emb); partial_out
 * [n_partials][emb] or NULL.  Synchronous. */
typedef struct mtts_dvector mtts_dvector;
int mtts_dvector_create(int n_mels, int hidden, int layers, int emb, int max_partials, int frames, int max_utts, int device, mtts_dvector** out);
void mtts_dvector_destroy(mtts_dvector* h);
/* HIP stream of this handle's launches (default: the null stream).  A trained encoder exchanges device scalars with the engine (its
 * gradient's sum of squares -> mtts_set_extra_grad_sumsq, the joint norm <- mtts_grad_norm_dev): put both handles on ONE stream. */
int mtts_dvector_set_stream(mtts_dvector* h, void* hip_stream);
const char* mtts_dvector_last_error(mtts_dvector* h);
int mtts_dvector_load(mtts_dvector* h, const char* name, const float* data, int64_t numel);
int mtts_dvector_embed(mtts_dvector* h, const float* mels, int n_partials, const int* utt_offsets, int n_utts, float* out, float* partial_out);
/* Trained speaker encoders (`speaker_emb: encoder` / `scratch_encoder`, config/algorithm/{encoder,scratch_encoder}.yaml: baseline
 * systems whose optimizer also owns the LSTM; speaker_encoder.py:54-55,59-60).  enable_training allocates the saved-state, gradient
 * and Adam buffers.  embed_train = embed that keeps the gate activations / cell states for the backward sweep; backward takes
 * dout [n_utts][emb] (host) = dLoss/d(embedding) — mtts_get_speaker_grad of the engine — and fills the parameter gradients by
 * back-propagation through time (hand-derived; autograd of nn.LSTM in the reference).  grad_sumsq returns a DEVICE scalar, the sum
 * of squares of those gradients, to be handed to mtts_set_extra_grad_sumsq so that mtts_outer_update clips by the joint norm of all
 * parameters (main.py:61); adam_step then applies the same clip coefficient (norm_dev = mtts_grad_norm_dev(engine), NULL: no clip)
 * and the Adam rule of optimizer.py:9-15.  export / import: which = 0 parameter, 1 gradient, 2 Adam m, 3 Adam v. */
int mtts_dvector_enable_training(mtts_dvector* h);
int mtts_dvector_embed_train(mtts_dvector* h, const float* mels, int n_partials, const int* utt_offsets, int n_utts, float* out);
int mtts_dvector_backward(mtts_dvector* h, const float* dout);
const float* mtts_dvector_grad_sumsq(mtts_dvector* h);
int mtts_dvector_adam_step(mtts_dvector* h, const float* norm_dev, float max_norm, float lr, float b1, float b2, float eps, float weight_decay);
int mtts_dvector_set_optimizer_step(mtts_dvector* h, int step);
int mtts_dvector_export(mtts_dvector* h, const char* name, int which, float* out, int64_t numel);
int mtts_dvector_import(mtts_dvector* h, const char* name, int which, const float* data, int64_t numel);

/* ---- waveform -> log-mel spectrogram + frame energy (front-end STFT / mel extraction) ------
 * Replaces `TacotronSTFT.mel_spectrogram` behind `Audio.tools.get_mel_from_wav` (audio/stft.py:128-178, audio/tools.py:8-15):
 * clip to [-1, 1], reflect-pad filter_length / 2, windowed DFT of every hop-spaced frame (STFT.transform, stft.py:52-77), magnitude,
 * mel = log(max(mel_basis @ magnitude, 1e-5)), energy = ||magnitude||_2 per frame.
 * load: forward_basis [2 * (filter_length / 2 + 1)][filter_length] = the reference's `STFT.forward_basis` buffer (real rows then
 * imaginary rows, window folded in, stft.py:27-46); mel_basis [n_mel][filter_length / 2 + 1] = `TacotronSTFT.mel_basis`
 * (librosa.filters.mel, stft.py:143-147).  Either may be NULL to keep what was loaded before.
 * mel_spectrogram: wav [n_samples] (host) -> mel [T][n_mel] (host; the reference returns the transpose, (n_mel, T)), energy [T],
 * T = n_samples / hop_length + 1; returns T, or < 0 on error.  Synchronous. */
typedef struct mtts_stft mtts_stft;
int mtts_stft_create(int filter_length, int hop_length, int n_mel, int max_samples, int device, mtts_stft** out);
void mtts_stft_destroy(mtts_stft* h);
int mtts_stft_set_stream(mtts_stft* h, void* hip_stream);
const char* mtts_stft_last_error(mtts_stft* h);
int mtts_stft_load(mtts_stft* h, const float* forward_basis, const float* mel_basis);
int mtts_stft_mel_spectrogram(mtts_stft* h, const float* wav, int n_samples, float* mel, float* energy);

#ifdef __cplusplus
}
#endif
#endif /* MTTS_H */
