"""ORACLE — test infrastructure, not product code.

CPU restatement (torch fp32, functional, autograd for the derivatives) of the Meta-TTS hot
path: FastSpeech2 forward, its 5-term loss, the MAML inner/outer loop and the outer
Adam/Noam/clip update.  Every function cites the reference file:line it follows
(paths relative to /root/reference).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product path
(``meta_tts_amd``) never does and fails loudly without its HIP library.

Pinning: ``tests/golden/make_golden.py`` imports the reference's own ``FastSpeech2`` /
``FastSpeech2Loss`` in the build container and stores inputs + outputs as fixtures;
``tests/test_oracle_golden.py`` checks this file against them.  The MAML update rule lives in
un-vendored learn2learn (requirements.txt:2, absent here) => that single rule is restated from
its published definition (theta' = theta - lr * grad, create_graph = second order) and is
"parity unpinned" against learn2learn itself; the model arithmetic underneath it is pinned.

Parameters are a dict {reference state_dict name (without the ``model.`` prefix): tensor}.
Dropout: the identity by default (the fixture configuration).  With ``dropout=`` a
``oracle.dropout_masks.DropoutMasks`` the reference's five dropout sites (SubLayers.py:54,90, modules.py:223,235,
Layers.py:133-134) multiply by keep / (1 - p) with the HIP engine's own counter-based masks — the configuration bench.py times.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# ReLU kinks.  The loss is piecewise smooth: at a ReLU whose pre-activation is within fp32 noise of zero two correct fp32
# implementations may land on different sides, and their GRADIENTS then differ by that unit's whole contribution (a few rows of a
# few tensors) although every forward value agrees to 1e-7.  A caller that compares gradients can ask which passes had such units:
# with KINK_LOG a list, every ReLU of the model appends (min |pre-activation|, number of units with |pre-activation| < KINK_EPS).
# The two L1 mel terms of the loss have the same property at |prediction - target| ~ 0: d|x|/dx = sign(x).  A weight gradient behind
# them is a sum of n random-sign terms (n = valid frames x n_mel ~ 1.7e5 per task), so ONE flipped sign moves it by ~2 / sqrt(n) = 0.5 %
# of its norm; mel_post agrees between two fp32 implementations to ~2e-5 after five inner steps, which puts about one element per task
# and pass inside that band (measured: tools/dropout_grad_probe.py).  fs2_loss appends ("l1", elements of mel / mel_post with
# |prediction - target| < L1_KINK_EPS) when asked.
KINK_LOG: Optional[list] = None
# With RELU_TAPS a list every ReLU appends (pre-activation, output) — graph tensors, so that a caller (oracle/arbiter.py) can price the
# gradient contribution of single units whose pre-activation sits inside fp32 noise of zero.
RELU_TAPS: Optional[list] = None
KINK_EPS = 1e-6
L1_KINK_EPS = 1e-4


# bf16-operand mode (the engine's numerics mode 1 / 2, BASELINE config C2 "bf16"): BOTH operands of every contraction — Linear, Conv1d,
# the two attention products — are rounded to bf16 (round-to-nearest-even) on the way into the product, forward AND backward (the
# input gradient multiplies round(dY) by round(W), the weight gradient round(dY) by round(X)); accumulation, bias, normalisations,
# softmax, losses and every stored tensor stay fp32.  With BF16_OPERANDS set the three product helpers below do exactly that through
# custom autograd Functions, which gives the bf16 mode a MODEL-level reference that rounds what the mode rounds —
# the fp32 oracle stays the distance that is reported, not the gate.
BF16_OPERANDS = False


def _r16(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _LinearBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        y = F.linear(_r16(x), _r16(w))
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gr = _r16(g)
        gx = gr @ _r16(w)
        gw = gr.reshape(-1, gr.shape[-1]).t() @ _r16(x).reshape(-1, x.shape[-1])
        gb = g.reshape(-1, g.shape[-1]).sum(0) if ctx.has_b else None      # (the bias gradient is a column sum of the fp32 dY, not a product)
        return gx, gw, gb


class _Conv1dBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, padding):
        ctx.save_for_backward(x, w)
        ctx.padding, ctx.has_b = padding, b is not None
        return F.conv1d(_r16(x), _r16(w), b, padding=padding)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gr = _r16(g)
        gx = torch.nn.grad.conv1d_input(x.shape, _r16(w), gr, padding=ctx.padding)
        gw = torch.nn.grad.conv1d_weight(_r16(x), w.shape, gr, padding=ctx.padding)
        gb = g.sum((0, 2)) if ctx.has_b else None
        return gx, gw, gb, None


class _BmmBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return torch.bmm(_r16(a), _r16(b))

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gr = _r16(g)
        return torch.bmm(gr, _r16(b).transpose(1, 2)), torch.bmm(_r16(a).transpose(1, 2), gr)


def _linear(x, w, b=None):
    return _LinearBF16.apply(x, w, b) if BF16_OPERANDS else F.linear(x, w, b)


def _conv1d(x, w, b=None, padding=0):
    return _Conv1dBF16.apply(x, w, b, padding) if BF16_OPERANDS else F.conv1d(x, w, b, padding=padding)


def _bmm(a, b):
    return _BmmBF16.apply(a, b) if BF16_OPERANDS else torch.bmm(a, b)


def _relu(x):
    if KINK_LOG is not None:
        a = x.detach().abs()
        KINK_LOG.append((float(a.min()), int((a < KINK_EPS).sum())))
    if RELU_TAPS is not None:
        y = F.relu(x)
        RELU_TAPS.append((x, y))
        return y
    return F.relu(x)


# --------------------------------------------------------------------------------------
# masks / tables
# --------------------------------------------------------------------------------------
def mask_from_lengths(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """utils/tools.py:91-99 — True marks padding: arange(max_len) >= len[:, None]."""
    if max_len is None:
        max_len = int(lengths.max().item())
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0)
    return ids >= lengths.unsqueeze(1)


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """transformer/Models.py:10-30 (float64 numpy arithmetic, cast to fp32)."""
    pos = torch.arange(n_position, dtype=torch.float64).unsqueeze(1)
    j = torch.arange(d_hid, dtype=torch.float64).unsqueeze(0)
    table = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2.0 * torch.floor(j / 2.0) / d_hid)
    table[:, 0::2] = torch.sin(table[:, 0::2])
    table[:, 1::2] = torch.cos(table[:, 1::2])
    return table.to(torch.float32)


# --------------------------------------------------------------------------------------
# transformer blocks
# --------------------------------------------------------------------------------------
def scaled_dot_product_attention(q, k, v, mask, temperature):
    """transformer/Modules.py:14-25 — softmax(masked_fill(q k^T / temperature, mask, -inf), dim=2) v."""
    attn = _bmm(q, k.transpose(1, 2)) / temperature
    attn = attn.masked_fill(mask, float("-inf"))
    attn = torch.softmax(attn, dim=2)
    return _bmm(attn, v), attn


def _drop(x, dropout, site: int, space: str, which: str):
    """One of the reference's nn.Dropout / F.dropout sites; identity without a mask source."""
    return x if dropout is None else dropout.apply(x, site, space, dropout.probs[which])


def multi_head_attention(x, p: Params, pre: str, slf_attn_mask, n_head: int, dropout=None, site: int = 0, space: str = "P"):
    """transformer/SubLayers.py:29-57 — head-major split (permute(2,0,1,3)), mask repeated per
    head, fc, dropout (:54), post-LayerNorm over (out + residual), eps 1e-5."""
    B, L, d = x.shape
    d_k = d // n_head
    q = _linear(x, p[f"{pre}.w_qs.weight"], p[f"{pre}.w_qs.bias"]).view(B, L, n_head, d_k)
    k = _linear(x, p[f"{pre}.w_ks.weight"], p[f"{pre}.w_ks.bias"]).view(B, L, n_head, d_k)
    v = _linear(x, p[f"{pre}.w_vs.weight"], p[f"{pre}.w_vs.bias"]).view(B, L, n_head, d_k)
    q = q.permute(2, 0, 1, 3).contiguous().view(-1, L, d_k)
    k = k.permute(2, 0, 1, 3).contiguous().view(-1, L, d_k)
    v = v.permute(2, 0, 1, 3).contiguous().view(-1, L, d_k)
    mask = slf_attn_mask.repeat(n_head, 1, 1)
    out, attn = scaled_dot_product_attention(q, k, v, mask, math.sqrt(d_k))
    out = out.view(n_head, B, L, d_k).permute(1, 2, 0, 3).contiguous().view(B, L, -1)
    out = _linear(out, p[f"{pre}.fc.weight"], p[f"{pre}.fc.bias"])
    out = _drop(out, dropout, site, space, "enc" if space == "P" else "dec")
    out = F.layer_norm(out + x, (d,), p[f"{pre}.layer_norm.weight"], p[f"{pre}.layer_norm.bias"], 1e-5)
    return out, attn


def positionwise_ffn(x, p: Params, pre: str, dropout=None, site: int = 0, space: str = "P"):
    """transformer/SubLayers.py:85-93 — LN(dropout(W2 * relu(W1 * x)) + x) with Conv1d W1 (k=9,pad=4), W2 (k=1)."""
    w1, w2 = p[f"{pre}.w_1.weight"], p[f"{pre}.w_2.weight"]
    h = _conv1d(x.transpose(1, 2), w1, p[f"{pre}.w_1.bias"], padding=(w1.shape[2] - 1) // 2)
    h = _conv1d(_relu(h), w2, p[f"{pre}.w_2.bias"], padding=(w2.shape[2] - 1) // 2)
    h = _drop(h.transpose(1, 2), dropout, site, space, "enc" if space == "P" else "dec")
    d = x.shape[-1]
    return F.layer_norm(h + x, (d,), p[f"{pre}.layer_norm.weight"], p[f"{pre}.layer_norm.bias"], 1e-5)


def fft_block(x, p: Params, pre: str, mask, slf_attn_mask, n_head: int, dropout=None, site: int = 0, space: str = "P"):
    """transformer/Layers.py:21-30 — MHA, zero padded rows, FFN, zero padded rows.  Mask-stream sites: `site` (attention), `site + 1` (FFN)."""
    x, attn = multi_head_attention(x, p, f"{pre}.slf_attn", slf_attn_mask, n_head, dropout, site, space)
    x = x.masked_fill(mask.unsqueeze(-1), 0)
    x = positionwise_ffn(x, p, f"{pre}.pos_ffn", dropout, site + 1, space)
    x = x.masked_fill(mask.unsqueeze(-1), 0)
    return x, attn


def _n_layers(p: Params, prefix: str) -> int:
    n = 0
    while f"{prefix}.layer_stack.{n}.slf_attn.w_qs.weight" in p:
        n += 1
    return n


def encoder(texts, mask, p: Params, n_head: int, training: bool, max_seq_len: int, dropout=None):
    """transformer/Models.py:73-100 — word embedding (pad row 0) + sinusoid positions, 4 FFT blocks."""
    B, L = texts.shape
    slf_attn_mask = mask.unsqueeze(1).expand(-1, L, -1)
    emb = F.embedding(texts, p["encoder.src_word_emb.weight"], padding_idx=0)
    d = emb.shape[-1]
    if (not training) and L > max_seq_len:
        pos = sinusoid_table(L, d)[:L].unsqueeze(0)
    else:
        pos = p["encoder.position_enc"][:, :L, :]
    x = emb + pos.expand(B, -1, -1)
    for i in range(_n_layers(p, "encoder")):
        x, _ = fft_block(x, p, f"encoder.layer_stack.{i}", mask, slf_attn_mask, n_head, dropout, 2 * i, "P")
    return x


def decoder(x, mask, p: Params, n_head: int, training: bool, max_seq_len: int, dropout=None):
    """transformer/Models.py:139-171 — training truncates to max_seq_len, eval extends the table."""
    B, L, d = x.shape
    if (not training) and L > max_seq_len:
        slf_attn_mask = mask.unsqueeze(1).expand(-1, L, -1)
        x = x + sinusoid_table(L, d)[:L].unsqueeze(0).expand(B, -1, -1)
    else:
        L = min(L, max_seq_len)
        slf_attn_mask = mask.unsqueeze(1).expand(-1, L, -1)
        x = x[:, :L, :] + p["decoder.position_enc"][:, :L, :].expand(B, -1, -1)
        mask = mask[:, :L]
        slf_attn_mask = slf_attn_mask[:, :, :L]
    for i in range(_n_layers(p, "decoder")):
        x, _ = fft_block(x, p, f"decoder.layer_stack.{i}", mask, slf_attn_mask, n_head, dropout, 64 + 2 * i, "F")
    return x, mask


# --------------------------------------------------------------------------------------
# variance adaptor
# --------------------------------------------------------------------------------------
def variance_predictor(x, mask, p: Params, pre: str, dropout=None, site: int = 128, space: str = "P"):
    """lightning/model/modules.py:242-250 (+ Conv :253-296) — [Conv1d k=3 -> ReLU -> LN -> Dropout]x2 -> Linear(->1)
    -> masked_fill(mask, 0)."""
    for i in (1, 2):
        w = p[f"{pre}.conv_layer.conv1d_{i}.conv.weight"]
        pad = (w.shape[2] - 1) // 2 if i == 1 else 1
        x = _conv1d(x.transpose(1, 2), w, p[f"{pre}.conv_layer.conv1d_{i}.conv.bias"], padding=pad).transpose(1, 2)
        x = _relu(x)
        x = F.layer_norm(x, (x.shape[-1],), p[f"{pre}.conv_layer.layer_norm_{i}.weight"],
                         p[f"{pre}.conv_layer.layer_norm_{i}.bias"], 1e-5)
        x = _drop(x, dropout, site + i - 1, space, "vp")
    out = F.linear(x, p[f"{pre}.linear_layer.weight"], p[f"{pre}.linear_layer.bias"]).squeeze(-1)   # (a 256 -> 1 row dot product: fp32 in every numerics mode of the engine)
    if mask is not None:
        out = out.masked_fill(mask, 0.0)
    return out


def length_regulate(x, durations, max_len: Optional[int]):
    """lightning/model/modules.py:167-190 + utils/tools.py:304-322 — repeat row i max(int(dur[i]),0)
    times, concatenate, zero-pad to max_len (or the batch max).  Returns (out, mel_len[int64])."""
    B = x.shape[0]
    reps = durations.to(torch.int64).clamp(min=0)
    outs, lens = [], []
    for b in range(B):
        e = torch.repeat_interleave(x[b], reps[b], dim=0)
        outs.append(e)
        lens.append(e.shape[0])
    T = max_len if max_len else max(lens)
    out = torch.stack([F.pad(e, (0, 0, 0, T - e.shape[0])) for e in outs])
    return out, torch.tensor(lens, dtype=torch.int64)


def variance_adaptor(x, src_mask, mel_mask, max_len, p_t, e_t, d_t, p: Params,
                     p_control=1.0, e_control=1.0, d_control=1.0, pitch_level="phoneme_level", energy_level="phoneme_level", dropout=None):
    """lightning/model/modules.py:102-158: duration predictor on x; phoneme-level features (:118-127): pitch predictor on x,
    x += pitch_emb[bucketize(target or pred*ctl, bins)]; energy predictor on the updated x, x += energy_emb[...]; length
    regulation with the targets or clamp(round(exp(logd) - 1) * ctl, 0); frame-level features (:139-148): the same two steps
    AFTER the length regulator, on the zero-padded frame rectangle with the mel mask."""
    va = "variance_adaptor"

    def embed(x, target, mask, control, name, space):        # get_pitch_embedding / get_energy_embedding, modules.py:80-100
        pred = variance_predictor(x, mask, p, f"{va}.{name}_predictor", dropout, 132 if name == "pitch" else 136, space)
        if target is not None:
            idx = torch.bucketize(target, p[f"{va}.{name}_bins"])
        else:
            pred = pred * control
            idx = torch.bucketize(pred, p[f"{va}.{name}_bins"])
        return pred, F.embedding(idx, p[f"{va}.{name}_embedding.weight"])

    logd = variance_predictor(x, src_mask, p, f"{va}.duration_predictor", dropout, 128, "P")
    pp = ep = None
    if pitch_level == "phoneme_level":
        pp, emb = embed(x, p_t, src_mask, p_control, "pitch", "P")
        x = x + emb
    if energy_level == "phoneme_level":
        ep, emb = embed(x, e_t, src_mask, e_control, "energy", "P")
        x = x + emb
    if d_t is not None:
        x, mel_len = length_regulate(x, d_t, max_len)
        d_rounded = d_t
    else:
        d_rounded = torch.clamp(torch.round(torch.exp(logd) - 1) * d_control, min=0)
        x, mel_len = length_regulate(x, d_rounded, max_len)
        mel_mask = mask_from_lengths(mel_len)
    if pitch_level == "frame_level":
        pp, emb = embed(x, p_t, mel_mask, p_control, "pitch", "R")
        x = x + emb
    if energy_level == "frame_level":
        ep, emb = embed(x, e_t, mel_mask, e_control, "energy", "R")
        x = x + emb
    return x, pp, ep, logd, d_rounded, mel_len, mel_mask


# --------------------------------------------------------------------------------------
# PostNet
# --------------------------------------------------------------------------------------
def postnet(x, p: Params, buffers: Optional[Dict[str, torch.Tensor]], training: bool, dropout=None):
    """transformer/Layers.py:129-137 — NCL: 4x[Conv1d k=5 -> BatchNorm1d -> tanh -> F.dropout 0.5] + [Conv -> BN -> F.dropout 0.5]
    (:133-134).  BatchNorm in training mode uses batch statistics over all B*T_max
    positions (padding included) and updates running stats with momentum 0.1 (unbiased var)."""
    h = x.transpose(1, 2)
    n = 0
    while f"postnet.convolutions.{n}.0.conv.weight" in p:
        n += 1
    for i in range(n):
        pre = f"postnet.convolutions.{i}"
        w = p[f"{pre}.0.conv.weight"]
        h = _conv1d(h, w, p[f"{pre}.0.conv.bias"], padding=(w.shape[2] - 1) // 2)
        rm = rv = None
        if buffers is not None:
            rm, rv = buffers[f"{pre}.1.running_mean"], buffers[f"{pre}.1.running_var"]
        if training or rm is None:
            h = F.batch_norm(h, rm, rv, p[f"{pre}.1.weight"], p[f"{pre}.1.bias"], True, 0.1, 1e-5)
            if buffers is not None and f"{pre}.1.num_batches_tracked" in buffers:
                buffers[f"{pre}.1.num_batches_tracked"] += 1
        else:
            h = F.batch_norm(h, rm, rv, p[f"{pre}.1.weight"], p[f"{pre}.1.bias"], False, 0.1, 1e-5)
        if i < n - 1:
            h = torch.tanh(h)
        if dropout is not None:
            h = _drop(h.transpose(1, 2), dropout, 192 + i, "R", "postnet").transpose(1, 2)
    return h.transpose(1, 2)


# --------------------------------------------------------------------------------------
# full model + loss
# --------------------------------------------------------------------------------------
def fs2_forward(p: Params, buffers, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None,
                max_mel_len=None, p_targets=None, e_targets=None, d_targets=None,
                p_control=1.0, e_control=1.0, d_control=1.0, *, n_head=(2, 2), max_seq_len=1000,
                training=False, average_spk_emb=False, pitch_level="phoneme_level", energy_level="phoneme_level", dropout=None):
    """lightning/model/fastspeech2.py:40-112 and, with ``average_spk_emb``, the learner variant
    lightning/systems/base_adaptor.py:41-95 (mean of the support speakers' rows, expanded).
    ``dropout``: a DropoutMasks (train-mode, teacher-forced passes only) or None = identity."""
    src_masks = mask_from_lengths(src_lens, max_src_len)
    mel_masks = mask_from_lengths(mel_lens, max_mel_len) if mel_lens is not None else None
    if dropout is not None:
        assert training and mel_lens is not None, "dropout masks: train-mode teacher-forced passes"
        dropout.bind(max_src_len, mel_lens.tolist(), min(int(max_mel_len), max_seq_len))
    out = encoder(texts, src_masks, p, n_head[0], training, max_seq_len, dropout)
    spk = F.embedding(speakers, p["speaker_emb.model.weight"])
    if average_spk_emb:
        spk = spk.mean(dim=0, keepdim=True).expand(out.shape[0], -1)
    out = out + spk.unsqueeze(1).expand(-1, max_src_len, -1)
    out, pp, ep, logd, d_rounded, mel_lens, mel_masks = variance_adaptor(
        out, src_masks, mel_masks, max_mel_len, p_targets, e_targets, d_targets, p,
        p_control, e_control, d_control, pitch_level, energy_level, dropout)
    out = out + spk.unsqueeze(1).expand(-1, out.shape[1], -1)
    out, mel_masks = decoder(out, mel_masks, p, n_head[1], training, max_seq_len, dropout)
    mel = _linear(out, p["mel_linear.weight"], p["mel_linear.bias"])
    mel_post = postnet(mel, p, buffers, training, dropout) + mel
    return (mel, mel_post, pp, ep, logd, d_rounded, src_masks, mel_masks, src_lens, mel_lens)


def fs2_loss(batch, preds, pitch_level="phoneme_level", energy_level="phoneme_level", kink_log: Optional[list] = None):
    """lightning/model/loss.py:19-92 — L1 over valid frames x n_mel for mel / postnet mel, MSE over
    valid phonemes (or valid frames for a frame-level feature, :54-63) for pitch and energy, over valid
    phonemes for log-duration (target log(d + 1)); total = plain sum."""
    mel_t, _, _, p_t, e_t, d_t = batch[6:]
    mel, mel_post, pp, ep, logd, _, src_masks, mel_masks, _, _ = preds
    sm, mm = ~src_masks, ~mel_masks
    logd_t = torch.log(d_t.to(logd.dtype) + 1)      # (loss.py:30: .float(); the fp64 arbiter runs the same code in double)
    mel_t = mel_t[:, : mm.shape[1], :]
    mel_l = F.l1_loss(mel.masked_select(mm.unsqueeze(-1)), mel_t.masked_select(mm.unsqueeze(-1)))
    post_l = F.l1_loss(mel_post.masked_select(mm.unsqueeze(-1)), mel_t.masked_select(mm.unsqueeze(-1)))
    if kink_log is not None:
        with torch.no_grad():
            tsel = mel_t.masked_select(mm.unsqueeze(-1))
            kink_log.append(("l1", int(((mel.masked_select(mm.unsqueeze(-1)) - tsel).abs() < L1_KINK_EPS).sum()),
                             int(((mel_post.masked_select(mm.unsqueeze(-1)) - tsel).abs() < L1_KINK_EPS).sum())))
    pm = sm if pitch_level == "phoneme_level" else mm
    em = sm if energy_level == "phoneme_level" else mm
    p_l = F.mse_loss(pp.masked_select(pm), p_t.masked_select(pm))
    e_l = F.mse_loss(ep.masked_select(em), e_t.masked_select(em))
    d_l = F.mse_loss(logd.masked_select(sm), logd_t.masked_select(sm))
    total = mel_l + post_l + d_l + p_l + e_l
    return (total, mel_l, post_l, p_l, e_l, d_l)


def to_torch_batch(batch):
    """numpy 12-tuple (meta_tts_amd.synth.make_batch) -> torch 12-tuple (lightning/collate.py:47-60)."""
    out = []
    for i, x in enumerate(batch):
        if hasattr(x, "dtype") and hasattr(x, "shape") and not isinstance(x, torch.Tensor):
            out.append(torch.from_numpy(x))
        else:
            out.append(int(x) if i in (5, 8) else x)
    return tuple(out)


# --------------------------------------------------------------------------------------
# MAML (learn2learn.algorithms.MAML restated; call sites lightning/systems/utils.py:17-77,
# lightning/systems/base_adaptor.py:98-124)
# --------------------------------------------------------------------------------------
def adapted_names(p: Params, modules: Sequence[str]) -> List[str]:
    """Tensors the inner loop updates: requires_grad parameters of the adapted sub-modules
    (base_adaptor.py:31-35; position_enc / *_bins are frozen so l2l skips them)."""
    frozen = ("position_enc", "pitch_bins", "energy_bins")
    return [k for k in p if k.split(".")[0] in modules and not k.endswith(frozen)]


def maml_task(p: Params, buffers, sup, qry, *, steps: int, lr: float, second_order: bool,
              modules: Sequence[str], n_head=(2, 2), max_seq_len=1000, training=True, dropout=None, kink_log: Optional[list] = None):
    """``dropout``: None, or a list of steps + 1 DropoutMasks (one per inner step, then the query pass — the order in which the
    engine draws its plan seeds, engine.h: meta_grad / run_encoder_ahead).  ``kink_log``: a list that receives one
    (min |pre-activation|, units below KINK_EPS) pair per ReLU of the QUERY pass and one ("l1", mel elements, mel_post elements within
    L1_KINK_EPS of the target) entry for its loss.

    One task of a meta-step: ``steps`` inner SGD updates on the support batch
    (base_adaptor.py:100-112, ``first_order = not train``), then the query pass with the support
    speaker ids and ``average_spk_emb=True`` (base_adaptor.py:114-124).  ``p`` tensors that should
    receive outer gradients must have requires_grad=True.  Returns
    (query 6-tuple, [support loss per step], fast weights, query predictions)."""
    names = adapted_names(p, modules)
    fast = {k: p[k] for k in names}
    sup_losses = []
    for s_ in range(steps):
        cur = dict(p)
        cur.update(fast)
        preds = fs2_forward(cur, buffers, *sup[2:], n_head=n_head, max_seq_len=max_seq_len, training=training,
                            dropout=dropout[s_] if dropout else None)
        loss = fs2_loss(sup, preds)
        sup_losses.append(loss)
        grads = torch.autograd.grad(loss[0], [fast[k] for k in names], create_graph=second_order,
                                    allow_unused=False)
        fast = {k: fast[k] - lr * g for k, g in zip(names, grads)}
        if not second_order:
            # first-order MAML: l2l detaches nothing but the graph of g is not kept; the update
            # theta' = theta - lr*g stays differentiable wrt theta with d theta'/d theta = I.
            pass
    cur = dict(p)
    cur.update(fast)
    global KINK_LOG
    KINK_LOG = kink_log          # (the query pass only: its ReLU kinks are the ones a first-order outer gradient can see)
    try:
        preds = fs2_forward(cur, buffers, sup[2], *qry[3:], n_head=n_head, max_seq_len=max_seq_len,
                            training=training, average_spk_emb=True, dropout=dropout[steps] if dropout else None)
    finally:
        KINK_LOG = None
    qloss = fs2_loss(qry, preds, kink_log=kink_log)
    return qloss, sup_losses, fast, preds


# --------------------------------------------------------------------------------------
# iMAML (lightning/systems/imaml.py:41-139, lightning/systems/utils.py:120-189).  PARITY UNPINNED at two
# un-vendored dependencies: learn2learn's `adapt` (plain SGD step, as above) and `hypergrad.CG_torch.cg` /
# `hypergrad.CG` (the `hypertorch` submodule is empty in the reference tree): the conjugate-gradient
# recurrence below restates that package's published implementation — x0 = 0, r0 = p0 = b, alpha = rTr / pAp,
# the loop breaks BEFORE adopting an iterate whose residual norm is below tol — computed exactly as utils.py
# drives it: A v = v - (dPhi/dw)^T v through autograd of the fixed-point map Phi.
# --------------------------------------------------------------------------------------
def imaml_task(p: Params, buffers, sup_batches: Sequence, cg_batches: Sequence, qry, sup_ids, *, lr: float,
               reg_param: float, K: int, modules: Sequence[str], tol: float = 1e-10, n_head=(2, 2),
               max_seq_len=1000, training=True):
    """One iMAML task.  ``sup_batches``: the support mini-batch of every inner step (Task.next_batch,
    imaml.py:64-70); ``cg_batches``: the mini-batch of every fixed-point-map evaluation inside CG
    (`stochastic`, imaml.py:88-91; K of them are used); ``sup_ids``: the support speaker ids of the query pass.
    Returns (query 6-tuple, fast weights after the inner loop, {name: hypergradient} for the adapted tensors,
    CG solution v)."""
    names = adapted_names(p, modules)
    theta = {k: p[k].detach() for k in names}

    def reg(fast):
        return sum(((theta[k] - fast[k]) ** 2).sum() for k in names)   # bias_reg_f, imaml.py:41-46

    def sup_loss(fast, batch):
        cur = dict(p); cur.update(fast)
        preds = fs2_forward(cur, buffers, *batch[2:], n_head=n_head, max_seq_len=max_seq_len, training=training)
        return fs2_loss(batch, preds)[0] + 0.5 * reg_param * reg(fast)

    fast = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    for batch in sup_batches:                                   # first-order regularised adapt, imaml.py:66-70
        g = torch.autograd.grad(sup_loss(fast, batch), [fast[k] for k in names])
        fast = {k: (fast[k] - lr * gi).detach().requires_grad_(True) for k, gi in zip(names, g)}
    w = [fast[k] for k in names]
    cur = dict(p); cur.update(fast)
    preds = fs2_forward(cur, buffers, sup_ids, *qry[3:], n_head=n_head, max_seq_len=max_seq_len, training=training,
                        average_spk_emb=True)
    qloss = fs2_loss(qry, preds)
    b = [x.detach() for x in torch.autograd.grad(qloss[0], w)]  # grad_outer_w, utils.py:155

    def fp_map(batch):                                          # imaml.py:82-98: one differentiable SGD step on the regularised loss
        g = torch.autograd.grad(sup_loss(fast, batch), w, create_graph=True)
        return [wi - lr * gi for wi, gi in zip(w, g)]

    it = iter(cg_batches)

    def A(xs):                                                  # dfp_map_dw, utils.py:160-170
        J = torch.autograd.grad(fp_map(next(it)), w, grad_outputs=xs)
        return [v - j for v, j in zip(xs, J)]

    dot = lambda u, v: sum((a * c).sum() for a, c in zip(u, v))
    x_last = [torch.zeros_like(v) for v in b]
    r_last = [v.clone() for v in b]
    p_last = [v.clone() for v in b]
    for _ in range(K):                                          # hypergrad.CG_torch.cg
        Ap = A(p_last)
        rTr = dot(r_last, r_last)
        alpha = rTr / dot(p_last, Ap)
        x = [xx + alpha * pp for xx, pp in zip(x_last, p_last)]
        r = [rr - alpha * ap for rr, ap in zip(r_last, Ap)]
        if float(torch.sqrt(dot(r, r))) < tol:
            break
        beta = dot(r, r) / rTr
        p_last = [rr + beta * pp for rr, pp in zip(r, p_last)]
        x_last, r_last = x, r
    # grads = (dPhi/d theta)^T v + dL_q/d theta: the map sees theta only through the proximal term -> lr * reg * v
    hyper = {k: lr * reg_param * v for k, v in zip(names, x_last)}
    return qloss, fast, hyper, dict(zip(names, x_last))


# --------------------------------------------------------------------------------------
# outer update (lightning/optimizer.py:6-16, lightning/scheduler.py:6-29, main.py:61)
# --------------------------------------------------------------------------------------
def noam_lr(step: int, d_model: int = 256, warm_up_step: int = 4000,
            anneal_steps: Sequence[int] = (300000, 400000, 500000), anneal_rate: float = 0.3) -> float:
    """LambdaLR factor times init_lr = d_model^-0.5; ``step`` is the 0-based scheduler step."""
    cur = step + 1
    lr = min(cur ** -0.5, warm_up_step ** -1.5 * cur)
    for s in anneal_steps:
        if cur > s:
            lr *= anneal_rate
    return float(d_model ** -0.5 * lr)


def clip_grad_norm_(grads: Sequence[torch.Tensor], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ semantics (PL gradient_clip_val, main.py:61): global L2
    norm, scale by max_norm / (norm + 1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.detach().double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adam_step(param, grad, m, v, step: int, lr: float, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0):
    """torch.optim.Adam single-tensor update (optimizer.py:9-15); ``step`` is 1-based."""
    if weight_decay != 0.0:
        grad = grad + weight_decay * param
    m.mul_(betas[0]).add_(grad, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(grad, grad, value=1 - betas[1])
    bc1 = 1 - betas[0] ** step
    bc2 = 1 - betas[1] ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)
