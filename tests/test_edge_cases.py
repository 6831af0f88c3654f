"""Edge cases of the hot path against the oracle: zero-length phonemes (LengthRegulator with duration 0 — leading, trailing
and whole runs), pitch / energy targets outside the quantisation range (bucketize -> first / last bin), single-utterance
and single-phoneme batches.  CPU: the product's sources through the SIMT emulator; GPU: the real kernels (full-size model)."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle_util import O, heads, synth, tiny_dims, torch_buffers, torch_params
from meta_tts_amd.config import ModelDims
from meta_tts_amd.engine import Engine

MODS = ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"]


def _edge_batch(dims, seed, B):
    """durations in [0, 4] with forced zeros at the ends, out-of-range pitch / energy targets."""
    b = list(synth.make_batch(seed, B, speaker=3, n_mel=dims.n_mel, vocab=dims.vocab, s_range=(6, 13), d_range=(0, 5), first_len=12))
    dur, src_lens = b[11].copy(), b[4]
    for i in range(B):
        s = int(src_lens[i])
        dur[i, 0] = 0                       # leading zero-length phoneme
        dur[i, s - 1] = 0                   # trailing one
        if s > 6:
            dur[i, 3:5] = 0                 # a run in the middle
        if dur[i, :s].sum() < 4:
            dur[i, 1] = 4
    mel_lens = dur.sum(axis=1)
    T = int(mel_lens.max())
    g = np.random.RandomState(seed + 1000)
    mels = np.zeros((B, T, dims.n_mel), np.float32)
    for i in range(B):
        mels[i, :mel_lens[i]] = g.standard_normal((int(mel_lens[i]), dims.n_mel))
    pit, ene = b[9].copy(), b[10].copy()
    pit[:, 0], pit[:, 1], pit[:, 2] = -50.0, 50.0, dims.pitch_min      # below min, above max, exactly min
    ene[:, 0], ene[:, 1], ene[:, 2] = -50.0, 50.0, dims.energy_max
    for i in range(B):                       # keep the padding zero
        s = int(src_lens[i])
        pit[i, s:] = 0; ene[i, s:] = 0
    b[6], b[7], b[8], b[9], b[10], b[11] = mels, mel_lens.astype(np.int64), T, pit.astype(np.float32), ene.astype(np.float32), dur
    return tuple(b)


def _check(eng, dims, batch, tol_out, tol_grad):
    eng.set_batches(0, [batch])
    eng.forward(0, use_fast=False, train=True)
    dev_loss = eng.loss(0)
    eng.backward(0, use_fast=False, scale=1.0, need_encoder=True)
    p = torch_params(dims, requires_grad=True)
    tb = O.to_torch_batch(batch)
    o = O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True)
    lo = O.fs2_loss(tb, o)
    out = eng.outputs(0, 0)
    for k, ref in (("mel", o[0]), ("mel_post", o[1]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
        assert np.abs(out[k] - ref.detach().numpy()).max() < tol_out, k
    np.testing.assert_allclose(dev_loss[0], [float(x) for x in lo], rtol=1e-4)
    names = ["variance_adaptor.pitch_embedding.weight", "variance_adaptor.energy_embedding.weight", "encoder.src_word_emb.weight",
             "variance_adaptor.duration_predictor.linear_layer.weight", "mel_linear.weight", "speaker_emb.model.weight",
             "decoder.layer_stack.0.slf_attn.w_qs.weight"]
    gs = torch.autograd.grad(lo[0], [p[n] for n in names])
    for n, g in zip(names, gs):
        ref = g.numpy()
        got = eng.export(n, 2, 0)
        assert np.abs(got - ref).max() <= tol_grad * np.abs(ref).max() + 2e-7, n
    return out


def test_zero_durations_and_out_of_range_targets_emulator():
    dims = tiny_dims()
    eng = Engine(dims, adapt_modules=MODS, max_tasks=1, max_B=3, max_S=16, max_T=96, lib_path=ge.build_emulator())
    eng.load_params(synth.make_params(dims, 0))
    b = _edge_batch(dims, 11, 3)
    assert (b[11][:, 0] == 0).all() and b[7].min() >= 4
    _check(eng, dims, b, 5e-5, 1e-3)
    # first / last embedding rows received the out-of-range targets' gradient
    g = eng.export("variance_adaptor.pitch_embedding.weight", 2, 0)
    assert np.abs(g[0]).max() > 0 and np.abs(g[-1]).max() > 0
    # single utterance, then a single-phoneme utterance
    _check(eng, dims, _edge_batch(dims, 12, 1), 5e-5, 1e-3)
    one = list(synth.make_batch(13, 1, speaker=1, n_mel=dims.n_mel, vocab=dims.vocab, s_range=(1, 2), d_range=(5, 6), first_len=1))
    assert one[5] == 1 and one[8] == 5
    _check(eng, dims, tuple(one), 5e-5, 1e-3)
    eng.close()


@pytest.mark.gpu
def test_zero_durations_and_out_of_range_targets_gpu():
    dims = ModelDims()
    eng = Engine(dims, adapt_modules=MODS, max_tasks=1, max_B=3, max_S=16, max_T=96)
    eng.load_params(synth.make_params(dims, 0))
    _check(eng, dims, _edge_batch(dims, 11, 3), 3e-4, 5e-3)
    _check(eng, dims, _edge_batch(dims, 12, 1), 3e-4, 5e-3)
    one = list(synth.make_batch(13, 1, speaker=1, s_range=(1, 2), d_range=(5, 6), first_len=1))
    _check(eng, dims, tuple(one), 3e-4, 5e-3)
    eng.close()


@pytest.mark.gpu
def test_longer_than_max_seq_len_follows_the_reference_in_both_modes():
    """T = 80 phonemes x ~13.5 frames > max_seq_len = 1000 (transformer/Models.py:145-162): in training mode the decoder keeps
    the first 1000 frames and the loss compares against the truncated target (loss.py:43-44); in eval mode nothing is
    dropped — the sinusoid table is extended.  Full-size model, one utterance; eval, train, eval again."""
    dims = ModelDims()
    b = synth.make_batch(14, 1, speaker=5, s_range=(80, 81), d_range=(12, 16), first_len=80)
    assert b[8] > dims.max_seq_len
    eng = Engine(dims, adapt_modules=MODS, max_tasks=1, max_B=1, max_S=80, max_T=int(b[8]))
    eng.load_params(synth.make_params(dims, 0))
    eng.set_batches(0, [b])
    p = torch_params(dims)
    buf = torch_buffers(dims)          # one set: the train-mode pass moves the BatchNorm running statistics on both sides
    tb = O.to_torch_batch(b)
    for train in (False, True, False):
        eng.forward(0, use_fast=False, train=train)
        dev_loss = eng.loss(0)
        out = eng.outputs(0, 0)
        with torch.no_grad():
            o = O.fs2_forward(p, buf, *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=train)
            lo = O.fs2_loss(tb, o)
        T = dims.max_seq_len if train else int(b[8])
        assert o[1].shape[1] == T and out["mel_post"].shape[1] == T
        tol = (1e-3, 5e-3) if train else (1e-4, 1e-3)     # train-mode BatchNorm over one utterance amplifies fp32 noise
        assert np.abs(out["mel_post"] - o[1].numpy()).mean() < tol[0] and np.abs(out["mel_post"] - o[1].numpy()).max() < tol[1]
        np.testing.assert_allclose(dev_loss[0], [float(x) for x in lo], rtol=5e-4)
    eng.close()
