"""Deterministic synthetic parameters and LibriTTS-shaped batches.

There is no network for corpora or checkpoints, so tests, golden fixtures and
bench.py all draw their inputs from here:

* weights: ``np.random.RandomState(crc32(name) ^ seed)`` scaled per tensor, keyed by
  the reference state_dict name, so any implementation regenerates them
  bit-identically without shipping 141 MB of floats;
* batches: the 12-tuple layout of lightning/collate.py:47-60 with lengths drawn
  as in SURVEY.md section 8(d).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from .config import ModelDims


def param_spec(dims: ModelDims) -> "OrderedDict[str, Tuple[Tuple[int, ...], bool]]":
    """name -> (shape, trainable), in reference ``state_dict`` order
    (names from lightning/model/fastspeech2.py:23-32,
    transformer/Models.py:56-70, transformer/SubLayers.py:18-26,71-83,
    lightning/model/modules.py:20-78,209-240, transformer/Layers.py:83-127)."""
    d = dims.d_model
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], bool]]" = OrderedDict()

    def fft(prefix: str, n_layers: int):
        for i in range(n_layers):
            p = f"{prefix}.layer_stack.{i}"
            for w in ("w_qs", "w_ks", "w_vs"):
                spec[f"{p}.slf_attn.{w}.weight"] = ((d, d), True)
                spec[f"{p}.slf_attn.{w}.bias"] = ((d,), True)
            spec[f"{p}.slf_attn.layer_norm.weight"] = ((d,), True)
            spec[f"{p}.slf_attn.layer_norm.bias"] = ((d,), True)
            spec[f"{p}.slf_attn.fc.weight"] = ((d, d), True)
            spec[f"{p}.slf_attn.fc.bias"] = ((d,), True)
            spec[f"{p}.pos_ffn.w_1.weight"] = ((dims.d_ff, d, dims.k1), True)
            spec[f"{p}.pos_ffn.w_1.bias"] = ((dims.d_ff,), True)
            spec[f"{p}.pos_ffn.w_2.weight"] = ((d, dims.d_ff, dims.k2), True)
            spec[f"{p}.pos_ffn.w_2.bias"] = ((d,), True)
            spec[f"{p}.pos_ffn.layer_norm.weight"] = ((d,), True)
            spec[f"{p}.pos_ffn.layer_norm.bias"] = ((d,), True)

    spec["encoder.position_enc"] = ((1, dims.max_seq_len + 1, d), False)
    spec["encoder.src_word_emb.weight"] = ((dims.vocab, d), True)
    fft("encoder", dims.enc_layers)
    spec["variance_adaptor.pitch_bins"] = ((dims.n_bins - 1,), False)
    spec["variance_adaptor.energy_bins"] = ((dims.n_bins - 1,), False)
    f, k = dims.vp_filter, dims.vp_kernel
    for pred in ("duration_predictor", "pitch_predictor", "energy_predictor"):
        p = f"variance_adaptor.{pred}"
        spec[f"{p}.conv_layer.conv1d_1.conv.weight"] = ((f, d, k), True)
        spec[f"{p}.conv_layer.conv1d_1.conv.bias"] = ((f,), True)
        spec[f"{p}.conv_layer.layer_norm_1.weight"] = ((f,), True)
        spec[f"{p}.conv_layer.layer_norm_1.bias"] = ((f,), True)
        spec[f"{p}.conv_layer.conv1d_2.conv.weight"] = ((f, f, k), True)
        spec[f"{p}.conv_layer.conv1d_2.conv.bias"] = ((f,), True)
        spec[f"{p}.conv_layer.layer_norm_2.weight"] = ((f,), True)
        spec[f"{p}.conv_layer.layer_norm_2.bias"] = ((f,), True)
        spec[f"{p}.linear_layer.weight"] = ((1, f), True)
        spec[f"{p}.linear_layer.bias"] = ((1,), True)
    spec["variance_adaptor.pitch_embedding.weight"] = ((dims.n_bins, d), True)
    spec["variance_adaptor.energy_embedding.weight"] = ((dims.n_bins, d), True)
    spec["decoder.position_enc"] = ((1, dims.max_seq_len + 1, d), False)
    fft("decoder", dims.dec_layers)
    spec["mel_linear.weight"] = ((dims.n_mel, d), True)
    spec["mel_linear.bias"] = ((dims.n_mel,), True)
    pd, pk = dims.postnet_dim, dims.postnet_kernel
    chans = [dims.n_mel] + [pd] * (dims.postnet_layers - 1) + [dims.n_mel]
    for i in range(dims.postnet_layers):
        spec[f"postnet.convolutions.{i}.0.conv.weight"] = ((chans[i + 1], chans[i], pk), True)
        spec[f"postnet.convolutions.{i}.0.conv.bias"] = ((chans[i + 1],), True)
        spec[f"postnet.convolutions.{i}.1.weight"] = ((chans[i + 1],), True)
        spec[f"postnet.convolutions.{i}.1.bias"] = ((chans[i + 1],), True)
    spec["speaker_emb.model.weight"] = ((dims.n_speaker, d), True)
    return spec


def buffer_spec(dims: ModelDims) -> "OrderedDict[str, Tuple[int, ...]]":
    """BatchNorm1d buffers of the PostNet (transformer/Layers.py:95,110,125)."""
    pd = dims.postnet_dim
    chans = [pd] * (dims.postnet_layers - 1) + [dims.n_mel]
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for i, c in enumerate(chans):
        out[f"postnet.convolutions.{i}.1.running_mean"] = (c,)
        out[f"postnet.convolutions.{i}.1.running_var"] = (c,)
        out[f"postnet.convolutions.{i}.1.num_batches_tracked"] = ()
    return out


ADAPT_PREFIXES_DEFAULT = ("speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet")


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """transformer/Models.py:10-30 — angle = pos / 10000^(2*(j//2)/d), sin on even, cos on odd
    columns, computed in float64 then cast to float32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid, dtype=np.float64)[None, :]
    table = pos / np.power(10000.0, 2.0 * np.floor(j / 2.0) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return table.astype(np.float32)


def make_param(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    g = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0xFFFFFFFF)
    x = g.standard_normal(size=shape).astype(np.float32)
    if len(shape) >= 2:
        if "emb" in name:  # embedding tables
            x *= np.float32(0.3)
        else:
            fan_in = int(np.prod(shape[1:]))
            x *= np.float32(1.0 / np.sqrt(fan_in))
    else:
        if name.endswith("weight"):  # LayerNorm / BatchNorm gains
            x = np.float32(1.0) + np.float32(0.1) * x
        else:
            x *= np.float32(0.1)
    return np.ascontiguousarray(x, dtype=np.float32)


def make_params(dims: ModelDims, seed: int = 0, weight_scale: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """All parameters (trainable and frozen) under reference names.  ``weight_scale`` multiplies every Linear / Conv1d
    weight matrix (not the embedding tables, gains or biases): 0.5 gives a model on which five inner SGD steps at the
    reference's lr = 1e-3 are contractive (tests/golden maml_small_lr1e-3_scaled.npz)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, trainable) in param_spec(dims).items():
        if name.endswith("position_enc"):
            out[name] = sinusoid_table(dims.max_seq_len + 1, dims.d_model)[None]
        elif name.endswith("pitch_bins"):
            # torch.linspace(min, max, n_bins - 1) in fp32 (lightning/model/modules.py:57-60)
            out[name] = np.linspace(dims.pitch_min, dims.pitch_max, dims.n_bins - 1).astype(np.float32)
        elif name.endswith("energy_bins"):
            out[name] = np.linspace(dims.energy_min, dims.energy_max, dims.n_bins - 1).astype(np.float32)
        else:
            out[name] = make_param(name, shape, seed)
            if weight_scale != 1.0 and len(shape) >= 2 and "emb" not in name:
                out[name] *= np.float32(weight_scale)
    out["encoder.src_word_emb.weight"][0] = 0.0  # padding_idx=0 row (transformer/Models.py:56-58)
    return out


def make_buffers(dims: ModelDims) -> "OrderedDict[str, np.ndarray]":
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in buffer_spec(dims).items():
        if name.endswith("running_var"):
            out[name] = np.ones(shape, np.float32)
        elif name.endswith("running_mean"):
            out[name] = np.zeros(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.int64)
    return out


def make_batch(seed: int, B: int, speaker: int = 7, n_mel: int = 80, vocab: int = 361,
               s_range: Tuple[int, int] = (40, 81), d_range: Tuple[int, int] = (2, 13),
               first_len: int | None = 80, pitch_level: str = "phoneme_level", energy_level: str = "phoneme_level"):
    """One padded batch in the reference 12-tuple layout (lightning/collate.py:47-60) as numpy arrays.

    Draw order is fixed by SURVEY.md section 8(d): src_lens, then per utterance dur, then per
    utterance text, pitch, energy, mel."""
    g = np.random.RandomState(seed)
    src_lens = g.randint(s_range[0], s_range[1], size=B).astype(np.int64)
    if first_len is not None:
        src_lens[0] = first_len
    durs = [g.randint(d_range[0], d_range[1], size=int(s)).astype(np.int64) for s in src_lens]
    mel_lens = np.array([int(d.sum()) for d in durs], dtype=np.int64)
    S, T = int(src_lens.max()), int(mel_lens.max())
    texts = np.zeros((B, S), np.int64)
    pitches = np.zeros((B, S), np.float32)
    energies = np.zeros((B, S), np.float32)
    durations = np.zeros((B, S), np.int64)
    mels = np.zeros((B, T, n_mel), np.float32)
    for i in range(B):
        s, t = int(src_lens[i]), int(mel_lens[i])
        texts[i, :s] = g.randint(1, vocab, size=s)
        pitches[i, :s] = g.standard_normal(s).astype(np.float32)
        energies[i, :s] = g.standard_normal(s).astype(np.float32)
        mels[i, :t] = g.standard_normal((t, n_mel)).astype(np.float32)
        durations[i, :s] = durs[i]
    # frame-level features (preprocess config `pitch.feature: frame_level`): one value per mel frame, drawn after everything else
    if pitch_level == "frame_level":
        pitches = np.zeros((B, T), np.float32)
        for i in range(B):
            pitches[i, :int(mel_lens[i])] = g.standard_normal(int(mel_lens[i])).astype(np.float32)
    if energy_level == "frame_level":
        energies = np.zeros((B, T), np.float32)
        for i in range(B):
            energies[i, :int(mel_lens[i])] = g.standard_normal(int(mel_lens[i])).astype(np.float32)
    ids = [f"synth-{seed}-{i}" for i in range(B)]
    raw_texts = ["" for _ in range(B)]
    speakers = np.full((B,), speaker, np.int64)
    return (ids, raw_texts, speakers, texts, src_lens, S, mels, mel_lens, T, pitches, energies, durations)


def make_task(j: int, B: int = 5, **kw):
    """Task j of a meta-batch: support seed 2j+1, query seed 2j+2, speaker 7+j."""
    sup = make_batch(2 * j + 1, B, speaker=7 + j, **kw)
    qry = make_batch(2 * j + 2, B, speaker=7 + j, **kw)
    return sup, qry


def contraction_macs(dims: ModelDims, src_lens, mel_lens, adapted_only: bool = False) -> int:
    """Valid (unpadded) multiply-accumulates of one forward (SURVEY.md section 8(d) formula,
    generalised to ``dims``): FFT block = QKV + out-proj + conv1 + conv2 + 2 attention products."""
    d, ff = dims.d_model, dims.d_ff

    def blk(L):
        return (4 * d * d + dims.k1 * d * ff + dims.k2 * ff * d) * L + 2 * d * L * L

    vp = dims.vp_kernel * d * dims.vp_filter + dims.vp_kernel * dims.vp_filter * dims.vp_filter + dims.vp_filter
    pn = dims.postnet_kernel * (2 * dims.n_mel * dims.postnet_dim
                                + (dims.postnet_layers - 2) * dims.postnet_dim ** 2)
    total = 0
    for s, t in zip(src_lens, mel_lens):
        s, t = int(s), int(min(t, dims.max_seq_len))
        if not adapted_only:
            total += dims.enc_layers * blk(s)
        total += 3 * vp * s + dims.dec_layers * blk(t) + d * dims.n_mel * t + pn * t
    return int(total)
