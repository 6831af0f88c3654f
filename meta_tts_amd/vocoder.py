"""MelGAN generator host layer (mirror of `LightningMelGAN`, reference lightning/utils.py:8-30).

The reference pulls the vocoder from torch.hub ("descriptinc/melgan-neurips", `load_melgan("multi_speaker")`) and
calls `vocoder.mel2wav(mel / ln 10)`; neither that code nor its weights are part of the reference tree, so this class
takes a state dict in the hub module's naming (`model.<i>.weight_g / weight_v / bias`, residual blocks
`model.<i>.block.{2,4}.*` and `model.<i>.shortcut.*`) or synthetic weights, folds the weight normalisation and re-packs
the ConvTranspose1d kernels into the polyphase images `libmtts.so` consumes (include/mtts.h, mtts_vocoder_*).
"""
from __future__ import annotations

import ctypes as C
import math
import zlib

import numpy as np

from . import _lib
from .engine import MttsError

RATIOS = (8, 8, 2, 2)


def generator_spec(n_mel=80, ngf=32, n_res=3, ratios=RATIOS):
    """[(hub name prefix, kind, shape of weight_v, dilation)] in module order; kinds: conv / convT."""
    spec = []
    mult = 2 ** len(ratios)
    idx = 1  # model.0 = ReflectionPad1d(3)
    spec.append((f"model.{idx}", "conv", (mult * ngf, n_mel, 7)))
    idx += 1
    for r in ratios:
        idx += 1  # LeakyReLU
        spec.append((f"model.{idx}", "convT", (mult * ngf, mult * ngf // 2, 2 * r)))
        idx += 1
        for j in range(n_res):
            c = mult * ngf // 2
            spec.append((f"model.{idx}.block.2", "conv", (c, c, 3)))
            spec.append((f"model.{idx}.block.4", "conv", (c, c, 1)))
            spec.append((f"model.{idx}.shortcut", "conv", (c, c, 1)))
            idx += 1
        mult //= 2
    idx += 2  # LeakyReLU, ReflectionPad1d(3)
    spec.append((f"model.{idx}", "conv", (1, ngf, 7)))
    return spec


def synthetic_state_dict(seed=0, n_mel=80, ngf=32, n_res=3, ratios=RATIOS):
    """Deterministic weights in the hub naming (weight_g / weight_v / bias).  The gains keep activations O(1) through the
    12 residual blocks: a weight-normed conv output has variance ~ g^2 per unit-variance input, so plain convs use g = 1,
    the two branches of a residual block 1/sqrt(2) each, and a ConvTranspose1d (norm per INPUT channel over [C_out][2r],
    two taps x C_in terms per output) sqrt(r * C_out / C_in)."""
    sd = {}
    for name, kind, shape in generator_spec(n_mel, ngf, n_res, ratios):
        g = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        v = g.standard_normal(shape).astype(np.float32)
        if kind == "convT":
            gain = math.sqrt((shape[2] // 2) * shape[1] / shape[0])
            nb = shape[1]
        else:
            gain = 1.0 / math.sqrt(2.0) if (name.endswith("block.4") or name.endswith("shortcut")) else 1.0
            nb = shape[0]
        sd[name + ".weight_v"] = v
        sd[name + ".weight_g"] = np.full((shape[0], 1, 1), gain, np.float32)
        sd[name + ".bias"] = (0.05 * g.standard_normal(nb)).astype(np.float32)
    return sd


def fold_weight_norm(sd, name):
    """w = g * v / ||v|| with the norm over every dim but 0 (torch.nn.utils.weight_norm default, also for ConvTranspose1d)."""
    if name + ".weight" in sd:
        return np.asarray(sd[name + ".weight"], np.float32)
    v = np.asarray(sd[name + ".weight_v"], np.float64)
    g = np.asarray(sd[name + ".weight_g"], np.float64).reshape(-1, 1, 1)
    n = np.sqrt((v ** 2).sum(axis=(1, 2), keepdims=True))
    return (g * v / n).astype(np.float32)


def pack_tensors(sd, n_mel=80, ngf=32, n_res=3, ratios=RATIOS):
    """hub-style state dict -> {libmtts tensor name: float32 array} (layouts documented in include/mtts.h)."""
    out = {}
    spec = generator_spec(n_mel, ngf, n_res, ratios)
    it = iter(spec)
    name, _, _ = next(it)
    out["conv_in.w"] = np.ascontiguousarray(fold_weight_norm(sd, name).transpose(0, 2, 1))  # [C0][7][n_mel]
    out["conv_in.b"] = np.asarray(sd[name + ".bias"], np.float32)
    for s, r in enumerate(ratios):
        name, _, shape = next(it)
        w = fold_weight_norm(sd, name)  # [C_in][C_out][2r]
        cin, cout, _ = shape
        img = np.empty((r, cout, 2 * cin), np.float32)
        for ph in range(r):
            img[ph, :, :cin] = w[:, :, ph + r].T   # multiplies x[q-1]
            img[ph, :, cin:] = w[:, :, ph].T       # multiplies x[q]
        out[f"up{s}.w"] = img
        out[f"up{s}.b"] = np.asarray(sd[name + ".bias"], np.float32)
        for j in range(n_res):
            n1, _, _ = next(it); n2, _, _ = next(it); ns, _, _ = next(it)
            out[f"res{s}.{j}.w1"] = np.ascontiguousarray(fold_weight_norm(sd, n1).transpose(0, 2, 1))  # [C][3][C]
            out[f"res{s}.{j}.b1"] = np.asarray(sd[n1 + ".bias"], np.float32)
            out[f"res{s}.{j}.w2"] = np.ascontiguousarray(fold_weight_norm(sd, n2)[:, :, 0])
            out[f"res{s}.{j}.b2"] = np.asarray(sd[n2 + ".bias"], np.float32)
            out[f"res{s}.{j}.ws"] = np.ascontiguousarray(fold_weight_norm(sd, ns)[:, :, 0])
            out[f"res{s}.{j}.bs"] = np.asarray(sd[ns + ".bias"], np.float32)
    name, _, _ = next(it)
    out["conv_out.w"] = np.ascontiguousarray(fold_weight_norm(sd, name)[0].T)  # [7][C_last]
    out["conv_out.b"] = np.asarray(sd[name + ".bias"], np.float32).reshape(1)
    return out


class MelGAN:
    """`LightningMelGAN` drop-in: inverse(mel) and infer(mels, max_wav_value, lengths) with the reference's semantics
    (mel (B, n_mel, T) like the reference's call sites; output waveform (B, T * hop))."""

    def __init__(self, state_dict=None, n_mel=80, ngf=32, n_res=3, ratios=RATIOS, max_B=8, max_T=1024, device=0, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.n_mel, self.ratios = n_mel, tuple(ratios)
        self.max_B, self.max_T = max_B, max_T
        h = C.c_void_p()
        arr = (C.c_int * len(ratios))(*ratios)
        if self.lib.mtts_vocoder_create(n_mel, ngf, n_res, arr, len(ratios), device, max_B, max_T, C.byref(h)) != 0:
            raise MttsError(self.lib.mtts_vocoder_last_error(None).decode())
        self.h = h
        self.hop = self.lib.mtts_vocoder_hop(h)
        sd = state_dict if state_dict is not None else synthetic_state_dict(0, n_mel, ngf, n_res, ratios)
        self.state_dict_ = sd
        for k, v in pack_tensors(sd, n_mel, ngf, n_res, ratios).items():
            a = np.ascontiguousarray(v, np.float32)
            if self.lib.mtts_vocoder_load(h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size) != 0:
                raise MttsError(self.lib.mtts_vocoder_last_error(h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.mtts_vocoder_destroy(self.h)
            self.h = None

    def set_stream(self, stream_ptr):
        self.lib.mtts_vocoder_set_stream(self.h, C.c_void_p(stream_ptr))

    def mel2wav(self, mel, lengths=None, mel_scale=1.0):
        """mel (B, n_mel, T) float -> (B, T * hop) float32; rows beyond lengths[b] * hop are zero."""
        mel = np.asarray(mel, np.float32)
        B, nm, T = mel.shape
        assert nm == self.n_mel
        lens = np.full(B, T, np.int32) if lengths is None else np.asarray(lengths, np.int32)
        x = np.ascontiguousarray(mel.transpose(0, 2, 1))  # channels-last rows
        wav = np.zeros((B, T * self.hop), np.float32)
        if self.lib.mtts_vocoder_infer(self.h, x.ctypes.data_as(C.c_void_p), B, T, lens.ctypes.data_as(C.c_void_p),
                                       C.c_float(mel_scale), wav.ctypes.data_as(C.c_void_p)) != 0:
            raise MttsError(self.lib.mtts_vocoder_last_error(self.h).decode())
        return wav

    def mel2wav_device(self, mel_ptr: int, utt_stride: int, B: int, T: int, lengths, wav_ptr: int, mel_scale: float = 1.0):
        """Device-to-device: mel at `mel_ptr` laid out [B][..][n_mel] with `utt_stride` floats between utterances (e.g.
        Engine.mel_device()), waveform written to `wav_ptr` as [B][T * hop] floats; asynchronous on the vocoder's stream."""
        lens = np.ascontiguousarray(lengths, np.int32)
        if self.lib.mtts_vocoder_infer_device(self.h, C.c_void_p(mel_ptr), C.c_int64(utt_stride), B, T, lens.ctypes.data_as(C.c_void_p),
                                              C.c_float(mel_scale), C.c_void_p(wav_ptr)) != 0:
            raise MttsError(self.lib.mtts_vocoder_last_error(self.h).decode())

    def inverse(self, mel):                      # lightning/utils.py:16-18
        return self.mel2wav(mel)

    def infer(self, mels, max_wav_value, lengths=None):   # lightning/utils.py:20-30
        mels = np.asarray(mels, np.float32)
        frame_lens = None if lengths is None else [int(math.ceil(l / self.hop)) for l in lengths]
        wavs = self.mel2wav(mels, None if frame_lens is None else np.minimum(frame_lens, mels.shape[2]), mel_scale=1.0 / math.log(10.0))
        wavs = (wavs * max_wav_value).astype("int16")
        wavs = [w for w in wavs]
        for i in range(len(mels)):
            if lengths is not None:
                wavs[i] = wavs[i][: lengths[i]]
        return wavs
