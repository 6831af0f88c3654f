#!/usr/bin/env python3
"""Generates tests/golden/stft.npz by running the REFERENCE's own mel front-end (/root/reference/audio/stft.py: STFT, TacotronSTFT;
audio/tools.py: get_mel_from_wav) in the build container.  Shims (the container has neither librosa nor a GPU):
  * librosa.util.pad_center / tiny / normalize: one-liners restated below (used for the window padding only);
  * librosa.filters.mel: the reference takes its mel basis from librosa — absent here and genuinely unobtainable, so the shim returns
    meta_tts_amd.audio.stft.mel_filterbank (the basis is stored in the fixture as an INPUT; what the fixture pins is everything the
    reference computes around it: Fourier basis x padded periodic Hann window, reflect padding, strided conv framing, magnitude,
    mel projection, log(clamp(., 1e-5)), per-frame energy, the [-1, 1] clip of get_mel_from_wav);
  * torch.Tensor.cuda: identity (STFT.transform hard-codes .cuda()).
The reference never travels: only this script and the arrays it writes are committed."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from meta_tts_amd.audio import stft as ours  # noqa: E402  (mel basis shim only)


def pad_center(data, size, axis=-1, **kw):
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, **kw)


def tiny(x):
    return np.finfo(np.asarray(x).dtype if np.issubdtype(np.asarray(x).dtype, np.floating) else np.float32).tiny


def normalize(S, norm=np.inf, **kw):
    return S if norm is None else S / np.max(np.abs(S))


librosa = types.ModuleType("librosa")
librosa.util = types.ModuleType("librosa.util")
librosa.util.pad_center, librosa.util.tiny, librosa.util.normalize = pad_center, tiny, normalize
librosa.filters = types.ModuleType("librosa.filters")
librosa.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: ours.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
sys.modules.update({"librosa": librosa, "librosa.util": librosa.util, "librosa.filters": librosa.filters})
torch.Tensor.cuda = lambda self, *a, **k: self

from audio.stft import TacotronSTFT  # noqa: E402  (the reference)
from audio.tools import get_mel_from_wav  # noqa: E402


def wave(n, sr, seed):   # tests/test_stft.py: _wave
    g = np.random.RandomState(seed)
    t = np.arange(n) / sr
    w = 0.4 * np.sin(2 * np.pi * 220 * t) + 0.3 * np.sin(2 * np.pi * 1870 * t + 1.0) + 0.05 * g.standard_normal(n)
    w[n // 3] = 1.7
    return w.astype(np.float32)


out = {}
for tag, (n_fft, hop, win, n_mel, sr, n) in {"small": (64, 16, 64, 12, 8000, 500), "short_window": (64, 16, 48, 12, 8000, 333),
                                             "libritts": (1024, 256, 1024, 80, 22050, 22050 + 77)}.items():
    st = TacotronSTFT(n_fft, hop, win, n_mel, sr, 0, None)
    wav = wave(n, sr, n_fft + n)
    mel, energy = get_mel_from_wav(wav, st)
    out[tag + "_cfg"] = np.asarray([n_fft, hop, win, n_mel, sr, n], np.int64)
    out[tag + "_wav"] = wav
    out[tag + "_mel_basis"] = st.mel_basis.numpy()
    out[tag + "_mel"] = mel
    out[tag + "_energy"] = energy
    if tag != "libritts":
        out[tag + "_forward_basis"] = st.stft_fn.forward_basis.numpy()
np.savez_compressed(os.path.join(HERE, "stft.npz"), **out)
print({k: v.shape for k, v in out.items()})
