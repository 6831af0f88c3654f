"""`TacotronSTFT` on libmtts.so (reference audio/stft.py:128-178).

The bases are built on the host exactly as the reference builds its buffers — `np.fft.fft(np.eye(n))` split into real / imaginary
rows and multiplied by the periodic Hann window (stft.py:27-46; scipy.signal.get_window, as there) — and the Slaney mel filter
bank of `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` (librosa is not part of this image: restated from its documented
algorithm, htk=False, norm="slaney").  The device does the framing, both contractions, magnitude, log and energy."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from ..engine import MttsError


def forward_basis(filter_length: int, win_length: int, window: str = "hann") -> np.ndarray:
    """[2 * (filter_length // 2 + 1)][filter_length] float32 — STFT.forward_basis without its singleton channel axis."""
    from scipy.signal import get_window
    fb = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    basis = np.vstack([np.real(fb[:cutoff, :]), np.imag(fb[:cutoff, :])]).astype(np.float32)   # torch.FloatTensor(...) in the reference
    assert filter_length >= win_length
    w = get_window(window, win_length, fftbins=True)
    lpad = (filter_length - win_length) // 2                                                    # librosa.util.pad_center
    w = np.pad(w, (lpad, filter_length - win_length - lpad))
    return (basis * w.astype(np.float32)[None, :]).astype(np.float32)


def _hz_to_mel(f):
    f = np.asanyarray(f, np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel (Slaney scale, slaney area normalisation): [n_mels][n_fft // 2 + 1] float32."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    fftfreqs = np.linspace(0, float(sr) / 2, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower, upper = -ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


class TacotronSTFT:
    """audio/stft.py:128: same constructor arguments; `mel_spectrogram(y)` takes (B, T) waveforms in [-1, 1] and returns
    (mel (B, n_mel, T'), energy (B, T')) float32 arrays."""

    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax, *, max_samples=22050 * 40,
                 device=0, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.filter_length, self.hop_length, self.win_length = filter_length, hop_length, win_length
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.forward_basis = forward_basis(filter_length, win_length)
        self.mel_basis = mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin or 0.0, mel_fmax)
        h = C.c_void_p()
        if self.lib.mtts_stft_create(filter_length, hop_length, n_mel_channels, max_samples, device, C.byref(h)) != 0:
            raise MttsError(self.lib.mtts_stft_last_error(None).decode())
        self.h = h
        self._check(self.lib.mtts_stft_load(self.h, self.forward_basis.ctypes.data_as(C.c_void_p), self.mel_basis.ctypes.data_as(C.c_void_p)))

    def _check(self, rc):
        if rc < 0:
            raise MttsError(self.lib.mtts_stft_last_error(self.h).decode())
        return rc

    def set_stream(self, stream_ptr: int):
        if self.lib.mtts_stft_set_stream(self.h, C.c_void_p(stream_ptr)) != 0:
            raise RuntimeError("mtts_stft_set_stream failed")

    def close(self):
        if getattr(self, "h", None):
            self.lib.mtts_stft_destroy(self.h)
            self.h = None

    __del__ = close

    def mel_spectrogram(self, y):
        y = np.ascontiguousarray(np.asarray(y.detach().cpu().numpy() if hasattr(y, "detach") else y, np.float32))
        assert y.ndim == 2
        assert y.min() >= -1 and y.max() <= 1      # stft.py:168-169
        T = y.shape[1] // self.hop_length + 1
        mel = np.empty((y.shape[0], T, self.n_mel_channels), np.float32)
        energy = np.empty((y.shape[0], T), np.float32)
        for b in range(y.shape[0]):
            got = self._check(self.lib.mtts_stft_mel_spectrogram(self.h, y[b].ctypes.data_as(C.c_void_p), y.shape[1], mel[b].ctypes.data_as(C.c_void_p),
                                                                 energy[b].ctypes.data_as(C.c_void_p)))
            assert got == T
        return mel.transpose(0, 2, 1), energy
