"""audio/tools.py:8-15."""
import numpy as np


def get_mel_from_wav(audio, _stft):
    """wav (n_samples,) -> (melspec (n_mel, T) float32, energy (T,) float32); the waveform is clipped to [-1, 1] first."""
    audio = np.clip(np.asarray(audio, np.float32)[None, :], -1, 1)
    melspec, energy = _stft.mel_spectrogram(audio)
    return melspec[0].astype(np.float32), energy[0].astype(np.float32)
