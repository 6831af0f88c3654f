"""Waveform -> log-mel + energy (csrc/melfront.h through include/mtts.h: mtts_stft_*) against oracle/stft_oracle.py (torch
restatement of audio/stft.py:15-77,128-178 and audio/tools.py:8-15) AND against tests/golden/stft.npz — outputs of the reference's own
TacotronSTFT / get_mel_from_wav (tests/golden/make_stft_golden.py).  Small transforms through the SIMT emulator, the reference's
LibriTTS configuration (1024 / 256 / 1024, 80 mels, 22050 Hz) on the MI355X."""
import os
import numpy as np
import pytest

import __graft_entry__ as ge
from meta_tts_amd.audio import stft as S
from meta_tts_amd.audio import tools
from oracle import stft_oracle as orc


def _wave(n, sr, seed):
    g = np.random.RandomState(seed)
    t = np.arange(n) / sr
    w = 0.4 * np.sin(2 * np.pi * 220 * t) + 0.3 * np.sin(2 * np.pi * 1870 * t + 1.0) + 0.05 * g.standard_normal(n)
    w[n // 3] = 1.7   # one sample outside [-1, 1]: get_mel_from_wav clips
    return w.astype(np.float32)


def _run(lib_path, n_fft, hop, win, n_mel, sr, n, max_samples):
    st = S.TacotronSTFT(n_fft, hop, win, n_mel, sr, 0, None, max_samples=max_samples, lib_path=lib_path)
    wav = _wave(n, sr, n_fft + n)
    mel, energy = tools.get_mel_from_wav(wav, st)
    rmel, renergy = orc.mel_spectrogram(wav, n_fft, hop, win, st.mel_basis)
    assert mel.shape == rmel.shape == (n_mel, n // hop + 1) and mel.dtype == np.float32
    np.testing.assert_allclose(energy, renergy, rtol=2e-5, atol=2e-5)
    # log of a clamped value: compare where the mel energy is above the clamp, and exactly at the clamp elsewhere
    live = rmel > np.log(2e-5)
    np.testing.assert_allclose(mel[live], rmel[live], rtol=0, atol=2e-4)
    assert np.all(mel[~live] <= np.log(3e-5))
    with pytest.raises(Exception):
        tools.get_mel_from_wav(wav[: n_fft // 2], st)      # too short for the reflection padding
    st.close()


def test_bases_follow_the_reference_construction():
    b = S.forward_basis(64, 64)
    assert b.shape == (66, 64) and b.dtype == np.float32
    np.testing.assert_allclose(b, orc.forward_basis(64, 64)[:, 0, :].numpy(), rtol=0, atol=0)
    m = S.mel_filterbank(22050, 1024, 80, 0, None)
    assert m.shape == (80, 513) and (m >= 0).all() and (m.sum(axis=1) > 0).all()
    # slaney normalisation: every triangle has (approximately) unit area in Hz
    hz_per_bin = 22050 / 1024
    np.testing.assert_allclose(m.sum(axis=1) * hz_per_bin, 1.0, rtol=0.12)
    peak = m.argmax(axis=1)
    assert np.all(np.diff(peak) > 0)                       # centre frequencies increase


def test_stft_emulator_small():
    _run(ge.build_emulator(), 64, 16, 64, 12, 8000, 500, 1024)


def test_stft_lengths_and_window_shorter_than_filter():
    """Frame count = n // hop + 1 for lengths that are / are not multiples of the hop, the shortest legal waveform, and a window
    shorter than the filter (centre-padded, stft.py:38-41)."""
    lib = ge.build_emulator()
    st = S.TacotronSTFT(64, 16, 48, 12, 8000, 0, None, max_samples=400, lib_path=lib)
    for n in (33, 64, 320, 333, 400):
        wav = _wave(n, 8000, n)
        mel, energy = tools.get_mel_from_wav(wav, st)
        rmel, renergy = orc.mel_spectrogram(wav, 64, 16, 48, st.mel_basis)
        assert mel.shape == (12, n // 16 + 1) == rmel.shape
        np.testing.assert_allclose(energy, renergy, rtol=2e-5, atol=2e-5)
        live = rmel > np.log(2e-5)
        np.testing.assert_allclose(mel[live], rmel[live], rtol=0, atol=2e-4)
    with pytest.raises(Exception):
        tools.get_mel_from_wav(_wave(401, 8000, 1), st)        # longer than max_samples
    with pytest.raises(Exception):
        S.TacotronSTFT(66, 16, 64, 12, 8000, 0, None, lib_path=lib)   # filter_length % 4 != 0
    st.close()


@pytest.mark.gpu
def test_stft_gpu_libritts_configuration():
    ge.build_device()
    _run(None, 1024, 256, 1024, 80, 22050, 22050 * 3 + 77, 22050 * 4)


# ---- the reference's own outputs (tests/golden/stft.npz) ----------------------------------------------------------------------
def _golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stft.npz"))


@pytest.mark.parametrize("tag", ["small", "short_window", "libritts"])
def test_oracle_matches_reference_fixture(tag):
    """Pins oracle/stft_oracle.py: same waveform, same mel basis -> the reference's mel / energy; the windowed Fourier basis bit for bit."""
    g = _golden()
    n_fft, hop, win, n_mel, sr, n = (int(x) for x in g[tag + "_cfg"])
    mel, energy = orc.mel_spectrogram(g[tag + "_wav"], n_fft, hop, win, g[tag + "_mel_basis"])
    np.testing.assert_allclose(energy, g[tag + "_energy"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(mel, g[tag + "_mel"], rtol=0, atol=5e-5)
    if tag != "libritts":
        np.testing.assert_array_equal(orc.forward_basis(n_fft, win).numpy(), g[tag + "_forward_basis"])
        np.testing.assert_array_equal(S.forward_basis(n_fft, win), g[tag + "_forward_basis"][:, 0, :])


def _device_vs_fixture(lib_path, tag):
    g = _golden()
    n_fft, hop, win, n_mel, sr, n = (int(x) for x in g[tag + "_cfg"])
    st = S.TacotronSTFT(n_fft, hop, win, n_mel, sr, 0, None, max_samples=n + 64, lib_path=lib_path)
    np.testing.assert_array_equal(st.mel_basis, g[tag + "_mel_basis"])     # the shim of the generating script IS this filter bank
    mel, energy = tools.get_mel_from_wav(g[tag + "_wav"], st)
    rmel, renergy = g[tag + "_mel"], g[tag + "_energy"]
    assert mel.shape == rmel.shape
    np.testing.assert_allclose(energy, renergy, rtol=2e-5, atol=2e-5)
    live = rmel > np.log(2e-5)
    np.testing.assert_allclose(mel[live], rmel[live], rtol=0, atol=2e-4)
    assert np.all(mel[~live] <= np.log(3e-5))
    st.close()


@pytest.mark.parametrize("tag", ["small", "short_window"])
def test_stft_emulator_vs_reference_fixture(tag):
    _device_vs_fixture(ge.build_emulator(), tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "short_window", "libritts"])
def test_stft_gpu_vs_reference_fixture(tag):
    ge.build_device()
    _device_vs_fixture(None, tag)
