"""TEST INFRASTRUCTURE ONLY — torch restatement of the reference's mel front-end, the checker of meta_tts_amd/csrc/melfront.h.

PINNED (round 3) by tests/golden/stft.npz — outputs of the REFERENCE's own audio/stft.py + audio/tools.py run in the build container by
tests/golden/make_stft_golden.py (librosa's three padding helpers shimmed, `.cuda()` made the identity): the windowed Fourier basis,
mel and energy at a small, a short-window and the LibriTTS configuration (tests/test_stft.py).  What stays unpinned is the mel FILTER
BANK only: the reference takes it from librosa.filters.mel, which is absent here, so the basis is an input of this oracle and of the
fixture.  Restated: STFT.__init__ (stft.py:27-46: np.fft.fft(np.eye(n)) -> real | imaginary rows -> FloatTensor -> times the padded
periodic Hann window), STFT.transform (:52-77: reflect pad n/2, F.conv1d with stride hop, sqrt(re^2 + im^2)),
TacotronSTFT.mel_spectrogram (:159-178: matmul with the mel basis, log(clamp(., 1e-5)), torch.norm over frequency) and
audio/tools.py:8-15 (clip).  Only tests/, __graft_entry__.smoke() and bench.py may import it."""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window


def forward_basis(filter_length, win_length, window="hann"):
    fourier_basis = np.fft.fft(np.eye(filter_length))
    cutoff = int(filter_length / 2 + 1)
    fourier_basis = np.vstack([np.real(fourier_basis[:cutoff, :]), np.imag(fourier_basis[:cutoff, :])])
    basis = torch.FloatTensor(fourier_basis[:, None, :])
    fft_window = get_window(window, win_length, fftbins=True)
    lpad = (filter_length - win_length) // 2
    fft_window = np.pad(fft_window, (lpad, filter_length - win_length - lpad))
    basis = basis * torch.from_numpy(fft_window).float()
    return basis.float()


def mel_spectrogram(wav, filter_length, hop_length, win_length, mel_basis):
    """wav (n,) -> (mel (n_mel, T), energy (T,))."""
    y = torch.clip(torch.FloatTensor(np.asarray(wav, np.float32)).unsqueeze(0), -1, 1)
    x = F.pad(y.view(1, 1, -1).unsqueeze(1), (int(filter_length / 2), int(filter_length / 2), 0, 0), mode="reflect").squeeze(1)
    ft = F.conv1d(x, forward_basis(filter_length, win_length), stride=hop_length, padding=0)
    cutoff = int(filter_length / 2 + 1)
    mag = torch.sqrt(ft[:, :cutoff, :] ** 2 + ft[:, cutoff:, :] ** 2)
    mel = torch.log(torch.clamp(torch.matmul(torch.from_numpy(np.asarray(mel_basis, np.float32)), mag), min=1e-5))
    return mel[0].numpy(), torch.norm(mag, dim=1)[0].numpy()
