"""TEST INFRASTRUCTURE ONLY — torch restatement of the reference's mel front-end, the checker of meta_tts_amd/csrc/melfront.h.

PARITY UNPINNED: audio/stft.py cannot be run here to make fixtures — `STFT.transform` moves its operands with hard-coded
`.cuda()` calls (stft.py:67-68; there is no GPU in the build container, and torch on the GPU box is not the reference) and the module
imports librosa (absent).  Restated line by line instead: STFT.__init__ (stft.py:27-46: np.fft.fft(np.eye(n)) -> real | imaginary
rows -> FloatTensor -> times the padded periodic Hann window), STFT.transform (:52-77: reflect pad n/2, F.conv1d with stride hop,
sqrt(re^2 + im^2)), TacotronSTFT.mel_spectrogram (:159-178: matmul with the mel basis, log(clamp(., 1e-5)), torch.norm over
frequency) and audio/tools.py:8-15 (clip).  The mel basis is an input here (librosa.filters.mel in the reference).  Only tests/,
__graft_entry__.smoke() and bench.py may import it."""
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
