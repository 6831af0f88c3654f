"""TEST INFRASTRUCTURE ONLY — torch restatement of the MelGAN generator used as the checker of meta_tts_amd/csrc/vocoder.h.

PARITY UNPINNED: the generator is an un-vendored torch.hub dependency of the reference (`descriptinc/melgan-neurips`,
loaded at lightning/utils.py:11-14, no commit pin; absent from /root/reference and not fetchable here).  This file
restates the published architecture of that hub entry (Kumar et al., "MelGAN", NeurIPS 2019: Generator(input_size 80,
ngf 32, n_residual_layers 3), ratios [8, 8, 2, 2], weight-normed Conv1d / ConvTranspose1d, LeakyReLU(0.2), reflection
padding, ResnetBlock = shortcut(x) + block(x)); the call semantics (mel / ln 10 in, x max_wav_value, int16, crop) follow
the reference's own wrapper, lightning/utils.py:16-30.  Only tests/, __graft_entry__.smoke() and bench.py may import it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _w(sd, name):
    if name + ".weight" in sd:
        return torch.as_tensor(np.asarray(sd[name + ".weight"], np.float32))
    v = torch.as_tensor(np.asarray(sd[name + ".weight_v"], np.float32)).double()
    g = torch.as_tensor(np.asarray(sd[name + ".weight_g"], np.float32)).double().reshape(-1, 1, 1)
    return (g * v / v.pow(2).sum(dim=(1, 2), keepdim=True).sqrt()).float()   # torch.nn.utils.weight_norm, dim=0


def _b(sd, name):
    return torch.as_tensor(np.asarray(sd[name + ".bias"], np.float32))


def mel2wav(sd, mel, n_res=3, ratios=(8, 8, 2, 2)):
    """mel: (B, n_mel, T) float32 -> (B, T * prod(ratios)); module order of the hub Generator."""
    x = torch.as_tensor(np.asarray(mel, np.float32))
    idx = 1
    x = F.conv1d(F.pad(x, (3, 3), mode="reflect"), _w(sd, f"model.{idx}"), _b(sd, f"model.{idx}"))
    idx += 1
    for r in ratios:
        idx += 1
        x = F.conv_transpose1d(F.leaky_relu(x, 0.2), _w(sd, f"model.{idx}"), _b(sd, f"model.{idx}"), stride=r,
                               padding=r // 2 + r % 2, output_padding=r % 2)
        idx += 1
        for j in range(n_res):
            d = 3 ** j
            p = f"model.{idx}"
            h = F.conv1d(F.pad(F.leaky_relu(x, 0.2), (d, d), mode="reflect"), _w(sd, p + ".block.2"), _b(sd, p + ".block.2"), dilation=d)
            h = F.conv1d(F.leaky_relu(h, 0.2), _w(sd, p + ".block.4"), _b(sd, p + ".block.4"))
            x = F.conv1d(x, _w(sd, p + ".shortcut"), _b(sd, p + ".shortcut")) + h
            idx += 1
    idx += 2
    x = F.conv1d(F.pad(F.leaky_relu(x, 0.2), (3, 3), mode="reflect"), _w(sd, f"model.{idx}"), _b(sd, f"model.{idx}"))
    return torch.tanh(x).squeeze(1).numpy()


def infer(sd, mels, max_wav_value, lengths=None, **kw):
    """lightning/utils.py:20-30."""
    wavs = mel2wav(sd, np.asarray(mels, np.float32) / math.log(10.0), **kw)
    wavs = (wavs * max_wav_value).astype("int16")
    wavs = [w for w in wavs]
    for i in range(len(mels)):
        if lengths is not None:
            wavs[i] = wavs[i][: lengths[i]]
    return wavs
