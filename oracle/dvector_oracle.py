"""TEST INFRASTRUCTURE ONLY — torch restatement of the d-vector speaker encoder used as the checker of meta_tts_amd/csrc/dvector.h.

PARITY UNPINNED for the encoder's forward: `VoiceEncoder` is the un-vendored `resemblyzer` package (imported at
lightning/model/speaker_encoder.py:7, absent here, so that module cannot be imported to generate fixtures).  Restated: the
architecture from the reference's own GE2E class (speaker_encoder.py:11-31: nn.LSTM(40, 256, 3, batch_first=True),
nn.Linear(256, 256), ReLU) — nn.LSTM itself IS the reference's operator and is called directly here — and resemblyzer's published
`VoiceEncoder.forward` (final hidden state of the last layer -> linear -> relu -> divide by the L2 norm).  The utterance-level
reduction follows speaker_encoder.py:71-76 line by line.  Only tests/, __graft_entry__.smoke() and bench.py may import it."""
import numpy as np
import torch


def build(sd, n_mels=40, hidden=256, emb=256, layers=3):
    lstm = torch.nn.LSTM(n_mels, hidden, layers, batch_first=True)
    linear = torch.nn.Linear(hidden, emb)
    with torch.no_grad():
        for k in range(layers):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(lstm, f"{n}_l{k}").copy_(torch.as_tensor(np.asarray(sd[f"lstm.{n}_l{k}"], np.float32)))
        linear.weight.copy_(torch.as_tensor(np.asarray(sd["linear.weight"], np.float32)))
        linear.bias.copy_(torch.as_tensor(np.asarray(sd["linear.bias"], np.float32)))
    return lstm, linear


def partial_embeds(sd, mels, **kw):
    """resemblyzer VoiceEncoder.forward: (N, T, n_mels) -> (N, emb), L2-normalised."""
    lstm, linear = build(sd, **kw)
    with torch.no_grad():
        _, (hidden, _) = lstm(torch.as_tensor(np.asarray(mels, np.float32)))
        raw = torch.relu(linear(hidden[-1]))
        return raw / torch.norm(raw, dim=1, keepdim=True)


def speaker_embeds(sd, ref_mels, ref_slices, **kw):
    """speaker_encoder.py:71-76."""
    pe = partial_embeds(sd, ref_mels, **kw)
    embeds = [pe[sl].mean(dim=0) for sl in ref_slices]
    return torch.stack([torch.nn.functional.normalize(e, dim=0) for e in embeds])
