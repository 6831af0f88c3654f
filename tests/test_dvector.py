"""d-vector speaker encoder (csrc/dvector.h through include/mtts.h: mtts_dvector_*) against oracle/dvector_oracle.py — torch's own
nn.LSTM / nn.Linear (the operators the reference's GE2E class is made of, speaker_encoder.py:11-31) and the utterance reduction of
speaker_encoder.py:71-76.  Small shapes through the SIMT emulator on CPU, the reference's shapes (40 mels, 160 frames, 3 x 256)
on the MI355X."""
import numpy as np
import pytest

import __graft_entry__ as ge
from meta_tts_amd import speaker_encoder as se
from oracle import dvector_oracle as orc


def _case(seed, n_utts, frames, cfg):
    g = np.random.RandomState(seed)
    counts = g.randint(1, 4, size=n_utts)
    n = int(counts.sum())
    mels = g.standard_normal((n, frames, cfg["n_mels"])).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(counts)])
    slices = [slice(int(off[i]), int(off[i + 1])) for i in range(n_utts)]
    return mels, slices


def _run(lib_path, cfg, frames, seed, n_utts):
    sd = se.synthetic_state_dict(seed, **cfg)
    enc = se.DVectorEncoder(sd, max_partials=64, max_utts=16, frames=frames, lib_path=lib_path, **cfg)
    mels, slices = _case(seed + 1, n_utts, frames, cfg)
    out, part = enc.embed(mels, slices, return_partials=True)
    ref_p = orc.partial_embeds(sd, mels, **cfg).numpy()
    ref = orc.speaker_embeds(sd, mels, slices, **cfg).numpy()
    np.testing.assert_allclose(part, ref_p, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out, ref, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, rtol=1e-5)
    # __call__ keeps the reference's (ref_mels, ref_slices) signature
    np.testing.assert_array_equal(enc((mels, slices)), out)
    with pytest.raises(ValueError):
        enc.embed(mels, slices[:-1])
    enc.close()


def test_dvector_emulator_small():
    _run(ge.build_emulator(), dict(n_mels=8, hidden=64, emb=64, layers=2), frames=7, seed=3, n_utts=3)


def test_state_dict_names_and_errors():
    cfg = dict(n_mels=8, hidden=64, emb=64, layers=1)
    sd = se.synthetic_state_dict(0, **cfg)
    ck = {"model.speaker_emb.model." + k: v for k, v in sd.items()}          # the reference checkpoint's prefix
    enc = se.DVectorEncoder(ck, max_partials=4, max_utts=2, frames=5, lib_path=ge.build_emulator(), **cfg)
    assert set(enc.state_dict()) == set(sd)
    bad = dict(sd)
    bad["linear.weight"] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        enc.load_state_dict(bad)
    with pytest.raises(ValueError):
        enc.embed(np.zeros((1, 6, 8), np.float32), [slice(0, 1)])            # wrong frame count
    with pytest.raises(Exception, match="at least one partial"):
        enc.embed(np.zeros((2, 5, 8), np.float32), [slice(0, 2), slice(2, 2)])   # an utterance without partial utterances
    with pytest.raises(ValueError):
        enc.embed(np.zeros((5, 5, 8), np.float32), [slice(0, 5)])            # more partials than the encoder was created for
    enc.close()


@pytest.mark.gpu
def test_dvector_gpu_reference_shapes():
    ge.build_device()
    _run(None, dict(n_mels=40, hidden=256, emb=256, layers=3), frames=160, seed=11, n_utts=5)


def _train_case(lib_path, cfg, frames, seed, n_utts, tol):
    """Back-propagation through time of the trained variants (speaker_emb: encoder / scratch_encoder) against torch autograd of the
    oracle (nn.LSTM + nn.Linear), then one joint-norm-clipped Adam step against torch.optim.Adam."""
    import ctypes as C
    import torch
    sd = se.synthetic_state_dict(seed, **cfg)
    enc = se.DVectorEncoder(sd, max_partials=64, max_utts=16, frames=frames, lib_path=lib_path, **cfg)
    enc.enable_training()
    mels, slices = _case(seed + 1, n_utts, frames, cfg)
    g = np.random.RandomState(seed + 2)
    dout = g.standard_normal((n_utts, cfg["emb"])).astype(np.float32)
    out = enc((mels, slices))                                   # training forward
    enc.backward(dout)
    lstm, linear = orc.build(sd, **cfg)
    _, (hidden, _) = lstm(torch.from_numpy(mels))
    raw = torch.relu(linear(hidden[-1]))
    pe = raw / torch.norm(raw, dim=1, keepdim=True)
    emb = torch.stack([torch.nn.functional.normalize(pe[sl].mean(dim=0), dim=0) for sl in slices])
    np.testing.assert_allclose(out, emb.detach().numpy(), rtol=2e-4, atol=2e-5)
    (emb * torch.from_numpy(dout)).sum().backward()
    ref = {f"lstm.{n}": p.grad.numpy() for n, p in lstm.named_parameters()}
    ref.update({f"linear.{n}": p.grad.numpy() for n, p in linear.named_parameters()})
    for name, r in ref.items():
        got = enc.export(name, 1)
        assert np.abs(got - r).max() <= tol * np.abs(r).max() + 1e-7, (name, np.abs(got - r).max(), np.abs(r).max())
    # Adam with the clip coefficient of a joint norm (here: the encoder alone, handed over as a device scalar)
    total = float(np.sqrt(sum((r.astype(np.float64) ** 2).sum() for r in ref.values())))
    max_norm = 0.5 * total
    lib = enc.lib
    norm = np.array([total], np.float32)
    if lib_path is None:
        tn = torch.from_numpy(norm).cuda()
        ptr = C.c_void_p(tn.data_ptr())
    else:
        ptr = norm.ctypes.data_as(C.c_void_p)
    enc.adam_step(ptr, max_norm, 1e-2)
    params = list(lstm.parameters()) + list(linear.parameters())
    torch.nn.utils.clip_grad_norm_(params, max_norm)
    opt = torch.optim.Adam(params, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
    opt.step()
    # first Adam step: w -= lr * g / (|g| + eps) — entries whose (clipped) gradient is of the order of eps = 1e-9 move by a
    # rounding-dependent fraction of the step, so compare where |g| >> eps
    for n, p in [(f"lstm.{n}", p) for n, p in lstm.named_parameters()] + [("linear.weight", linear.weight)]:
        live = np.abs(p.grad.numpy()) > 1e-6
        assert live.mean() > 0.2      # rows of dead ReLU units have no gradient at all
        np.testing.assert_allclose(enc.export(n)[live], p.detach().numpy()[live], rtol=1e-4, atol=5e-6, err_msg=n)
    # the updated weights are the ones the next forward uses
    out2 = enc.embed(mels, slices)
    ref2 = orc.speaker_embeds({k: enc.export(k) for k in sd}, mels, slices, **cfg).numpy()
    np.testing.assert_allclose(out2, ref2, rtol=2e-4, atol=2e-5)
    enc.close()


def test_dvector_training_emulator_small():
    _train_case(ge.build_emulator(), dict(n_mels=8, hidden=64, emb=64, layers=2), frames=7, seed=4, n_utts=3, tol=2e-4)


@pytest.mark.gpu
def test_dvector_training_gpu_reference_shapes():
    ge.build_device()
    _train_case(None, dict(n_mels=40, hidden=256, emb=256, layers=3), frames=160, seed=12, n_utts=4, tol=2e-3)
